#!/bin/bash
# schedule diagnostics: default (prefetch), latents first, and each with the weight-gradient side stream off
for i in 1 2; do
for m in "--latents prefetch" "--latents first"; do
 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$m', round(d['ms_per_step'],3), 'g_a.2 in-region', d['roofline'].get('avg_launch_ms'), 'isolated', d['roofline'].get('isolated',{}).get('avg_launch_ms'))"
done; done
