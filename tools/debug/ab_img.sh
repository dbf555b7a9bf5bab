#!/bin/bash
# in-step A/B of plan selectors: ab_img.sh "sel=val[,sel=val]" "sel=val" ...   (three alternating passes, 20 timed steps each)
for i in 1 2 3; do
for t in "$@"; do
 STEM_BENCH_TUNING=$t python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$t', round(d['ms_per_step'],3))"
done; done
