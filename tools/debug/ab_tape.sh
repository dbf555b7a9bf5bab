#!/bin/bash
for i in 1 2 3; do
for t in 0 1; do
 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tape $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tape $t', round(d['ms_per_step'],3))"
done; done
