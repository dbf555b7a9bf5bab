#!/bin/bash
# in-step A/B of two builds of the library: ab_lib.sh <base.so> [passes]   (alternating, 20 timed steps each)
base=$1; n=${2:-3}
for i in $(seq $n); do
 for lib in "$base" ""; do
  STEM_HIP_LIBRARY=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${lib:-current}', round(d['ms_per_step'],3))"
 done
done
