#!/bin/bash
# in-step A/B of environment switches: ab_env.sh "VAR=val VAR2=val" "VAR=val" ...  (use "-" for the defaults; three alternating passes)
for i in 1 2 3; do
for t in "$@"; do
 if [ "$t" = "-" ]; then e=""; else e="$t"; fi
 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$t', round(d['ms_per_step'],3))"
done; done
