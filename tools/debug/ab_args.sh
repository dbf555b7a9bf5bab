#!/bin/bash
# in-step A/B of bench.py argument sets: ab_args.sh "<args A>" "<args B>" ...  (use "-" for the defaults; three alternating passes)
for i in 1 2 3; do
for t in "$@"; do
 if [ "$t" = "-" ]; then a=""; else a="$t"; fi
 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$t', round(d['ms_per_step'],3))"
done; done
