#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "-" "STEM_STREAM_CUMASK=latents=block:152" "STEM_STREAM_CUMASK=latents=block:168" "STEM_STREAM_CUMASK=latents=mod8:5" 2>&1 | tee gpurun_out/r05_ab_cumask3.log
for a in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --latents-ahead $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ahead $a', round(d['ms_per_step'],3))"; done | tee -a gpurun_out/r05_ab_cumask3.log
