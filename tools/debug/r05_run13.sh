#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "-" "STEM_STREAM_CUMASK=latents=block:160" "STEM_STREAM_CUMASK=latents=block:224" "STEM_STREAM_CUMASK=" "STEM_BENCH_TUNING=fx3_mfma=32" 2>&1 | tee gpurun_out/r05_ab_cumask.log
