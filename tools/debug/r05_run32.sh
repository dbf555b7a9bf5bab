#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "-" "STEM_BENCH_TUNING=fx3_depth=2" "STEM_BENCH_TUNING=fx3_depth=2 STEM_STREAM_CUMASK=latents=block:192" "STEM_BENCH_TUNING=fx3_tile=64" 2>&1 | tee gpurun_out/r05_ab_depth.log
