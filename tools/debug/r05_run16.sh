#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "STEM_BENCH_TUNING=wg3_minch=64" "STEM_BENCH_TUNING=wg3_minch=86" "STEM_BENCH_TUNING=wg3_minch=128" "STEM_BENCH_TUNING=wg3_minch=256" "STEM_BENCH_TUNING=wg3_minch=48" 2>&1 | tee gpurun_out/r05_ab_wg3_minch.log
