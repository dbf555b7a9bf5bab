#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "-" "STEM_BENCH_TUNING=wg3_minch=16" "STEM_BENCH_TUNING=wg3_minch=64" "STEM_BENCH_TUNING=fx3_img_w=80" "STEM_BENCH_TUNING=fx3_img_w=300" "STEM_ENGINE_TPM_FIRST=0" "STEM_ENGINE_TPM_FIRST_BWD=0" "STEM_BENCH_TUNING=fx3_gen_tile=64" "STEM_BENCH_TUNING=fx3_gen_tile=128" 2>&1 | tee gpurun_out/r05_ab_knobs.log
