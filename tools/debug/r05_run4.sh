#!/bin/bash
mkdir -p gpurun_out
python tools/debug/tconv_sweep.py 2>&1 | tee gpurun_out/r05_tconv_sweep.log
bash tools/debug/ab_env.sh "-" "STEM_BENCH_TUNING=fx3_gen_img=1" "STEM_BENCH_TUNING=tconv_cps=12" "STEM_BENCH_TUNING=tconv_cps=24" 2>&1 | tee gpurun_out/r05_ab_img_tconv.log
