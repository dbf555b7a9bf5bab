#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_f16x3.py tests/test_hip_fullsize.py tests/test_hip_trainer.py -m gpu -x -q > gpurun_out/r05_run17_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r05_run17_tests.log
bash tools/debug/ab_env.sh "-" "STEM_BENCH_TUNING=wg3_split=3" "STEM_BENCH_TUNING=wg3_split=5" "STEM_STREAM_CUMASK=latents=block:144" "STEM_STREAM_CUMASK=latents=block:176" 2>&1 | tee gpurun_out/r05_ab_after_minch.log
