#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_dp2.py tests/test_hip_trainer.py -m gpu -x -q > gpurun_out/r05_run9_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r05_run9_tests.log
bash tools/debug/ab_env.sh "-" "STEM_DIST_SINGLE=1" "STEM_DIST_SINGLE=1 STEM_DP_MIN_BYTES=25165824" 2>&1 | tee gpurun_out/r05_ab_rccl_world1.log
