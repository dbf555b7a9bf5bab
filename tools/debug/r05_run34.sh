#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_trainer.py tests/test_hip_models.py tests/test_hip_dp2.py -m gpu -x -q > gpurun_out/r05_run34_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r05_run34_tests.log
bash tools/debug/ab_env.sh "-" "STEM_TRAINER_OVERWRITE_GRADS=0" 2>&1 | tee gpurun_out/r05_ab_overwrite.log
