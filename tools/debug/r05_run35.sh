#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_trainer.py tests/test_hip_models.py -m gpu -x -q > gpurun_out/r05_run35_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r05_run35_tests.log
bash tools/debug/ab_env.sh "-" "STEM_ENGINE_SHARE_IN_PLANES=0" 2>&1 | tee gpurun_out/r05_ab_share_in.log
