#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_trainer.py tests/test_hip_models.py tests/test_hip_fullsize.py -m gpu -x -q > gpurun_out/r05_run33_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r05_run33_tests.log
bash tools/debug/ab_env.sh "-" "STEM_ENGINE_EPM_DGRAD_BY_PRIOR=0" 2>&1 | tee gpurun_out/r05_ab_epm_by_prior.log
