#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_trainer.py tests/test_hip_codec.py tests/test_hip_models.py -m gpu -x -q > gpurun_out/r05_run6_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r05_run6_tests.log
bash tools/debug/ab_env.sh "-" "STEM_ENGINE_PACK_FIRST=0" 2>&1 | tee gpurun_out/r05_ab_pack_first.log
