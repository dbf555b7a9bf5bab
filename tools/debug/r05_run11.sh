#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_f16x3.py -m gpu -x -q -k "pair_pack" 2>&1 | tail -15
python -m pytest tests/test_hip_trainer.py tests/test_hip_models.py -m gpu -x -q > gpurun_out/r05_run11_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r05_run11_tests.log
bash tools/debug/ab_env.sh "-" "STEM_ENGINE_PACK_PAIR=0" 2>&1 | tee gpurun_out/r05_ab_pack_pair.log
