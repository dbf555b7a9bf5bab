#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_models.py tests/test_hip_trainer.py tests/test_hip_ops.py -m gpu -x -q > gpurun_out/r05_run29_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r05_run29_tests.log
cp spatiotemporalentropymodel_amd/libstem_hip.so /tmp/new.so
bash tools/debug/ab_lib.sh $PWD/tools/debug/ab/libstem_hip_prev.so 3 2>&1 | tee gpurun_out/r05_ab_unpack_vec.log
