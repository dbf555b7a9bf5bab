#!/bin/bash
# round 5, first GPU call: the new tape tests, the whole GPU suite, a default bench, A/B of the split-K ticket's memory order
mkdir -p gpurun_out
python -m pytest tests/test_hip_trainer.py tests/test_hip_dp2.py tests/test_hip_fullsize.py -m gpu -x -q -k "taped or fresh or two_ranks" > gpurun_out/r05_new_tests.log 2>&1
echo "new tests rc=$?" | tee -a gpurun_out/r05_new_tests.log
tail -30 gpurun_out/r05_new_tests.log
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite.log 2>&1
echo "suite rc=$?" | tee -a gpurun_out/r05_gpu_suite.log
tail -5 gpurun_out/r05_gpu_suite.log
python bench.py > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err
tail -c 600 gpurun_out/r05_bench_a.json
bash tools/debug/ab_lib.sh $PWD/tools/debug/ab/libstem_hip_relaxed.so 2 2>&1 | tee gpurun_out/r05_ab_splitk_order.log
