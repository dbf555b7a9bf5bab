#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/f16
timeout 2800 python3 -m pytest tests -m gpu -q -x --deselect tests/test_hip_dp2.py 2>&1 | grep -v "^E    *+\|tensor(\[" | tail -40 > gpurun_out/f16/pytest.log
timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 > gpurun_out/f16/bench.log
bash tools/debug/prof_bench.sh f16/trace > /dev/null 2>&1
