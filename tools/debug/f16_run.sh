#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/f16
for v in "" "fx3_gen_tile=64" "fx3_gen_tile=128"; do echo "== $v"; python3 tools/debug/pytest_tuned.py $v -- tests/test_hip_roi.py tests/test_hip_f16x3.py -m gpu -q 2>&1 | tail -2; done > gpurun_out/f16/pytest.log
