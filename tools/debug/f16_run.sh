#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/f16
for t in f16x3_check f16x3_chain wgrad3_check; do echo "=== $t"; timeout 300 python3 tools/debug/$t.py 2>&1 | grep -v "amdgpu.ids\|Consider\|scale =" | tail -14; done > gpurun_out/f16/acc.log 2>&1
echo "=== gen TPM.2" >> gpurun_out/f16/acc.log; timeout 300 python3 tools/debug/f16x3_gen_check.py TPM.2 2>&1 | tail -12 >> gpurun_out/f16/acc.log
timeout 600 python3 -m pytest tests/test_hip_dp2.py -m gpu -q -x 2>&1 | tail -3 >> gpurun_out/f16/acc.log
