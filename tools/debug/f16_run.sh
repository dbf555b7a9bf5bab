#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/f16
timeout 2800 python3 -m pytest tests -m gpu -q -x --deselect tests/test_hip_dp2.py 2>&1 | grep -v "^E    *+\|tensor(\[" | tail -8 > gpurun_out/f16/pytest.log
for i in 1 2 3; do for v in 0 1; do echo -n "adam_block_max=$v: "; STEM_ADAM_BLOCK_MAX=$v python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"ms_per_step\"],3), d[\"config\"][\"final_loss_bpp\"])"; done; done > gpurun_out/f16/ab.log 2>&1
