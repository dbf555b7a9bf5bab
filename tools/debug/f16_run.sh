#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/f16
timeout 2800 python3 -m pytest tests -m gpu -q 2>&1 | grep -v "^E    *+\|tensor(\[" | tail -60 > gpurun_out/f16/pytest.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 > gpurun_out/f16/smoke.log
