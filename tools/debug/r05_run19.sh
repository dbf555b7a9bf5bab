#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "-" "STEM_DIST_SINGLE=1" "STEM_DIST_SINGLE=1 STEM_DP_EXPERIMENT=nocoll" "STEM_DIST_SINGLE=1 STEM_DP_EXPERIMENT=nowait" "STEM_DIST_SINGLE=1 STEM_DP_MIN_BYTES=1000000000" 2>&1 | tee gpurun_out/r05_ab_rccl1_c.log
