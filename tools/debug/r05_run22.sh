#!/bin/bash
mkdir -p gpurun_out
export STEM_DIST_SINGLE=1 STEM_DP_EXPERIMENT=fakestream
bash tools/debug/ab_env.sh "STEM_DP_FAKE=side" "STEM_DP_FAKE=default" "STEM_DP_FAKE=prio0" "STEM_DP_FAKE=dummy STEM_DP_DUMMIES=1" "STEM_DP_FAKE=dummy STEM_DP_DUMMIES=2" "STEM_DP_FAKE=dummy STEM_DP_DUMMIES=3" 2>&1 | tee gpurun_out/r05_ab_rccl1_e.log
