#!/bin/bash
mkdir -p gpurun_out
export STEM_DIST_SINGLE=1 STEM_DP_EXPERIMENT=fakestream
bash tools/debug/ab_env.sh "STEM_DP_FAKEMODE=full" "STEM_DP_FAKEMODE=nofinal" "STEM_DP_FAKEMODE=reconly" "STEM_DP_FAKEMODE=full STEM_DP_MIN_BYTES=1000000000" "STEM_DP_FAKEMODE=nofinal STEM_DP_MIN_BYTES=1000000000" 2>&1 | tee gpurun_out/r05_ab_rccl1_f.log
