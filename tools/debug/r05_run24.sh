#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "-" "STEM_ENGINE_BRANCH=0" "STEM_DIST_SINGLE=1 STEM_ENGINE_BRANCH=0" "STEM_DIST_SINGLE=1 STEM_ENGINE_OVERLAP=0" "STEM_DIST_SINGLE=1" 2>&1 | tee gpurun_out/r05_ab_rccl1_g.log
