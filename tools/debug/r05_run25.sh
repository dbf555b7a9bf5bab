#!/bin/bash
mkdir -p gpurun_out
export STEM_DIST_SINGLE=1 STEM_DP_EXPERIMENT=fakestream
STEM_DP_FAKE=low python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep "priority range"
bash tools/debug/ab_env.sh "STEM_DP_FAKE=side" "STEM_DP_FAKE=low" "STEM_DP_FAKE=side GPU_MAX_HW_QUEUES=2" "STEM_DP_FAKE=side GPU_MAX_HW_QUEUES=16" 2>&1 | tee gpurun_out/r05_ab_rccl1_h.log
