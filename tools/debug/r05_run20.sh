#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "-" "GPU_MAX_HW_QUEUES=8" "STEM_DIST_SINGLE=1" "STEM_DIST_SINGLE=1 GPU_MAX_HW_QUEUES=8" "STEM_DIST_SINGLE=1 GPU_MAX_HW_QUEUES=6" 2>&1 | tee gpurun_out/r05_ab_hwq.log
