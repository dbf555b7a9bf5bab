#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "-" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=3" "GPU_MAX_HW_QUEUES=1" "STEM_DIST_SINGLE=1 GPU_MAX_HW_QUEUES=2" "STEM_DIST_SINGLE=1 GPU_MAX_HW_QUEUES=3" "STEM_DIST_SINGLE=1 GPU_MAX_HW_QUEUES=1" 2>&1 | tee gpurun_out/r05_ab_hwq2.log
