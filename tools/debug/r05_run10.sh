#!/bin/bash
mkdir -p gpurun_out
export STEM_DIST_SINGLE=1
bash tools/debug/gantt2.sh r05_gantt_rccl1
grep -i "nccl\|rccl\|copy\|memset\|fill" gpurun_out/r05_gantt_rccl1/gantt.txt | head -20
