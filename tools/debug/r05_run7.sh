#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_dynrange.py -m gpu -q -s > gpurun_out/r05_dynrange.log 2>&1
echo "dynrange rc=$?"; grep -E "per channel|passed|failed|Error|assert" gpurun_out/r05_dynrange.log | tail -60
python -m pytest tests/test_hip_roi.py -m gpu -x -q -s > gpurun_out/r05_roi_tests.log 2>&1
echo "roi rc=$?"; grep -E "f64 gate|passed|failed" gpurun_out/r05_roi_tests.log | tail -20
