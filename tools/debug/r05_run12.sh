#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_trainer.py -m gpu -x -q -k "prefetcher" 2>&1 | tail -5
bash tools/debug/ab_env.sh "-" "STEM_BENCH_PIPELINE=0" 2>&1 | tee gpurun_out/r05_ab_pipeline.log
