#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_trainer.py tests/test_hip_f16x3.py -m gpu -x -q -k "switches or transposed or taped_step_is_bit" > gpurun_out/r05_run5_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r05_run5_tests.log
bash tools/debug/ab_env.sh "-" "STEM_ENGINE_TPM_WGRAD_INLINE=0" 2>&1 | tee gpurun_out/r05_ab_tpm_wgrad_inline.log
bash tools/debug/gantt2.sh r05b_gantt_palone --latents first
