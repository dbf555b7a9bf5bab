#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "-" "STEM_ENGINE_CTX_ON_SIDE=1" "STEM_DIST_SINGLE=1" "STEM_DIST_SINGLE=1 STEM_ENGINE_TPM_WGRAD_INLINE=0" "STEM_DIST_SINGLE=1 STEM_STREAM_CUMASK=latents=block:192" 2>&1 | tee gpurun_out/r05_ab_rccl1_b.log
