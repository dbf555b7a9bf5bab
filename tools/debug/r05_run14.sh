#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "STEM_STREAM_CUMASK=latents=block:160" "STEM_STREAM_CUMASK=latents=block:128" "STEM_STREAM_CUMASK=latents=block:144" "STEM_STREAM_CUMASK=latents=mod8:5" "STEM_STREAM_CUMASK=latents=block:176" 2>&1 | tee gpurun_out/r05_ab_cumask2.log
