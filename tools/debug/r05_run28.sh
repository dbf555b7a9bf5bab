#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "-" "STEM_STREAM_CUMASK=latents=block:128,side=tail:128,compute=tail:128" "STEM_STREAM_CUMASK=latents=block:112,side=tail:144,compute=tail:144" "STEM_STREAM_CUMASK=latents=block:96,side=tail:160,compute=tail:160" "STEM_STREAM_CUMASK=latents=block:144,side=tail:112,compute=tail:112" "STEM_STREAM_CUMASK=latents=block:128,side=tail:160,compute=tail:160" 2>&1 | tee gpurun_out/r05_ab_partition.log
