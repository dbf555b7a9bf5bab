#!/bin/bash
mkdir -p gpurun_out
bash tools/debug/ab_env.sh "-" "STEM_STREAM_PRIO=latents=0,side=-1,compute=0,branch=-1" "STEM_STREAM_PRIO=latents=0,side=0,compute=0,branch=-1" "STEM_STREAM_PRIO=latents=0,side=-1,compute=-1,branch=0" "STEM_STREAM_PRIO=latents=1,side=-1,compute=-1,branch=-1 STEM_STREAM_CUMASK=" "STEM_STREAM_PRIO=latents=1,side=0,compute=0,branch=-1 STEM_STREAM_CUMASK=" 2>&1 | tee gpurun_out/r05_ab_prio.log
