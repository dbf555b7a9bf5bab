#!/bin/bash
# Bench step time under scheduling switches, interleaved with the default on one box (dev tool):
#   bash tools/debug/bench_switch_sweep.sh > gpurun_out/switch_sweep.log
run() {   # label, env assignments..., -- bench args
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    local out
    out=$(env "${envs[@]}" python3 bench.py --no-cpu-baseline --steps 20 "$@" 2>/dev/null | tail -1 |
          python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))")
    echo "$label: $out"
}
for rep in 1 2; do
    run "default" -- 
    run "wgrad lanes 2" STEM_ENGINE_WGRAD_LANES=2 --
    run "context branch" STEM_ENGINE_CTX_BRANCH=1 --
    run "latents ahead 2" -- --latents-ahead 2
    run "latents on 176 CUs" STEM_STREAM_CUMASK=latents=block:176 --
    run "side priority 0" STEM_STREAM_PRIO=latents=0,side=0,compute=-1 --
    run "side on 224 CUs too" STEM_STREAM_CUMASK=latents=block:192,side=block:224 --
done
