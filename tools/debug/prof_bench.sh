#!/bin/bash
# prof_bench.sh <outdir under gpurun_out> [bench args]: kernel trace of bench.py + timeline summary
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 "$@" > $out/trace.log 2>&1
find /tmp/pb_trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
t=$(find /tmp/pb_trace -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then python3 $GRAFT_REPO_ROOT/tools/timeline.py $t 0.4 "conv_f16x3_kernel<128@262144" 21 > $out/timeline.txt 2>&1; fi
tail -1 $out/trace.log | cut -c1-160
head -45 $out/timeline.txt
