#!/bin/bash
# gantt2.sh <outdir under gpurun_out> [bench args...]: kernel trace of bench.py and a Gantt listing of one P-frame step
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/gt_trace
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gt_trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 "$@" > $out/trace.log 2>&1
t=$(find /tmp/gt_trace -name "*kernel_trace.csv" | head -1)
TIMELINE_DUMP_MS=2.6 TIMELINE_DUMP_FROM=adam_bmax_kernel python3 $GRAFT_REPO_ROOT/tools/timeline.py $t 0.4 > $out/gantt.txt 2>&1
