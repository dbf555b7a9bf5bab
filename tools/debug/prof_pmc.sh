#!/bin/bash
# prof_pmc.sh <outdir under gpurun_out> <python script + args...>: one kernel-trace pass + three PMC passes, each under `timeout`
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; script=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -o t -- python3 $script "$@" > $out/trace.log 2>&1
find /tmp/p_trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
i=0
for pmc in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/p_pmc$i -o p -- python3 $script "$@" > $out/pmc$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/pmc.csv /tmp/p_pmc1 /tmp/p_pmc2 /tmp/p_pmc3 > /dev/null 2>&1
head -4 $out/kernel_stats.csv
grep -E "^kernel|conv_f16x3|igemm|c4gdn|wgrad_f16x3" $out/pmc.csv
