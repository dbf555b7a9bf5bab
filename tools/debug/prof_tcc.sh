#!/bin/bash
# prof_tcc.sh <outdir under gpurun_out> <script> <args...>: L2 hit/miss + HBM-side traffic passes (separate --pmc runs)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; script=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for pmc in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_READ_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/q_pmc$i -o p -- python3 $script "$@" > $out/tcc$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/tcc.csv /tmp/q_pmc1 /tmp/q_pmc2 /tmp/q_pmc3 /tmp/q_pmc4 /tmp/q_pmc5 > /dev/null 2>&1
grep -E "^kernel|conv_f16x3|igemm|c4gdn|wgrad_f16x3" $out/tcc.csv
