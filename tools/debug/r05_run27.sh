#!/bin/bash
mkdir -p gpurun_out
for t in "-" "-" "fx3_gen_img=3" "-" "fx3_gen_img=3"; do
  if [ "$t" = "-" ]; then e=""; else e="$t"; fi
  STEM_BENCH_TUNING="$e" python bench.py --config roi --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('roi tuning [$t]', round(d['ms_per_step'],1), 'ms')" | tee -a gpurun_out/r05_roi_rule_ab_warm.log
done
STEM_DIST_SINGLE=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('rccl world-1 (GPU_MAX_HW_QUEUES=2 set by bench.py):', round(d['ms_per_step'],3))" | tee gpurun_out/r05_rccl1_hwq2_auto.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('no group:', round(d['ms_per_step'],3), d['roofline'].get('cu_mask_cus'), d['roofline'].get('frac_of_masked_cus'))" | tee -a gpurun_out/r05_rccl1_hwq2_auto.log
