#!/bin/bash
# round 5, second GPU call: kernel trace of the default bench; configs[4] regression bisect over the two kernel forms round 4 added
mkdir -p gpurun_out
bash tools/debug/prof_bench.sh r05a_prof > gpurun_out/r05a_prof.log 2>&1
for t in "-" "fx3_gen_img=1" "wg3_row=1" "fx3_gen_img=1,wg3_row=1"; do
  if [ "$t" = "-" ]; then e=""; else e="$t"; fi
  STEM_BENCH_TUNING="$e" python bench.py --config roi --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('roi tuning [$t]', round(d['ms_per_step'],1), 'ms', d.get('roofline',{}).get('avg_launch_ms'))" | tee -a gpurun_out/r05_roi_bisect.log
done
