#!/bin/bash
# collect_round_profiles.sh <tag>: everything profiles/ quotes for a round, in one gpurun call -> gpurun_out/<tag>/
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
cd $R
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --gpus 2 --steps 2 --warmup 1 > $out/bench_gpus2_one_device.json 2> $out/bench_gpus2.err
STEM_DIST_SINGLE=1 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_rccl_world1.json 2> $out/bench_rccl_world1.err
STEM_DIST_SINGLE=1 bash tools/debug/ab_env.sh "-" "STEM_DP_THREADED=2" "STEM_DP_THREADED=0" > $out/ab_rccl_world1_issue_modes.log 2>&1
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_no_group.json 2>/dev/null
python3 bench.py --config roi > $out/bench_roi_b16.json 2> $out/bench_roi.err
STEM_LAYERS_F16X3=0 python3 bench.py --config roi > $out/bench_roi_b16_fp32_layers.json 2>/dev/null
bash tools/debug/prof_bench.sh $tag/bench_trace > /dev/null 2>&1
bash tools/debug/prof_pmc.sh $tag/pmc_ga2 tools/debug/f16x3_prof.py planes 5 > $out/pmc_ga2.log 2>&1
bash tools/debug/prof_tcc.sh $tag/tcc_ga2 tools/debug/f16x3_prof.py planes 5 > $out/tcc_ga2.log 2>&1
bash tools/debug/prof_pmc.sh $tag/pmc_gen_tpm4 tools/debug/f16x3_gen_check.py TPM.4 > $out/pmc_gen_tpm4.log 2>&1
bash tools/debug/prof_pmc.sh $tag/pmc_c4gdn tools/debug/c4gdn_prof.py > $out/pmc_c4gdn.log 2>&1
bash tools/debug/prof_tcc.sh $tag/tcc_c4gdn tools/debug/c4gdn_prof.py > $out/tcc_c4gdn.log 2>&1
python3 tools/debug/c4gdn_time.py > $out/c4gdn_time.log 2>&1
python3 tools/debug/route_vs_oracle.py > $out/route_vs_oracle.log 2>&1
for t in f16x3_check f16x3_chain wgrad3_check f16x3_gen_check; do python3 tools/debug/$t.py 2>&1 | grep -v "amdgpu.ids\|Warning\|scale =" > $out/accuracy_$t.log; done
python3 tools/debug/f16x3_split_sweep.py gen 2>&1 | grep -v amdgpu > $out/split_sweep_gen.log
python3 tools/debug/f16x3_split_sweep.py wgrad 2>&1 | grep -v amdgpu > $out/split_sweep_wgrad.log
python3 tools/debug/f16x3_depth_sweep.py 2>&1 | grep -v amdgpu > $out/depth_sweep.log
python3 tools/eval_pframe_bench.py --frames 8 > $out/eval_1080p_persistent_default.log 2>&1
STEM_AR_PERSISTENT=0 python3 tools/eval_pframe_bench.py --frames 3 > $out/eval_1080p_per_position_loop.log 2>&1
python3 tools/eval_pframe_bench.py --frames 3 --sequences 8 > $out/eval_1080p_8_sequences_concurrent.log 2>&1
STEM_AR_CONCURRENT=0 python3 tools/eval_pframe_bench.py --frames 3 --sequences 8 > $out/eval_1080p_8_sequences_lockstep.log 2>&1
X=spatiotemporalentropymodel_amd/libstem_hip_exper.so; [ -f $X ] || X=spatiotemporalentropymodel_amd/libstem_hip_xtmp.so
if [ -f $X ]; then       # per-position phase timers of the persistent decoder
  STEM_HIP_LIBRARY=$R/$X python3 tools/eval_pframe_bench.py --frames 2 > $out/eval_1080p_persistent_phases.log 2>&1
fi
python3 tools/debug/host_lag.py > $out/host_lag.log 2>&1
python3 tools/host_overhead.py > $out/host_overhead.log 2>&1
STEM_HOST_TAPE=0 python3 tools/host_overhead.py > $out/host_overhead_python_enqueue.log 2>&1
# round 4: image-tile form of the general kernel, filter-row form of the weight gradient, event timeline of the default run
bash tools/debug/prof_pmc.sh $tag/pmc_img_tpm4 tools/debug/f16x3_img_prof.py TPM.4 img > $out/pmc_img_tpm4.log 2>&1
bash tools/debug/prof_pmc.sh $tag/pmc_img_tpm4_split5 tools/debug/f16x3_img_prof.py TPM.4 img 5 > $out/pmc_img_tpm4_split5.log 2>&1
bash tools/debug/prof_pmc.sh $tag/pmc_wgrad_row_tpm4 tools/debug/wgrad3_check.py TPM.4 > $out/pmc_wgrad_row_tpm4.log 2>&1
python3 tools/debug/wgrad3_check.py 2>&1 | grep -v "amdgpu.ids" > $out/wgrad_row_vs_per_tap.log
python3 tools/debug/f16x3_img_check.py 2>&1 | grep -v "amdgpu.ids" > $out/img_check.log
STEM_BENCH_TIMELINE=1 python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline 2> $out/event_timeline.log > /dev/null
ls -la $out
# round 5: the transposed faces on the fp16 kernel (one launch over four sub-pixel phases), the pair pack, a P-frame step alone
python3 tools/debug/tconv_sweep.py 2>&1 | grep -v "amdgpu.ids" > $out/tconv_sweep.log
python3 tools/debug/pack_time.py 2>&1 | grep -v "amdgpu.ids" > $out/pack_time.log
bash tools/debug/gantt2.sh $tag/gantt_palone --latents first > /dev/null 2>&1
bash tools/debug/gantt2.sh $tag/gantt_default > /dev/null 2>&1
python3 -m pytest tests/test_hip_dynrange.py -m gpu -q -s 2>&1 | grep -E "per channel|passed|failed" > $out/dynrange_per_channel.log
python3 -m pytest tests/test_hip_roi.py -m gpu -q -s 2>&1 | grep -E "f64 gate|passed|failed" > $out/roi_f64_gates.log
ls -la $out
