#!/bin/bash
# collect_r06.sh: what profiles/ quotes for round 6, in one gpurun call -> gpurun_out/r06/
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06; mkdir -p $out; cd $R
timeout 900 python3 -m pytest tests/test_hip_spread.py tests/test_hip_trainer.py tests/test_hip_dp2.py -q -m gpu -s -p no:cacheprovider 2>&1 | grep -E "per channel|f64 gate|passed|failed|Error|assert" | cut -c1-500 > $out/tests_spread_trainer_dp2.log
python3 bench.py --config eval > $out/bench_eval_1080p_gop12.json 2> $out/bench_eval.err
STEM_DIST_SINGLE=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_rccl_world1.json 2> $out/bench_rccl_world1.err
STEM_BENCH_VERIFY=1 python3 bench.py --gpus 2 --steps 4 --warmup 3 --no-cpu-baseline > $out/bench_gpus2_one_device_verify.json 2> $out/bench_gpus2.err
bash tools/debug/prof_bench.sh r06/bench_trace > /dev/null 2>&1
bash tools/debug/prof_pmc.sh r06/pmc_ga2 tools/debug/f16x3_prof.py planes 5 > $out/pmc_ga2.log 2>&1
bash tools/debug/prof_tcc.sh r06/tcc_ga2 tools/debug/f16x3_prof.py planes 5 > $out/tcc_ga2.log 2>&1
bash tools/debug/prof_tcc.sh r06/tcc_c4gdn tools/debug/c4gdn_prof.py > $out/tcc_c4gdn.log 2>&1
bash tools/debug/prof_pmc.sh r06/pmc_c4gdn tools/debug/c4gdn_prof.py > $out/pmc_c4gdn.log 2>&1
bash tools/debug/prof_pmc.sh r06/pmc_img_tpm4 tools/debug/f16x3_img_prof.py TPM.4 img > $out/pmc_img_tpm4.log 2>&1
bash tools/debug/prof_pmc.sh r06/pmc_wgrad_row_tpm4 tools/debug/wgrad3_check.py TPM.4 > $out/pmc_wgrad_row_tpm4.log 2>&1
bash tools/debug/ab_env.sh "-" "STEM_STREAM_CUMASK=latents=block:192" "STEM_STREAM_CUMASK=latents=block:176" "STEM_STREAM_CUMASK=" > $out/ab_cumask.log 2>&1
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
ls -la $out
