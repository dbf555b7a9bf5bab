#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_f16x3.py -m gpu -x -q -k "transposed or strided_weight" > gpurun_out/r05_tconv_tests.log 2>&1
echo "tconv unit tests rc=$?"; tail -25 gpurun_out/r05_tconv_tests.log
python -m pytest tests/test_hip_trainer.py tests/test_hip_models.py tests/test_hip_fullsize.py tests/test_hip_codec.py -m gpu -x -q > gpurun_out/r05_model_tests.log 2>&1
echo "model tests rc=$?"; tail -25 gpurun_out/r05_model_tests.log
for t in "-" "-"; do
  python bench.py --config roi --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('roi', round(d['ms_per_step'],1), 'ms')" | tee -a gpurun_out/r05_roi_after_rule.log
done
bash tools/debug/ab_env.sh "-" "STEM_ENGINE_TRANSPOSED_F16X3=0" 2>&1 | tee gpurun_out/r05_ab_transposed.log
