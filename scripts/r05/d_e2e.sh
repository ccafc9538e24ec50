#!/bin/bash
# end-to-end: parity subset with the one-launch AFNO layer forced on, then the train step A/B on one box
mkdir -p gpurun_out
DPOT_AFNO_LAYER=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py -m gpu -q -x --timeout=600 \
  -k "(test_gpu_model and not baseline_configs_forward) or test_full_model_gradients_vs_oracle and (TINY-32 or SMALL-1 or MEDIUM-1) or test_bf16_channel_mlp_mode_vs_oracle and (MEDIUM-1 or SMALL-32)" > gpurun_out/r05_d_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r05_d_pytest.log
tail -8 gpurun_out/r05_d_pytest.log
for cfg in S M T; do
  for v in 0 1 0 1; do
    echo "config $cfg DPOT_AFNO_LAYER=$v" >> gpurun_out/r05_d_ab.txt
    DPOT_AFNO_LAYER=$v timeout 300 python bench.py --config $cfg --brief --skip-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])" >> gpurun_out/r05_d_ab.txt 2>&1
  done
done
cat gpurun_out/r05_d_ab.txt
