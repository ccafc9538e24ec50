#!/bin/bash
# one-launch AFNO layer BACKWARD: parity (op level, model level with the layer forced on), layer A/B, train-step A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout=600 -k "afno_layer" > gpurun_out/r05_g_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r05_g_pytest.log
tail -25 gpurun_out/r05_g_pytest.log
DPOT_AFNO_LAYER=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py -m gpu -q -x --timeout=600 \
  -k "(test_gpu_model and not baseline_configs_forward) or test_full_model_gradients_vs_oracle and (TINY-32 or SMALL-1 or MEDIUM-1) or one_launch or test_vs_reference_golden and (SMALL-32 or MEDIUM-32)" > gpurun_out/r05_g_pytest2.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r05_g_pytest2.log
tail -12 gpurun_out/r05_g_pytest2.log
timeout 300 python scripts/afno_layer_bwd_bench.py > gpurun_out/r05_f4_bwd_fused_vs_launches.txt 2>&1
cat gpurun_out/r05_f4_bwd_fused_vs_launches.txt
rm -f gpurun_out/r05_f4_bwd_step_ab.txt
for cfg in S M; do
  for v in 0 1 0 1; do
    echo -n "config $cfg DPOT_AFNO_LAYER_BWD=$v: " >> gpurun_out/r05_f4_bwd_step_ab.txt
    DPOT_AFNO_LAYER_BWD=$v timeout 300 python bench.py --config $cfg --brief --skip-cpu-baseline --no-other-configs --no-alt --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms/step', d['value'], d['unit'])" >> gpurun_out/r05_f4_bwd_step_ab.txt 2>&1
  done
done
for v in 0 1; do
  echo -n "config T DPOT_AFNO_LAYER=1 DPOT_AFNO_LAYER_BWD=$v: " >> gpurun_out/r05_f4_bwd_step_ab.txt
  DPOT_AFNO_LAYER=1 DPOT_AFNO_LAYER_BWD=$v timeout 300 python bench.py --config T --brief --skip-cpu-baseline --no-other-configs --no-alt --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms/step', d['value'], d['unit'])" >> gpurun_out/r05_f4_bwd_step_ab.txt 2>&1
done
cat gpurun_out/r05_f4_bwd_step_ab.txt
