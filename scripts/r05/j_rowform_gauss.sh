#!/bin/bash
# row-form (transposing-read) bf16 weight gradients + three-product AFNO weight gradients: parity, then step A/Bs on one box
mkdir -p gpurun_out
./scripts/ubench/tr_read_layout > gpurun_out/r05_tr_read_layout.txt 2>&1; cat gpurun_out/r05_tr_read_layout.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout=600 -k "row_form or pair or wgrad or afno" > gpurun_out/r05_j_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r05_j_pytest.log; tail -15 gpurun_out/r05_j_pytest.log
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py -m gpu -q -x --timeout=900 \
  -k "test_vs_reference_golden or test_bf16_channel_mlp or test_full_model_gradients_vs_oracle or (test_gpu_model and not baseline_configs_forward) or block_finalize" > gpurun_out/r05_j_pytest2.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r05_j_pytest2.log; tail -15 gpurun_out/r05_j_pytest2.log
O=gpurun_out/r05_rowform_gauss_step_ab.txt; rm -f $O
ab() {  # cfg, env assignment
  echo -n "config $1 $2: " >> $O
  env $2 timeout 300 python bench.py --config $1 --brief --skip-cpu-baseline --no-other-configs --no-alt --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms/step', d['value'], d['unit'])" >> $O 2>&1
}
for v in 0 1 0 1; do ab M DPOT_BF16P_ROWFORM=$v; done
for v in 0 1; do ab L DPOT_BF16P_ROWFORM=$v; done
for v in 0 1 0 1; do ab T DPOT_AFNO_WGRAD_GAUSS=$v; done
for v in 0 1 0 1; do ab S DPOT_AFNO_WGRAD_GAUSS=$v; done
for v in 0 1; do ab M DPOT_AFNO_WGRAD_GAUSS=$v; done
cat $O
timeout 300 python scripts/tn_bench.py > gpurun_out/r05_tn_bench_gauss.txt 2>&1; head -30 gpurun_out/r05_tn_bench_gauss.txt
