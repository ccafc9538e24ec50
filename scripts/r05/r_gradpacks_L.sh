#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout=600 -k "groupnorm" > gpurun_out/r05_r_pytest.log 2>&1; tail -5 gpurun_out/r05_r_pytest.log
timeout 1200 python -m pytest tests/test_gpu_sizes.py -m gpu -q -x --timeout=900 -k "gradient_packs or test_vs_reference_golden or bf16_recompute" > gpurun_out/r05_r_pytest2.log 2>&1; tail -6 gpurun_out/r05_r_pytest2.log
O=gpurun_out/r05_grad_packs_step_ab_L.txt; rm -f $O
ab() {
  echo -n "config $1 $2: " >> $O
  env $2 timeout 600 python bench.py --config $1 --brief --skip-cpu-baseline --no-other-configs --no-alt --steps $3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms/step', d['value'], d['unit'])" >> $O 2>&1
}
for v in 0 1 0 1; do ab L DPOT_GRAD_PACKS=$v 8; done
for v in 0 1; do ab L20 DPOT_GRAD_PACKS=$v 2; done
cat $O
