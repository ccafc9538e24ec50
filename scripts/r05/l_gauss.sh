#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout=600 -k "wgrad or row_form or bf16" > gpurun_out/r05_l_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r05_l_pytest.log; tail -5 gpurun_out/r05_l_pytest.log
O=gpurun_out/r05_gauss_step_ab.txt; rm -f $O
ab() {
  echo -n "config $1 $2: " >> $O
  env $2 timeout 300 python bench.py --config $1 --brief --skip-cpu-baseline --no-other-configs --no-alt --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms/step', d['value'], d['unit'])" >> $O 2>&1
}
for v in 0 1 0 1; do ab T DPOT_AFNO_WGRAD_GAUSS=$v; done
for v in 0 1 0 1; do ab S DPOT_AFNO_WGRAD_GAUSS=$v; done
for v in 0 1 0 1; do ab M DPOT_AFNO_WGRAD_GAUSS=$v; done
cat $O
timeout 300 python scripts/tn_bench.py > gpurun_out/r05_tn_bench_gauss.txt 2>&1; head -40 gpurun_out/r05_tn_bench_gauss.txt
