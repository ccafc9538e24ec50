#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout=600 -k "groupnorm" > gpurun_out/r05_q_pytest.log 2>&1; tail -5 gpurun_out/r05_q_pytest.log
timeout 900 python -m pytest tests/test_gpu_sizes.py tests/test_gpu_train2.py -m gpu -q -x --timeout=600 -k "gradient_packs or test_vs_reference_golden and (SMALL or MEDIUM) or test_bf16_channel_mlp or segmented or two_process" > gpurun_out/r05_q_pytest2.log 2>&1; tail -6 gpurun_out/r05_q_pytest2.log
O=gpurun_out/r05_grad_packs_step_ab.txt; rm -f $O
ab() {
  echo -n "config $1 $2: " >> $O
  env $2 timeout 300 python bench.py --config $1 --brief --skip-cpu-baseline --no-other-configs --no-alt --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms/step', d['value'], d['unit'])" >> $O 2>&1
}
for v in 0 1 0 1; do ab S DPOT_GRAD_PACKS=$v; done
for v in 0 1 0 1; do ab M DPOT_GRAD_PACKS=$v; done
cat $O
