#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05_gauss_step_ab.txt; rm -f $O
ab() {
  echo -n "config $1 $2: " >> $O
  env $2 timeout 300 python bench.py --config $1 --brief --skip-cpu-baseline --no-other-configs --no-alt --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms/step', d['value'], d['unit'])" >> $O 2>&1
}
for v in 0 1 0 1; do ab T DPOT_AFNO_WGRAD_GAUSS=$v; done
for v in 0 1 0 1; do ab S DPOT_AFNO_WGRAD_GAUSS=$v; done
for v in 0 1; do ab M DPOT_AFNO_WGRAD_GAUSS=$v; done
cat $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=8 -p no:cacheprovider ) > gpurun_out/r05_gpu_tests.txt 2>&1
tail -22 gpurun_out/r05_gpu_tests.txt
