#!/bin/bash
# round 4, batch B: new tests (128-point FFT, 1024^2 model, per-model precision, grouped opt-out), B-direct bf16 GEMM
# correctness + timing, 2-process host-time dry run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_b
( time timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sizes.py tests/test_gpu_train2.py -m gpu -q -s -k "rfft2_irfft2_vs_torch or 1024_resolution or per_model_attribute or graph_replay_raises" ) 2>&1 | grep -v amdgpu.ids | tail -8 > ${O}_newtests.log
( time timeout 1500 python -m pytest tests/test_gpu_optout.py -m gpu -q -s ) 2>&1 | grep -v amdgpu.ids | tail -14 > ${O}_optout.log
for bd in 1; do
  ( DPOT_BF16P_BD=$bd timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sizes.py -m gpu -q -x -k "bf16 and not x6 and not LARGE" ) 2>&1 | grep -v amdgpu.ids | tail -6 > ${O}_bd${bd}_tests.log
done
for bd in 0 1 2; do
  DPOT_BF16P_BD=$bd timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2" > ${O}_bd${bd}_bench.txt
done
for bd in 0 1 2; do
  DPOT_BF16P_BD=$bd timeout 300 python bench.py --config M --brief --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('M BD=$bd', d['ms_per_step'], d['value'])" >> ${O}_bd_step.txt
  DPOT_BF16P_BD=$bd timeout 300 python bench.py --config L --brief --steps 8 --warmup 3 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('L BD=$bd', d['ms_per_step'], d['value'])" >> ${O}_bd_step.txt
done
DPOT_BENCH_DEBUG_GLOO=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > ${O}_gloo2.json 2> ${O}_gloo2.err
cat ${O}_newtests.log ${O}_optout.log ${O}_bd1_tests.log
paste -d'|' ${O}_bd0_bench.txt ${O}_bd1_bench.txt | cut -c1-200
cat ${O}_bd2_bench.txt ${O}_bd_step.txt
head -c 1500 ${O}_gloo2.json; tail -3 ${O}_gloo2.err
