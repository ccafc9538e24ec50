#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_l
for rep in 1 2; do for v in base gelu_pk; do
  L=""; [ $v != base ] && L="DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_$v.so"
  echo "== $v" >> ${O}_bench.txt
  env $L timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep "fc1 fwd, bf16\|^[ML] " >> ${O}_bench.txt
done; done
( DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_gelu_pk.so timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sizes.py -m gpu -q -x -k "bf16 and not x6 and not LARGE" ) 2>&1 | grep -v amdgpu.ids | tail -3 > ${O}_tests.log
for v in base gelu_pk; do
  L=""; [ $v != base ] && L="DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_$v.so"
  env $L timeout 300 python bench.py --config M --brief --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('M $v', d['ms_per_step'], d['value'])" >> ${O}_step.txt
done
cat ${O}_bench.txt ${O}_tests.log ${O}_step.txt
