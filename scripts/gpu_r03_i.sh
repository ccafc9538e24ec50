#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "bf16 or large_shape" 2>&1 | tail -3 > gpurun_out/r03i_tests.log
for R in 3 4 5 6; do
  DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_ring$R.so timeout 600 python scripts/bf16p_train_bench.py M L > gpurun_out/r03i_bf16p_ring$R.txt 2>&1
done
timeout 600 python bench.py --config M --steps 10 --warmup 3 > gpurun_out/r03i_bench_M.json 2> gpurun_out/r03i_bench_M.err
DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_ring3.so timeout 600 python bench.py --config M --steps 10 --warmup 3 > gpurun_out/r03i_bench_M_ring3.json 2> gpurun_out/r03i_bench_M_ring3.err
tail -3 gpurun_out/r03i_tests.log
for R in 3 4 5 6; do echo "ring $R"; grep -v amdgpu gpurun_out/r03i_bf16p_ring$R.txt | grep -v "round 2\|inference\|pack_both\|RASTER"; done
head -c 260 gpurun_out/r03i_bench_M.json; echo; head -c 260 gpurun_out/r03i_bench_M_ring3.json
