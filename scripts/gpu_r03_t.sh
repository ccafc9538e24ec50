#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "bf16 or pack or large_shape" 2>&1 | tail -3 > gpurun_out/r03t_tests.log
F="amdgpu\|RASTER\|round 2"
{
echo "== duo auto (packed launches)"; timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep -v "$F"
echo "== duo never"; DPOT_BF16P_DUO=0 timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
echo "== duo always"; DPOT_BF16P_DUO=2 timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
echo "== duo auto, no priority seed"; DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_duoprio0.so timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
echo "== duo auto, priority on even rounds"; DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_duoprio2.so timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
} > gpurun_out/r03t_bf16p.txt
timeout 600 python bench.py --config M --steps 10 --warmup 3 > gpurun_out/r03t_bench_M.json 2> gpurun_out/r03t_bench_M.err
cat gpurun_out/r03t_tests.log gpurun_out/r03t_bf16p.txt; head -c 250 gpurun_out/r03t_bench_M.json
