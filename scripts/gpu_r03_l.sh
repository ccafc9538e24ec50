#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/r03l_*.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "bf16 or large_shape" 2>&1 | tail -3 > gpurun_out/r03l_tests.log
timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep -v "amdgpu\|RASTER" > gpurun_out/r03l_bf16p.txt
DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_abl_NOEPI.so timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "amdgpu\|RASTER\|round 2" > gpurun_out/r03l_bf16p_noepi.txt
timeout 600 python bench.py --config M --steps 10 --warmup 3 > gpurun_out/r03l_bench_M.json 2> gpurun_out/r03l_bench_M.err
timeout 600 python bench.py --config L --steps 10 --warmup 3 > gpurun_out/r03l_bench_L.json 2> gpurun_out/r03l_bench_L.err
tail -3 gpurun_out/r03l_tests.log; cat gpurun_out/r03l_bf16p.txt; echo NOEPI; cat gpurun_out/r03l_bf16p_noepi.txt
head -c 260 gpurun_out/r03l_bench_M.json; echo; head -c 260 gpurun_out/r03l_bench_L.json
