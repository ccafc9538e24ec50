#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "fused_kernels" 2>&1 | tail -3 > gpurun_out/r03h_tests.log
timeout 300 python scripts/gn_dft_bench.py > gpurun_out/r03h_gn_dft_bench.txt 2>&1
timeout 600 python bench.py --no-alt --no-pipeline --skip-cpu-baseline > gpurun_out/r03h_bench.json 2> gpurun_out/r03h_bench.err
tail -3 gpurun_out/r03h_tests.log; grep -v amdgpu gpurun_out/r03h_gn_dft_bench.txt; head -c 300 gpurun_out/r03h_bench.json
