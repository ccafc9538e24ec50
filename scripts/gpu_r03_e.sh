#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash scripts/gpu_census_M.sh M bf16 --seq > /dev/null 2>&1; cp gpurun_out/censusM.txt gpurun_out/r03e_census_M_seq.txt
DPOT_BF16P_RASTER=0 bash scripts/gpu_census_M.sh M bf16 --seq > /dev/null 2>&1; cp gpurun_out/censusM.txt gpurun_out/r03e_census_M_seq_raster0.txt
timeout 900 python -m pytest tests/test_gpu_data.py tests/test_gpu_sizes.py -m gpu -x -q -k "data or golden or MEDIUM or SMALL" 2>&1 | tail -8 > gpurun_out/r03e_tests.log
timeout 600 python bench.py --config M --steps 10 --warmup 3 > gpurun_out/r03e_bench_M.json 2> gpurun_out/r03e_bench_M.err
grep "gemm_bf16p_kernel" gpurun_out/r03e_census_M_seq.txt | head -60
tail -5 gpurun_out/r03e_tests.log; head -c 300 gpurun_out/r03e_bench_M.json
