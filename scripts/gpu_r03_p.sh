#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "wgrad2" 2>&1 | tail -3 > gpurun_out/r03p_tests.log
timeout 1500 python -m pytest tests/test_gpu_sizes.py -m gpu -x -q -k "LARGE" 2>&1 | tail -8 >> gpurun_out/r03p_tests.log
bash scripts/gpu_census_M.sh L bf16 > /dev/null 2>&1; cp gpurun_out/censusL.txt gpurun_out/r03p_census_L.txt
timeout 900 python bench.py --config L --steps 10 --warmup 3 > gpurun_out/r03p_bench_L.json 2> gpurun_out/r03p_bench_L.err
cat gpurun_out/r03p_tests.log; head -14 gpurun_out/r03p_census_L.txt; head -c 250 gpurun_out/r03p_bench_L.json
