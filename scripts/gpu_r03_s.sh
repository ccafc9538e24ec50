#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "bf16 or pack or large_shape" 2>&1 | tail -3 > gpurun_out/r03s_tests.log
timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2" > gpurun_out/r03s_bf16p.txt
timeout 600 python bench.py --config M --steps 10 --warmup 3 > gpurun_out/r03s_bench_M.json 2> gpurun_out/r03s_bench_M.err
cat gpurun_out/r03s_tests.log gpurun_out/r03s_bf16p.txt; head -c 250 gpurun_out/r03s_bench_M.json
