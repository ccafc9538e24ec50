#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
DPOT_BF16P_RASTER=0 timeout 900 python scripts/bf16p_train_bench.py > gpurun_out/r03b_bf16p_raster0.txt 2>&1
DPOT_BF16P_RASTER=1 timeout 900 python scripts/bf16p_train_bench.py > gpurun_out/r03b_bf16p_raster1.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_sizes.py tests/test_gpu_ops.py -m gpu -x -q -k "bf16" 2>&1 | tail -25 > gpurun_out/r03b_tests.log
for c in S M L; do timeout 600 python bench.py --config $c --steps 10 --warmup 3 > gpurun_out/r03b_bench_$c.json 2> gpurun_out/r03b_bench_$c.err; done
cat gpurun_out/r03b_bf16p_raster0.txt gpurun_out/r03b_bf16p_raster1.txt; tail -8 gpurun_out/r03b_tests.log
for c in S M L; do head -c 300 gpurun_out/r03b_bench_$c.json; echo; tail -2 gpurun_out/r03b_bench_$c.err; done
