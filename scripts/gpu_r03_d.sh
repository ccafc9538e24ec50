#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r03d_tests_ops.log
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03d_tests_model.log
timeout 600 python bench.py --no-alt --no-pipeline --skip-cpu-baseline > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench.err
timeout 600 python bench.py --config M --steps 10 --warmup 3 > gpurun_out/r03d_bench_M.json 2> gpurun_out/r03d_bench_M.err
timeout 600 python bench.py --config S --steps 10 --warmup 3 > gpurun_out/r03d_bench_S.json 2> gpurun_out/r03d_bench_S.err
bash scripts/gpu_census_M.sh M > /dev/null 2>&1; cp gpurun_out/censusM.txt gpurun_out/r03d_census_M.txt
bash scripts/gpu_census_M.sh T f32 > /dev/null 2>&1; cp gpurun_out/censusT.txt gpurun_out/r03d_census_T.txt
tail -12 gpurun_out/r03d_tests_ops.log; tail -4 gpurun_out/r03d_tests_model.log
for f in r03d_bench r03d_bench_M r03d_bench_S; do head -c 260 gpurun_out/$f.json; echo; tail -2 gpurun_out/$f.err; done
head -24 gpurun_out/r03d_census_M.txt; head -40 gpurun_out/r03d_census_T.txt
