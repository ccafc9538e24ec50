#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "fused_kernels or 256_tile or derivative or pack_both or bf16" 2>&1 | tail -25 > gpurun_out/r03c_tests_ops.log
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_train2.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03c_tests_model.log
DPOT_BF16Q=0 timeout 600 python scripts/bf16p_train_bench.py M L > gpurun_out/r03c_bf16p_q0.txt 2>&1
DPOT_BF16Q=1 timeout 600 python scripts/bf16p_train_bench.py M L > gpurun_out/r03c_bf16p_q1.txt 2>&1
timeout 600 python bench.py --no-alt --no-pipeline --skip-cpu-baseline > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err
DPOT_GN_DFT=0 timeout 600 python bench.py --no-alt --no-pipeline --skip-cpu-baseline > gpurun_out/r03c_bench_nogd.json 2> gpurun_out/r03c_bench_nogd.err
timeout 600 python bench.py --config M --steps 10 --warmup 3 > gpurun_out/r03c_bench_M.json 2> gpurun_out/r03c_bench_M.err
DPOT_BF16Q=0 timeout 600 python bench.py --config M --steps 10 --warmup 3 > gpurun_out/r03c_bench_M_q0.json 2> gpurun_out/r03c_bench_M_q0.err
bash scripts/gpu_census_M.sh M > /dev/null 2>&1; cp gpurun_out/censusM.txt gpurun_out/r03c_census_M.txt
tail -6 gpurun_out/r03c_tests_ops.log; tail -6 gpurun_out/r03c_tests_model.log; cat gpurun_out/r03c_bf16p_q0.txt gpurun_out/r03c_bf16p_q1.txt
for f in r03c_bench r03c_bench_nogd r03c_bench_M r03c_bench_M_q0; do head -c 260 gpurun_out/$f.json; echo; tail -2 gpurun_out/$f.err; done
head -30 gpurun_out/r03c_census_M.txt
