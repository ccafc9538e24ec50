#!/bin/bash
# round 3, first GPU pass: full GPU suite + default bench + the self-launched 2-rank dry run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/r03a_tests.log
echo "pytest rc=$?" >> gpurun_out/r03a_tests.log
timeout 600 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
DPOT_BENCH_DEBUG_GLOO=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r03a_bench_g2.json 2> gpurun_out/r03a_bench_g2.err
DPOT_BENCH_DEBUG_GLOO=1 timeout 900 python bench.py --gpus 2 --config L20 --batch 1 --steps 2 --warmup 1 > gpurun_out/r03a_bench_g2_L20.json 2> gpurun_out/r03a_bench_g2_L20.err
tail -5 gpurun_out/r03a_tests.log; cat gpurun_out/r03a_bench.json | head -c 600; echo; tail -3 gpurun_out/r03a_bench_g2.err; head -c 400 gpurun_out/r03a_bench_g2.json; echo; tail -3 gpurun_out/r03a_bench_g2_L20.err; head -c 400 gpurun_out/r03a_bench_g2_L20.json
