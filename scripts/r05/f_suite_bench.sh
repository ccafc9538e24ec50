#!/bin/bash
# full GPU suite with durations (gate time), then the default bench line (driver form)
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=25 -p no:cacheprovider ) > gpurun_out/r05_gpu_tests.txt 2>&1
tail -45 gpurun_out/r05_gpu_tests.txt
( time timeout 600 python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
tail -c 6000 gpurun_out/r05_bench_default.json; tail -5 gpurun_out/r05_bench_default.err
