#!/bin/bash
mkdir -p gpurun_out
bash scripts/gpu_census.sh > /dev/null 2>&1; mv gpurun_out/census.txt gpurun_out/r05_census_T_seq.txt; head -5 gpurun_out/r05_census_T_seq.txt
( time timeout 600 python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
tail -c 1500 gpurun_out/r05_bench_default.json; tail -4 gpurun_out/r05_bench_default.err
