#!/bin/bash
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 -p no:cacheprovider ) > gpurun_out/r05_gpu_tests.txt 2>&1
tail -30 gpurun_out/r05_gpu_tests.txt
timeout 600 python scripts/dp_host_time.py > gpurun_out/r05_dp_host_time.txt 2>&1; cat gpurun_out/r05_dp_host_time.txt
