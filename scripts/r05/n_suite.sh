#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout=600 -k "rfft2 or dft_adjoints" > gpurun_out/r05_n_fft.log 2>&1; tail -6 gpurun_out/r05_n_fft.log
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=10 -p no:cacheprovider ) > gpurun_out/r05_gpu_tests.txt 2>&1
tail -24 gpurun_out/r05_gpu_tests.txt
