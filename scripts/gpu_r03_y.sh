#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py tests/test_gpu_train2.py -m gpu -q -k "adam or golden or oracle or replay" 2>&1 | tail -4 > gpurun_out/r03y_tests.log
for c in T M L; do timeout 600 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | head -c 220; echo; done > gpurun_out/r03y_bench.txt
cat gpurun_out/r03y_tests.log gpurun_out/r03y_bench.txt
