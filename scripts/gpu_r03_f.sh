#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "afno or mixer or block or model or golden or large_shape" 2>&1 | tail -6 > gpurun_out/r03f_tests.log
timeout 300 python scripts/afno_mlp_bench.py > gpurun_out/r03f_afno_bench.txt 2>&1
timeout 600 python bench.py --no-alt --no-pipeline --skip-cpu-baseline > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err
DPOT_BF16P_RASTER=1 timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "large_shape or bf16" 2>&1 | tail -3 >> gpurun_out/r03f_tests.log
tail -8 gpurun_out/r03f_tests.log; cat gpurun_out/r03f_afno_bench.txt | grep -v amdgpu; head -c 300 gpurun_out/r03f_bench.json
