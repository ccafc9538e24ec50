#!/bin/bash
# round 5, batch a: parity of the one-launch AFNO layer + the A/B timing against the three launches
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout=300 -k "afno_layer_one_launch" > gpurun_out/r05_a_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r05_a_pytest.log
tail -30 gpurun_out/r05_a_pytest.log
timeout 300 python scripts/afno_layer_bench.py > gpurun_out/r05_f4_fused_vs_3launch.txt 2>&1
cat gpurun_out/r05_f4_fused_vs_3launch.txt
