#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout=300 -k "afno_layer_one_launch" > gpurun_out/r05_c_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r05_c_pytest.log
tail -5 gpurun_out/r05_c_pytest.log
timeout 300 python scripts/afno_layer_bench.py > gpurun_out/r05_f4_fused_vs_3launch_v2.txt 2>&1
cat gpurun_out/r05_f4_fused_vs_3launch_v2.txt
DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_aftiming.so timeout 300 python scripts/afno_layer_phases.py > gpurun_out/r05_f4_phases_v2.txt 2>&1
cat gpurun_out/r05_f4_phases_v2.txt
