#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train2.py tests/test_gpu_model.py -m gpu -q -k "adam or golden or oracle or replay" 2>&1 | tail -4 > gpurun_out/r03z_tests.log
{ echo "== float4, contiguous range per workgroup"; python scripts/adam_bench.py 2>&1 | grep adam
echo "== scalar grid-stride (round 2 form)"; DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_adamscalar.so python scripts/adam_bench.py 2>&1 | grep adam; } > gpurun_out/r03z_adam.txt
cat gpurun_out/r03z_tests.log gpurun_out/r03z_adam.txt
