#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "bf16 or pack or large_shape" 2>&1 | tail -15 > gpurun_out/r03v_tests.log
F="amdgpu\|RASTER\|round 2"
{
echo "== direct epilogue, duo auto"; timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep -v "$F"
echo "== direct epilogue, duo never"; DPOT_BF16P_DUO=0 timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
echo "== no activation math"; DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_noact.so timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
echo "== no pack stores"; DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_nostore.so timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
} > gpurun_out/r03v_bf16p.txt
for c in M L; do timeout 600 python bench.py --config $c --steps 6 --warmup 2 2>/dev/null | head -c 200; echo; done > gpurun_out/r03v_bench.txt
cat gpurun_out/r03v_tests.log gpurun_out/r03v_bf16p.txt gpurun_out/r03v_bench.txt
