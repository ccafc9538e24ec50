#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "bf16 or pack or large_shape or pair or 192" 2>&1 | tail -5 > gpurun_out/r03ae_tests.log
timeout 1200 python -m pytest tests/test_gpu_sizes.py -m gpu -q -k "bf16" 2>&1 | tail -3 >> gpurun_out/r03ae_tests.log
F="amdgpu\|RASTER\|round 2"
{ echo "== 192-wide tiles (default)"; timeout 600 python scripts/bf16p_train_bench.py L 2>&1 | grep -v "$F"
echo "== DPOT_BF16P_TILE192=0"; DPOT_BF16P_TILE192=0 timeout 600 python scripts/bf16p_train_bench.py L 2>&1 | grep -v "$F"; } > gpurun_out/r03ae_bf16p.txt
for d in 1 0 1 0; do DPOT_BF16P_TILE192=$d timeout 900 python bench.py --config L20 --steps 3 --warmup 1 2>/dev/null | head -c 200; echo " tile192=$d"; done > gpurun_out/r03ae_bench.txt
cat gpurun_out/r03ae_tests.log gpurun_out/r03ae_bf16p.txt gpurun_out/r03ae_bench.txt
