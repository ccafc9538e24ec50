#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "bf16 or pack or large_shape or pair" 2>&1 | tail -5 > gpurun_out/r03ad_tests.log
F="amdgpu\|RASTER\|round 2"
{ echo "== 192-wide tiles for the pair launch (default)"; timeout 600 python scripts/bf16p_train_bench.py L 2>&1 | grep -v "$F" | grep "L B\|pair"
echo "== DPOT_BF16P_TILE192=0"; DPOT_BF16P_TILE192=0 timeout 600 python scripts/bf16p_train_bench.py L 2>&1 | grep -v "$F" | grep "L B\|pair"; } > gpurun_out/r03ad_bf16p.txt
for d in 1 0 1 0; do DPOT_BF16P_TILE192=$d timeout 600 python bench.py --config L --steps 6 --warmup 2 2>/dev/null | head -c 200; echo " tile192=$d"; done > gpurun_out/r03ad_bench.txt
cat gpurun_out/r03ad_tests.log gpurun_out/r03ad_bf16p.txt gpurun_out/r03ad_bench.txt
