#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
F="amdgpu\|RASTER\|round 2"
{ for r in 0 2; do echo "== DPOT_BF16P_RASTER=$r"; DPOT_BF16P_RASTER=$r timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F" | grep "fc2 fwd\|fc1 dgrad"; done
for r in 0 2 0 2; do DPOT_BF16P_RASTER=$r timeout 600 python bench.py --config M --steps 10 --warmup 3 2>/dev/null | head -c 200; echo " raster=$r"; done
for r in 0 2; do DPOT_BF16P_RASTER=$r timeout 600 python bench.py --config S --steps 10 --warmup 3 2>/dev/null | head -c 200; echo " raster=$r"; done; } > gpurun_out/r03an.txt
cat gpurun_out/r03an.txt
