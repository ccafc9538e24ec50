#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
F="amdgpu\|RASTER\|round 2"
{
echo "== duo auto"; timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
for d in 300 500 700; do
echo "== duo auto, second workgroup delayed by nslab x $d ns"; DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_duodelay$d.so timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
done
echo "== no epilogue: duo always"; DPOT_BF16P_DUO=2 DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_duonoepi.so timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
echo "== no epilogue: duo never"; DPOT_BF16P_DUO=0 DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_duonoepi.so timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
} > gpurun_out/r03u_bf16p.txt
for d in 1 0; do
DPOT_BF16P_DUO=$d timeout 600 python bench.py --config M --steps 10 --warmup 3 2>/dev/null | head -c 200; echo
DPOT_BF16P_DUO=$d timeout 600 python bench.py --config L --steps 5 --warmup 2 2>/dev/null | head -c 200; echo
done > gpurun_out/r03u_bench.txt
cat gpurun_out/r03u_bf16p.txt gpurun_out/r03u_bench.txt
