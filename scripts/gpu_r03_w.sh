#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
F="amdgpu\|RASTER\|round 2"
{
echo "== base"; timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
echo "== non-temporal pack stores"; DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_nt.so timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
echo "== base"; timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "$F"
} > gpurun_out/r03w_bf16p.txt
for v in "" nt; do
L=""; [ -n "$v" ] && L=dpot_amd/lib/variants/libdpot_hip_$v.so
for c in M L; do DPOT_HIP_LIB=$L timeout 600 python bench.py --config $c --steps 6 --warmup 2 2>/dev/null | head -c 200; echo; done
done > gpurun_out/r03w_bench.txt
cat gpurun_out/r03w_bf16p.txt gpurun_out/r03w_bench.txt
