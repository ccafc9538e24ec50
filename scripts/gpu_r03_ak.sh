#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in "" base "" base; do L=""; [ -n "$v" ] && L=dpot_amd/lib/variants/libdpot_hip_$v.so; DPOT_HIP_LIB=$L timeout 600 python bench.py --steps 40 --warmup 10 2>/dev/null | head -c 190; echo " variant=$v"; done > gpurun_out/r03ak.txt
for v in "" base; do L=""; [ -n "$v" ] && L=dpot_amd/lib/variants/libdpot_hip_$v.so; DPOT_HIP_LIB=$L timeout 600 python bench.py --config S --steps 10 --warmup 3 2>/dev/null | head -c 190; echo " variant=$v"; done >> gpurun_out/r03ak.txt
cat gpurun_out/r03ak.txt
