#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for d in 1 2 1 2; do DPOT_BF16P_DUO=$d timeout 600 python bench.py --config L --steps 6 --warmup 2 2>/dev/null | head -c 200; echo " duo=$d"; done > gpurun_out/r03ac.txt
cat gpurun_out/r03ac.txt
