#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for d in 0 32 16 0; do DPOT_DFT_CC=$d timeout 900 python bench.py --config L --steps 6 --warmup 2 2>/dev/null | head -c 200; echo " DFT_CC=$d"; done > gpurun_out/r03ag_bench.txt
cat gpurun_out/r03ag_bench.txt
