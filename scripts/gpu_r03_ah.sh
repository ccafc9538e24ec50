#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for c in M S; do for d in 0 5 0 5; do DPOT_AFNO_MLP_RT=$d timeout 900 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | head -c 200; echo " RT=$d"; done; done > gpurun_out/r03ah_bench.txt
cat gpurun_out/r03ah_bench.txt
