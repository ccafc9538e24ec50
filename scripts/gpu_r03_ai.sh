#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{ timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
DPOT_BENCH_DEBUG_GLOO=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 2>&1 | tail -2 | cut -c1-600
DPOT_BENCH_DEBUG_GLOO=1 timeout 900 python bench.py --gpus 2 --config M --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-600
DPOT_BENCH_DEBUG_GLOO=1 timeout 1200 python bench.py --gpus 2 --config L20 --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-600
} > gpurun_out/r03ai.txt 2>&1
cat gpurun_out/r03ai.txt
