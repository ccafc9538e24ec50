#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for b in 16 24 32; do timeout 900 python bench.py --config L --batch $b --steps 4 --warmup 2 2>gpurun_out/r03x_err_$b.txt | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['config'].get('global_batch'), d['value'], d['ms_per_step'], d.get('peak_mem_GB'), d['config'].get('workload'))
"; done > gpurun_out/r03x_L.txt
cat gpurun_out/r03x_L.txt; tail -3 gpurun_out/r03x_err_32.txt
