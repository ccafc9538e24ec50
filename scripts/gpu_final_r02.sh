#!/bin/bash
# end-of-round measurement batch: default bench line, eager kernel stats + census, BASELINE configs
mkdir -p gpurun_out
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tail -1 gpurun_out/final_bench.json | cut -c1-300
bash scripts/gpu_prof.sh final_prof --no-alt > /dev/null 2>&1
bash scripts/gpu_census.sh > /dev/null 2>&1
python scripts/gpu_configs2.py T S M L 2>/dev/null | grep '^{' > gpurun_out/final_configs.jsonl
cat gpurun_out/final_configs.jsonl | cut -c1-200
