#!/bin/bash
# eager train-step census: kernels of one step (between two adam launches)
mkdir -p gpurun_out; R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/census -o r -- python $R/bench.py --steps 4 --warmup 2 --no-graph --skip-cpu-baseline --no-alt "$@" > $R/gpurun_out/census.log 2>&1
cd $R; python scripts/step_census.py gpurun_out/census/r_results.db --seq > gpurun_out/census.txt 2>&1; rm -rf gpurun_out/census; cat gpurun_out/census.txt
