#!/bin/bash
# rocprofv3 kernel trace of a short eager bench; summary -> gpurun_out/<name>.stats.txt
NAME=${1:-prof}
shift
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$NAME -o r -- python $R/bench.py --steps 5 --warmup 2 --no-graph --skip-cpu-baseline "$@" > $R/gpurun_out/$NAME.log 2>&1
cd $R
python scripts/rocpd_stats.py gpurun_out/$NAME/r_results.db 60 > gpurun_out/$NAME.stats.txt 2>&1
python scripts/rocpd_seq.py gpurun_out/$NAME/r_results.db 300 60 > gpurun_out/$NAME.seq.txt 2>&1; rm -rf gpurun_out/$NAME
grep '"metric"' gpurun_out/$NAME.log | head -1
