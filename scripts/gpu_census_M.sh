mkdir -p gpurun_out; R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/census${1:-M} -o r -- python $R/scripts/census_config.py ${1:-M} ${2:-bf16} $CENSUS_GEMM > $R/gpurun_out/census${1:-M}.log 2>&1
cd $R; python scripts/step_census.py gpurun_out/census${1:-M}/r_results.db $3 > gpurun_out/census${1:-M}.txt 2>&1; rm -rf gpurun_out/census${1:-M}; head -40 gpurun_out/census${1:-M}.txt
