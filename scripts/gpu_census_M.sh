mkdir -p gpurun_out; R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/censusM -o r -- python $R/scripts/census_config.py M bf16 > $R/gpurun_out/censusM.log 2>&1
cd $R; python scripts/step_census.py gpurun_out/censusM/r_results.db > gpurun_out/censusM.txt 2>&1; rm -rf gpurun_out/censusM; head -40 gpurun_out/censusM.txt
