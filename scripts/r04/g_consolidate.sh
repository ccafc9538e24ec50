#!/bin/bash
# round 4, batch G: consolidation on HEAD - full GPU suite (timed), default bench line, rocprofv3 kernel stats of bench.py,
# step censuses T / M / L(16), PMC passes for the mixer and the bf16 GEMM forms
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_g
( time timeout 2700 python -m pytest tests -m gpu -q ) 2>&1 | grep -v amdgpu.ids | tail -12 > ${O}_gpu_tests.txt
bash scripts/r04/pmc_mixer.sh > ${O}_pmc_mixer.log 2>&1
bash scripts/r04/pmc_bf16p.sh M > ${O}_pmc_bf16p.log 2>&1
( time timeout 600 python bench.py ) > ${O}_bench.json 2> ${O}_bench.err
bash scripts/gpu_prof.sh r04_g_prof --no-alt --no-pipeline --no-other-configs > /dev/null 2>&1
bash scripts/gpu_census_M.sh T f32 > /dev/null 2>&1; cp gpurun_out/censusT.txt ${O}_census_T.txt
bash scripts/gpu_census_M.sh M bf16 > /dev/null 2>&1; cp gpurun_out/censusM.txt ${O}_census_M.txt
CENSUS_BATCH=16 bash scripts/gpu_census_M.sh L bf16 > /dev/null 2>&1; cp gpurun_out/censusL.txt ${O}_census_L.txt
timeout 600 python scripts/bf16p_train_bench.py S M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2" > ${O}_bf16p_train_bench.txt
cat ${O}_gpu_tests.txt; head -c 400 ${O}_bench.json; echo; tail -4 ${O}_bench.err; head -8 gpurun_out/r04_g_prof.stats.txt; head -6 ${O}_census_M.txt; head -6 ${O}_census_L.txt; tail -20 ${O}_pmc_bf16p.log
