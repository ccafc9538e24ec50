#!/bin/bash
# round 4, batch N: end-of-round records on HEAD - GPU suite (timed), default bench line, rocprofv3 kernel stats of bench.py,
# step censuses T / M / L(16) (M / L in the bench's mode: bf16 channel MLP, gemm precision auto), 2-process gloo dry runs, bf16 GEMM forms
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_final
( time timeout 2700 python -m pytest tests -m gpu -q ) 2>&1 | grep -v amdgpu.ids | tail -10 > ${O}_gpu_tests.txt
( time timeout 600 python bench.py ) > ${O}_bench.json 2> ${O}_bench.err
bash scripts/gpu_prof.sh r04_final_prof --no-alt --no-pipeline --no-other-configs > /dev/null 2>&1
bash scripts/gpu_census_M.sh T f32 > /dev/null 2>&1; cp gpurun_out/censusT.txt ${O}_census_T.txt
DPOT_GEMM_PRECISION=auto bash scripts/gpu_census_M.sh M bf16 > /dev/null 2>&1; cp gpurun_out/censusM.txt ${O}_census_M.txt
DPOT_GEMM_PRECISION=auto CENSUS_BATCH=16 bash scripts/gpu_census_M.sh L bf16 > /dev/null 2>&1; cp gpurun_out/censusL.txt ${O}_census_L.txt
DPOT_BENCH_DEBUG_GLOO=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 > ${O}_gloo2_T.json 2> ${O}_gloo2_T.err
DPOT_BENCH_DEBUG_GLOO=1 timeout 900 python bench.py --gpus 2 --config L20 --batch 1 --steps 2 --warmup 1 > ${O}_gloo2_L20.json 2> ${O}_gloo2_L20.err
timeout 600 python scripts/bf16p_train_bench.py S M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2" > ${O}_bf16p_train_bench.txt
cat ${O}_gpu_tests.txt; head -c 300 ${O}_bench.json; echo; head -6 gpurun_out/r04_final_prof.stats.txt; head -5 ${O}_census_M.txt; head -4 ${O}_census_L.txt
