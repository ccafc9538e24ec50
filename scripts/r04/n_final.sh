#!/bin/bash
# round 4, batch N: end-of-round records on HEAD - GPU suite (timed), default bench line, rocprofv3 kernel stats of bench.py,
# step censuses T / M / L(16), bf16 GEMM forms; DFT channel-chunk choice at DPOT-L batch 16
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_final
for cc in; do
  E=""; [ $cc != 0 ] && E="DPOT_DFT_CC=$cc"
  env $E timeout 300 python bench.py --config L --brief --steps 8 --warmup 3 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('L DFT_CC=$cc', d['ms_per_step'], d['value'])" >> ${O}_dft_cc.txt
done
( time timeout 2700 python -m pytest tests -m gpu -q ) 2>&1 | grep -v amdgpu.ids | tail -10 > ${O}_gpu_tests.txt
( time timeout 600 python bench.py ) > ${O}_bench.json 2> ${O}_bench.err
bash scripts/gpu_prof.sh r04_final_prof --no-alt --no-pipeline --no-other-configs > /dev/null 2>&1
bash scripts/gpu_census_M.sh T f32 > /dev/null 2>&1; cp gpurun_out/censusT.txt ${O}_census_T.txt
DPOT_GEMM_PRECISION=auto bash scripts/gpu_census_M.sh M bf16 > /dev/null 2>&1; cp gpurun_out/censusM.txt ${O}_census_M.txt
DPOT_GEMM_PRECISION=auto CENSUS_BATCH=16 bash scripts/gpu_census_M.sh L bf16 > /dev/null 2>&1; cp gpurun_out/censusL.txt ${O}_census_L.txt
timeout 600 python scripts/bf16p_train_bench.py S M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2" > ${O}_bf16p_train_bench.txt
cat ${O}_dft_cc.txt ${O}_gpu_tests.txt; head -c 300 ${O}_bench.json; echo; head -6 gpurun_out/r04_final_prof.stats.txt; head -5 ${O}_census_M.txt; head -4 ${O}_census_L.txt
