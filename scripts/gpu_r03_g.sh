#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_sizes.py -m gpu -x -q -k "afno or mixer or block or model or golden or TINY or SMALL or rollout" 2>&1 | tail -6 > gpurun_out/r03g_tests.log
timeout 300 python scripts/afno_mlp_bench.py > gpurun_out/r03g_afno_bench.txt 2>&1
timeout 600 python bench.py --no-alt --no-pipeline --skip-cpu-baseline > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err
bash scripts/gpu_pmc_mixer.sh > gpurun_out/r03g_pmc.log 2>&1
tail -5 gpurun_out/r03g_tests.log; grep -v amdgpu gpurun_out/r03g_afno_bench.txt; head -c 300 gpurun_out/r03g_bench.json; echo; cat gpurun_out/pmc_mixer_r03.json
