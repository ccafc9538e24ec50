#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/r03n_*
V=$PWD/dpot_amd/lib/variants
DPOT_HIP_LIB=$V/libdpot_hip_regstage.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "bf16 or large_shape" 2>&1 | tail -3 > gpurun_out/r03n_tests.log
timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2" > gpurun_out/r03n_base.txt
DPOT_HIP_LIB=$V/libdpot_hip_regstage.so timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2" > gpurun_out/r03n_regstage.txt
DPOT_HIP_LIB=$V/libdpot_hip_regstage_NOEPI.so timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep "fc2 fwd\|fc1 dgrad\|inference\|pair" > gpurun_out/r03n_regstage_noepi.txt
DPOT_HIP_LIB=$V/libdpot_hip_regstage.so timeout 600 python bench.py --config M --steps 10 --warmup 3 > gpurun_out/r03n_bench_M_regstage.json 2> gpurun_out/r03n_bench_M_regstage.err
tail -3 gpurun_out/r03n_tests.log; echo BASE; cat gpurun_out/r03n_base.txt; echo REGSTAGE; cat gpurun_out/r03n_regstage.txt; echo NOEPI; cat gpurun_out/r03n_regstage_noepi.txt; head -c 250 gpurun_out/r03n_bench_M_regstage.json
