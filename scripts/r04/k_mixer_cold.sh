#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_k
timeout 600 python scripts/r04/mixer_cold.py 2>&1 | grep -v amdgpu > ${O}_mixer_cold.txt
( time timeout 2700 python -m pytest tests -m gpu -q --durations=45 ) 2>&1 | grep -v amdgpu.ids | tail -64 > ${O}_gpu_tests_durations.txt
cat ${O}_mixer_cold.txt; tail -60 ${O}_gpu_tests_durations.txt
