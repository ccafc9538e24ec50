#!/bin/bash
# round 4, last batch: GPU suite on HEAD (timed) + rocprofv3 kernel stats of `bench.py --config M` / `--config L`
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_last
( time timeout 2700 python -m pytest tests -m gpu -q --durations=12 ) 2>&1 | grep -v amdgpu.ids | tail -24 > ${O}_gpu_tests.txt
bash scripts/gpu_prof.sh r04_last_prof_M --config M --no-alt > /dev/null 2>&1
bash scripts/gpu_prof.sh r04_last_prof_L --config L --no-alt > /dev/null 2>&1
tail -22 ${O}_gpu_tests.txt; head -8 gpurun_out/r04_last_prof_M.stats.txt | cut -c1-150; head -6 gpurun_out/r04_last_prof_L.stats.txt | cut -c1-150
