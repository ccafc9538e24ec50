#!/bin/bash
# round 4, first GPU batch: new parity tests, full suite, default bench line incl. other_configs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sizes.py -m gpu -q -s -k "large_batch16 or bf16_recompute" 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r04_a_newtests.log
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_data.py tests/test_gpu_train2.py -m gpu -q -s -k "large_group_mean or fused_kernels_vs_separate or wrongly_shaped or graph_replay_raises" 2>&1 | grep -v amdgpu.ids | tail -12 >> gpurun_out/r04_a_newtests.log
( time timeout 600 python bench.py ) > gpurun_out/r04_a_bench.json 2> gpurun_out/r04_a_bench.err
( time timeout 2400 python -m pytest tests/test_gpu_optout.py -m gpu -q -s ) 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r04_a_optout.log
cat gpurun_out/r04_a_newtests.log
head -c 600 gpurun_out/r04_a_bench.json; echo; tail -5 gpurun_out/r04_a_bench.err
tail -16 gpurun_out/r04_a_optout.log
