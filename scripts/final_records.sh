#!/bin/bash
# end-of-round records on HEAD (R = round tag, default r06): GPU suite, default bench line (+ verbose form), rocprofv3 kernel
# stats of the same bench command, step censuses (T / S / M / L), mixer counters, 2-process gloo dry run of the N > 1 path
R=${1:-r06}
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=8 -p no:cacheprovider ) > gpurun_out/${R}_final_gpu_tests.txt 2>&1
tail -16 gpurun_out/${R}_final_gpu_tests.txt
( time timeout 900 python bench.py --full-json gpurun_out/${R}_final_bench_full.json ) > gpurun_out/${R}_final_bench.json 2> gpurun_out/${R}_final_bench.err
wc -c gpurun_out/${R}_final_bench.json; tail -c 1500 gpurun_out/${R}_final_bench.json; grep -v "bench-full" gpurun_out/${R}_final_bench.err | tail -4
grep -v "bench-full" gpurun_out/${R}_final_bench.err > gpurun_out/${R}_final_bench.err.txt; rm -f gpurun_out/${R}_final_bench.err
bash scripts/gpu_prof.sh ${R}prof --no-other-configs --no-alt --no-pipeline > /dev/null 2>&1; mv gpurun_out/${R}prof.stats.txt gpurun_out/${R}_final_bench_kernel_stats.txt; rm -f gpurun_out/${R}prof.seq.txt gpurun_out/${R}prof.log; head -14 gpurun_out/${R}_final_bench_kernel_stats.txt
bash scripts/gpu_census.sh > /dev/null 2>&1; mv gpurun_out/census.txt gpurun_out/${R}_final_census_T.txt
for c in S M; do CENSUS_GEMM=auto bash scripts/gpu_census_M.sh $c bf16 > /dev/null 2>&1; mv gpurun_out/census$c.txt gpurun_out/${R}_final_census_$c.txt; done
CENSUS_GEMM=auto CENSUS_BATCH=16 bash scripts/gpu_census_M.sh L bf16 > /dev/null 2>&1; mv gpurun_out/censusL.txt gpurun_out/${R}_final_census_L.txt
head -12 gpurun_out/${R}_final_census_M.txt
DPOT_BENCH_DEBUG_GLOO=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 2> gpurun_out/${R}_final_gloo2_T.err | grep "^{" | tail -1 > gpurun_out/${R}_final_gloo2_T.json; tail -c 900 gpurun_out/${R}_final_gloo2_T.json; grep -v "bench-full" gpurun_out/${R}_final_gloo2_T.err | tail -3
rm -f gpurun_out/*.log
