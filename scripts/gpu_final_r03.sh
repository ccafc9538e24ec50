#!/bin/bash
# end-of-round measurement batch: full GPU suite, default bench line, eager kernel stats + census, BASELINE configs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r03_final_tests.log
python bench.py > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err
bash scripts/gpu_prof.sh r03_final_prof --no-alt --no-pipeline > /dev/null 2>&1
bash scripts/gpu_census_M.sh T f32 > /dev/null 2>&1; cp gpurun_out/censusT.txt gpurun_out/r03_final_census_T.txt
bash scripts/gpu_census_M.sh M bf16 > /dev/null 2>&1; cp gpurun_out/censusM.txt gpurun_out/r03_final_census_M.txt
bash scripts/gpu_census_M.sh L bf16 > /dev/null 2>&1; cp gpurun_out/censusL.txt gpurun_out/r03_final_census_L.txt
for c in S M L L20; do timeout 900 python bench.py --config $c --steps 10 --warmup 3 > gpurun_out/r03_final_bench_$c.json 2> gpurun_out/r03_final_bench_$c.err; done
timeout 900 python bench.py --config L --batch 24 --steps 6 --warmup 2 > gpurun_out/r03_final_bench_L_B24.json 2> gpurun_out/r03_final_bench_L_B24.err
timeout 300 python scripts/gn_dft_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r03_final_gn_dft_bench.txt
timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2" > gpurun_out/r03_final_bf16p_train_bench.txt
tail -4 gpurun_out/r03_final_tests.log
for f in r03_final_bench r03_final_bench_S r03_final_bench_M r03_final_bench_L r03_final_bench_L20 r03_final_bench_L_B24; do head -c 230 gpurun_out/$f.json | cut -c1-230; echo; done
head -12 gpurun_out/r03_final_prof.stats.txt
