#!/bin/bash
# round 4, batch C: B-direct bf16 GEMM with four 128 x 64 waves (two workgroups per CU) vs eight 128 x 32 waves; Adam / sumsq
# rewrite checks; DPOT-Tiny dispatch sequence
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_c
( DPOT_BF16P_BD=1 DPOT_BF16P_BD_CPW=2 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sizes.py -m gpu -q -x -k "bf16 and not x6 and not LARGE" ) 2>&1 | grep -v amdgpu.ids | tail -6 > ${O}_cpw2_tests.log
( timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train2.py -m gpu -q -x ) 2>&1 | grep -v amdgpu.ids | tail -4 > ${O}_adam_tests.log
for cpw in 1 2; do
  DPOT_BF16P_BD=1 DPOT_BF16P_BD_CPW=$cpw timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2" > ${O}_cpw${cpw}_bench.txt
done
for cpw in 1 2; do
  DPOT_BF16P_BD=1 DPOT_BF16P_BD_CPW=$cpw timeout 300 python bench.py --config M --brief --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('M CPW=$cpw', d['ms_per_step'], d['value'])" >> ${O}_step.txt
  DPOT_BF16P_BD=1 DPOT_BF16P_BD_CPW=$cpw timeout 300 python bench.py --config L --brief --steps 8 --warmup 3 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('L CPW=$cpw', d['ms_per_step'], d['value'])" >> ${O}_step.txt
  DPOT_BF16P_BD=1 DPOT_BF16P_BD_CPW=$cpw timeout 300 python bench.py --config S --brief --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S CPW=$cpw', d['ms_per_step'], d['value'])" >> ${O}_step.txt
done
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/censusT -o r -- python $R/scripts/census_config.py T f32 > $R/gpurun_out/censusT.log 2>&1
cd $R; python scripts/step_census.py gpurun_out/censusT/r_results.db --seq > ${O}_census_T_seq.txt 2>&1; rm -rf gpurun_out/censusT
cat ${O}_cpw2_tests.log ${O}_adam_tests.log
paste -d'|' ${O}_cpw1_bench.txt ${O}_cpw2_bench.txt | cut -c1-75,100-175
cat ${O}_step.txt; head -5 ${O}_census_T_seq.txt
