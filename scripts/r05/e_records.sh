#!/bin/bash
# round 5, first records on the restored tree: f4 A/B (layer + train step), fp32 GEMM counters, DPOT-S census + counters
mkdir -p gpurun_out
timeout 300 python scripts/afno_layer_bench.py > gpurun_out/r05_f4_fused_vs_3launch.txt 2>&1
cat gpurun_out/r05_f4_fused_vs_3launch.txt
rm -f gpurun_out/r05_f4_step_ab.txt
for cfg in S M T; do
  for v in 0 1 0 1; do
    echo -n "config $cfg DPOT_AFNO_LAYER=$v: " >> gpurun_out/r05_f4_step_ab.txt
    DPOT_AFNO_LAYER=$v timeout 300 python bench.py --config $cfg --brief --skip-cpu-baseline --no-other-configs --no-alt --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms/step', d['value'], d['unit'])" >> gpurun_out/r05_f4_step_ab.txt 2>&1
  done
done
cat gpurun_out/r05_f4_step_ab.txt
bash scripts/r05/pmc_gemm_f32.sh > gpurun_out/r05_pmc_gemm_f32.log 2>&1; tail -60 gpurun_out/r05_pmc_gemm_f32.log
bash scripts/gpu_census_M.sh S bf16 > /dev/null 2>&1; mv gpurun_out/censusS.txt gpurun_out/r05_census_S.txt; head -50 gpurun_out/r05_census_S.txt
bash scripts/r05/pmc_bf16p.sh S > gpurun_out/r05_pmc_bf16p_S.log 2>&1; tail -80 gpurun_out/r05_pmc_bf16p_S.log
