#!/bin/bash
mkdir -p gpurun_out
bash scripts/r05/pmc_bf16p.sh M > gpurun_out/r05_pmc_bf16p_M.log 2>&1; tail -90 gpurun_out/r05_pmc_bf16p_M.log
mv gpurun_out/r05_pmc_bf16p_M.json gpurun_out/r05_pmc_bf16p_M_rowform1.json
DPOT_BF16P_ROWFORM=0 bash scripts/r05/pmc_bf16p.sh M > gpurun_out/r05_pmc_bf16p_M0.log 2>&1; tail -30 gpurun_out/r05_pmc_bf16p_M0.log
mv gpurun_out/r05_pmc_bf16p_M.json gpurun_out/r05_pmc_bf16p_M_rowform0.json
