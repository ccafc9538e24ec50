#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/r03m_ablation.txt
for V in NOEPI NOEPI_NOBREAD NOEPI_NOMMA NOEPI_NOMMA_NOBREAD; do
  echo "== $V" >> gpurun_out/r03m_ablation.txt
  DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_$V.so timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep "fc2 fwd\|fc1 dgrad\|inference" >> gpurun_out/r03m_ablation.txt
done
cat gpurun_out/r03m_ablation.txt
