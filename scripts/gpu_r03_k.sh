#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/r03k_ablation.txt
for V in abl_NOEPI nload8_NOEPI nload8 nload8_NOMMA; do
  L=$PWD/dpot_amd/lib/variants/libdpot_hip_$V.so
  echo "== $V" >> gpurun_out/r03k_ablation.txt
  DPOT_HIP_LIB=$L timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "amdgpu\|round 2\|pack_both\|RASTER" >> gpurun_out/r03k_ablation.txt
done
cat gpurun_out/r03k_ablation.txt
