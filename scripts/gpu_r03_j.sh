#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for V in base abl_NOBAR abl_NOMMA abl_NOEPI abl_NOEPI_NOBAR; do
  if [ $V = base ]; then L=$PWD/dpot_amd/lib/libdpot_hip.so; else L=$PWD/dpot_amd/lib/variants/libdpot_hip_$V.so; fi
  echo "== $V" >> gpurun_out/r03j_ablation.txt
  DPOT_HIP_LIB=$L timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep -v "amdgpu\|round 2\|pack_both\|RASTER" >> gpurun_out/r03j_ablation.txt
done
cat gpurun_out/r03j_ablation.txt
