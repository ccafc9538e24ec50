#!/bin/bash
# round 4, batch H: fragment-read scheduling variants of the B-direct kernel (variant builds, one box)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_h
for v in base bd_rf bd_pf; do
  L=""; [ $v != base ] && L="DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_$v.so"
  for cpw in 0 1; do
    echo "== $v CPW=$cpw" >> ${O}_bench.txt
    env $L DPOT_BF16P_BD_CPW=$cpw timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2\|pack_both\|inference" >> ${O}_bench.txt
  done
done
cat ${O}_bench.txt
