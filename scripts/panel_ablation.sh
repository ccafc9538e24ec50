#!/bin/bash
# where the 44 us of gemm_panel_kernel<4,8,1> at 8192 x 512 x 512 go: ablation builds (garbage results) on one box
mkdir -p gpurun_out
O=gpurun_out/r05_panel_ablation.txt; rm -f $O
for v in "" panel_NOSTORE panel_NOEPI panel_NOMFMA; do
  echo "== variant: ${v:-product}" >> $O
  if [ -z "$v" ]; then timeout 300 python scripts/panel_bench.py 2>&1 | head -3 >> $O
  else DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_$v.so timeout 300 python scripts/panel_bench.py 2>&1 | head -3 >> $O; fi
done
for rt in 2 3; do
  echo "== product, DPOT_PANEL_RT=$rt" >> $O
  DPOT_PANEL_RT=$rt timeout 300 python scripts/panel_bench.py 2>&1 | head -3 >> $O
done
cat $O
