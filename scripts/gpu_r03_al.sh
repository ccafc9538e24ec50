#!/bin/bash
# the opt-out switches of the round-3 kernels still give a correct step: bf16 size tests + op tests under each
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for e in DPOT_BF16P_DUO=0 DPOT_BF16P_TILE192=0 DPOT_GN_ONLOAD=0 DPOT_GN_DFT=0; do
  echo "== $e"
  env $e timeout 1500 python -m pytest tests/test_gpu_sizes.py tests/test_gpu_ops.py -m gpu -q -k "bf16 or pack or large_shape or pair or 192 or on_the_load or L" 2>&1 | tail -2
done > gpurun_out/r03al.txt 2>&1
cat gpurun_out/r03al.txt
