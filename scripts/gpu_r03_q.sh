#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/r03q_ablation.txt
for V in base abl_NOACT abl_NOTRANS abl_NOSTORE abl_NOEPI; do
  if [ $V = base ]; then L=$PWD/dpot_amd/lib/libdpot_hip.so; else L=$PWD/dpot_amd/lib/variants/libdpot_hip_$V.so; fi
  echo "== $V" >> gpurun_out/r03q_ablation.txt
  DPOT_HIP_LIB=$L timeout 600 python scripts/bf16p_train_bench.py M 2>&1 | grep "fc1 fwd, bf16\|inference\|fc2 dgrad, bf16" >> gpurun_out/r03q_ablation.txt
done
cat gpurun_out/r03q_ablation.txt
