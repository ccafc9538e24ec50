#!/bin/bash
# where do the ~40 us of fc1 forward's epilogue go?  ablation builds of the B-direct kernel (results are garbage, timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_t
for v in base abl_noact abl_nostore abl_notrans abl_nodact abl_norows abl_nont base; do
  L=""; [ $v != base ] && L="DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_$v.so"
  echo "== $v" >> ${O}_bench.txt
  env $L timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep "fc1 fwd, bf16\|fc1 fwd, inference\|fc2 dgrad, bf16\|^[ML] " >> ${O}_bench.txt
done
cat ${O}_bench.txt
