#!/bin/bash
# round 6, call 4: new tests; AFNO mixer X-swizzle A/B on one box (variant xsw3 = the rounds 2-5 image); Adam packs A/B; Tiny step
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train2.py -x -q -m gpu -k "selection or adam_writes or one_rank_rccl" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "afno" 2>&1 | tail -3
O=gpurun_out/r06_mixer_swizzle_ab.txt
{
  echo "AFNO mixer X-slab swizzle: XOR mask (r >> 2) & 3 (rounds 2-5: two lanes per 16-byte slot in every ds_read_b128 lane group) vs & 2 (round 6); one box"
  for rep in 1 2; do
    echo "== old image (variant xsw3)"; DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_xsw3.so python scripts/afno_mlp_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
    echo "== new image"; python scripts/afno_mlp_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
  done
} > $O 2>&1
cat $O | cut -c1-330
rm -f gpurun_out/r06_step_ab.txt
bash scripts/ab_config.sh gpurun_out/r06_step_ab.txt T 40 "DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_xsw3.so" "DPOT_X=1" "DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_xsw3.so" "DPOT_X=1" > /dev/null
bash scripts/ab_config.sh gpurun_out/r06_step_ab.txt M 20 "DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_xsw3.so DPOT_ADAM_PACKS=0" "DPOT_ADAM_PACKS=0" "DPOT_ADAM_PACKS=1" "DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_xsw3.so DPOT_ADAM_PACKS=0" "DPOT_ADAM_PACKS=0" "DPOT_ADAM_PACKS=1" > /dev/null
bash scripts/ab_config.sh gpurun_out/r06_step_ab.txt L 8 "DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_xsw3.so DPOT_ADAM_PACKS=0" "DPOT_ADAM_PACKS=0" "DPOT_ADAM_PACKS=1" "DPOT_ADAM_PACKS=0" "DPOT_ADAM_PACKS=1"
