#!/bin/bash
# round 4, batch F: persistent B-direct kernels (stores of a tile drain under the next tile's main loop)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_f
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sizes.py -m gpu -q -x -k "bf16 and not x6 and not LARGE" ) 2>&1 | grep -v amdgpu.ids | tail -4 > ${O}_tests.log
for ps in 0 1; do
  echo "== DPOT_BF16P_BD_PERSIST=$ps" >> ${O}_bench.txt
  DPOT_BF16P_BD_PERSIST=$ps timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2\|pack_both" >> ${O}_bench.txt
done
run() { local tag=$1; shift
  env "$@" timeout 300 python bench.py --config $CFG --brief --steps $ST --warmup $WU 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG $tag', d['ms_per_step'], d['value'])" >> ${O}_step.txt
}
for rep in 1 2; do
for c in "M 20 5" "L 8 3" "L20 3 1"; do
  set -- $c; CFG=$1; ST=$2; WU=$3
  run "persist=0" DPOT_BF16P_BD_PERSIST=0
  run "persist=1" DPOT_BF16P_BD_PERSIST=1
done
done
cat ${O}_tests.log ${O}_bench.txt ${O}_step.txt
