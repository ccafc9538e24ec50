#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_m
for rep in 1 2; do for rm in 0 1; do
  echo "== DPOT_BF16P_ROWMAJOR=$rm" >> ${O}_bench.txt
  DPOT_BF16P_ROWMAJOR=$rm timeout 600 python scripts/bf16p_train_bench.py M L 2>&1 | grep -v "amdgpu\|RASTER=\|round 2\|pack_both\|inference" >> ${O}_bench.txt
done; done
run() { local tag=$1; shift
  env "$@" timeout 300 python bench.py --config $CFG --brief --steps $ST --warmup $WU 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG $tag', d['ms_per_step'], d['value'])" >> ${O}_step.txt
}
for rep in 1 2; do for c in "M 20 5" "L 8 3"; do set -- $c; CFG=$1; ST=$2; WU=$3
  for rm in 0 1; do run "rowmajor=$rm" DPOT_BF16P_ROWMAJOR=$rm; done; done; done
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sizes.py -m gpu -q -x -k "bf16 and not x6 and not LARGE" ) 2>&1 | grep -v amdgpu.ids | tail -3 > ${O}_tests.log
bash scripts/r04/pmc_bf16p.sh M > ${O}_pmc.log 2>&1
cat ${O}_step.txt ${O}_tests.log; tail -75 ${O}_pmc.log | grep -E "fc|pair|traffic_over|mfma_util|FETCH_SIZE_KiB|WRITE_SIZE_KiB"
