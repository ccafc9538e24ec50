#!/bin/bash
# round 4, batch E: look-ahead depth of the B-direct kernels; shape rule vs BD everywhere vs off; cost of the prep launches
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_e
for la in 3 4 5 7; do
  echo "== DPOT_BF16P_BD=3 DPOT_BF16P_BD_P=$la" >> ${O}_la.txt
  DPOT_BF16P_BD=3 DPOT_BF16P_BD_P=$la timeout 600 python scripts/bf16p_train_bench.py S M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2\|pack_both" >> ${O}_la.txt
done
echo "== DPOT_BF16P_BD=0" >> ${O}_la.txt
DPOT_BF16P_BD=0 timeout 600 python scripts/bf16p_train_bench.py S M L 2>&1 | grep -v "amdgpu\|RASTER\|round 2\|pack_both" >> ${O}_la.txt
run() { local tag=$1; shift
  env "$@" timeout 300 python bench.py --config $CFG --brief --steps $ST --warmup $WU 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG $tag', d['ms_per_step'], d['value'])" >> ${O}_step.txt
}
for c in "S 20 5" "M 20 5" "L 8 3"; do
  set -- $c; CFG=$1; ST=$2; WU=$3
  run "BD=0" DPOT_BF16P_BD=0
  run "rule P3" DPOT_BF16P_BD=1
  run "rule P5" DPOT_BF16P_BD=1 DPOT_BF16P_BD_P=5
  run "rule P7" DPOT_BF16P_BD=1 DPOT_BF16P_BD_P=7
  run "all P7" DPOT_BF16P_BD=3 DPOT_BF16P_BD_P=7
done
timeout 300 python scripts/r04/prep_cost.py 2>&1 | grep -v amdgpu > ${O}_prep_cost.txt
( DPOT_BF16P_BD=3 DPOT_BF16P_BD_P=7 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sizes.py -m gpu -q -x -k "bf16 and not x6 and not LARGE" ) 2>&1 | grep -v amdgpu.ids | tail -4 > ${O}_p7_tests.log
cat ${O}_la.txt; cat ${O}_step.txt ${O}_prep_cost.txt ${O}_p7_tests.log
