#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_q
for c in "S 20 5" "M 20 5" "L 8 3" "L20 3 1"; do set -- $c; for gp in f32 auto; do
  timeout 400 python bench.py --config $1 --brief --steps $2 --warmup $3 --gemm-precision $gp 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 gemm=$gp', d['ms_per_step'], d['value'], d['config']['peak_mem_GB'])" >> ${O}_step.txt
done; done
cat ${O}_step.txt
