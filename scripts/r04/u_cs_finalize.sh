#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_u
( timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sizes.py tests/test_gpu_train2.py -m gpu -q -x -k "(bf16 and not x6) or large_batch16 or finalize or two_process" ) 2>&1 | grep -v amdgpu.ids | tail -4 > ${O}_tests.log
for rep in 1 2; do for c in "M 20 5" "L 8 3"; do set -- $c; for f in 0 1; do
  DPOT_BLOCK_FINALIZE=$f timeout 300 python bench.py --config $1 --brief --no-alt --steps $2 --warmup $3 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 finalize=$f', d['ms_per_step'], d['value'])" >> ${O}_step.txt
done; done; done
cat ${O}_tests.log ${O}_step.txt
