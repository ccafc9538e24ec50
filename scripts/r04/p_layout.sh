#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_p
( timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_train2.py tests/test_gpu_sizes.py -m gpu -q -x -k "not LARGE and not large" ) 2>&1 | grep -v amdgpu.ids | tail -4 > ${O}_tests.log
for rep in 1 2 3; do for f in 0 1; do
  DPOT_LAYOUT_JOBS=$f timeout 300 python bench.py --brief --steps 100 --warmup 20 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('T layout_jobs=$f', d['ms_per_step'], d['value'])" >> ${O}_step.txt
done; done
cat ${O}_tests.log ${O}_step.txt
