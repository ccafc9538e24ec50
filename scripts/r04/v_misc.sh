#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_v
( timeout 1500 python -m pytest tests/test_gpu_sizes.py -m gpu -q -x -k "LARGE or large" ) 2>&1 | grep -v amdgpu.ids | tail -4 > ${O}_tests.log
for rep in 1 2; do for bd in 1 3; do
  DPOT_BF16P_BD=$bd timeout 300 python bench.py --config S --brief --no-alt --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S BD=$bd', d['ms_per_step'], d['value'])" >> ${O}_step.txt
done; done
for rep in 1 2; do for f in 0 1; do
  DPOT_BLOCK_FINALIZE=$f timeout 300 python bench.py --config L --brief --no-alt --steps 8 --warmup 3 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('L finalize=$f', d['ms_per_step'], d['value'])" >> ${O}_step.txt
done; done
cat ${O}_tests.log ${O}_step.txt
