#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_s
for c in "S 20 5" "M 20 5" "L 8 3"; do set -- $c; for w in 1 0; do
  DPOT_AFNO_WGRAD2=$w timeout 400 python bench.py --config $1 --brief --no-alt --steps $2 --warmup $3 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 afno_wgrad2=$w', d['ms_per_step'], d['value'])" >> ${O}_step.txt
done; done
cat ${O}_step.txt
