#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_r
( time timeout 1500 python -m pytest tests/test_gpu_optout.py -m gpu -q -s -k "GEMM_PRECISION" ) 2>&1 | grep -v amdgpu.ids | tail -12 > ${O}_optout_auto.log
( time timeout 600 python bench.py ) > ${O}_bench.json 2> ${O}_bench.err
cat ${O}_optout_auto.log; tail -4 ${O}_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_r_bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['dtype'])
for o in d.get('other_configs',[]): print({k:o.get(k) for k in ('config','value','ms_per_step','gemm_precision','dtype','wall_s','error')}, o.get('gemm_f32'))
PY
