mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train2.py -q -x --timeout=600 -k "dp_step or segmented or one_graph or collectives" -p no:cacheprovider 2>&1 | tail -5
for c in T M; do DPOT_BENCH_FORCE_DP=1 timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-other-configs --no-alt --no-pipeline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config'].get('workload','')[:60], d['ms_per_step'], d['value'], json.dumps(d['config'].get('dp', d['config'].get('parallelism')))[:600])"; done
