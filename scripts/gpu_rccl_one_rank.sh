#!/bin/bash
# The N>1 code path against the real RCCL library on a 1-GPU box: the one-rank parity test, then bench.py on the same box as
# one graph (N=1 product path), as the segmented chain with the bucket all-reduces called on a one-rank RCCL communicator
# (DPOT_BENCH_FORCE_DP=1; RCCL enqueues no device work for one rank: host path + stream hand-offs), and with those calls
# captured inside one graph (DPOT_DP_ONE_GRAPH=1, opt-in).
mkdir -p gpurun_out
O=gpurun_out/r05_rccl_one_rank.txt
{
  [ -n "$SKIP_TEST" ] || timeout 900 python -m pytest tests/test_gpu_train2.py -q -s -k "one_rank_rccl" 2>&1 | tail -5
  for cfg in T M; do
    echo "== config $cfg: one graph"
    timeout 600 python bench.py --config $cfg --brief --no-alt 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['config'].get('host_us_per_step'), 'us host')"
    echo "== config $cfg: segmented chain, one-rank RCCL all-reduce per bucket"
    DPOT_BENCH_FORCE_DP=1 timeout 600 python bench.py --config $cfg 2>gpurun_out/r05_rccl_one_rank_$cfg.err | tail -1 > gpurun_out/r05_rccl_one_rank_$cfg.json
    python -c "import json; d=json.load(open('gpurun_out/r05_rccl_one_rank_$cfg.json')); c=d['config']; print(d['ms_per_step'], 'ms', c['host_us_per_step'], 'us host', c['collectives'], c['buckets_MB'], c['dp'])" || tail -5 gpurun_out/r05_rccl_one_rank_$cfg.err
    echo "== config $cfg: ONE graph holding the step and its bucket all-reduces (DPOT_DP_ONE_GRAPH=1)"
    DPOT_DP_ONE_GRAPH=1 DPOT_BENCH_FORCE_DP=1 timeout 600 python bench.py --config $cfg 2>gpurun_out/r05_rccl_one_graph_$cfg.err | tail -1 > gpurun_out/r05_rccl_one_graph_$cfg.json
    python -c "import json; d=json.load(open('gpurun_out/r05_rccl_one_graph_$cfg.json')); c=d['config']; print(d['ms_per_step'], 'ms', c['host_us_per_step'], 'us host', c['launch'], c['dp'])" || tail -5 gpurun_out/r05_rccl_one_graph_$cfg.err
  done
} > $O 2>&1
cat $O
