#!/bin/bash
# DPOT-L, 20-step rollout, batch 16: AR steps that keep their activations (selective recomputation) - step time and peak memory
mkdir -p gpurun_out
O=gpurun_out/r05_l20_keep_last.txt
{
  for k in ${KEEPS:-0 2 4}; do
    echo "== DPOT_BENCH_KEEP_LAST=$k"
    DPOT_BENCH_KEEP_LAST=$k timeout 900 python bench.py --config L20 --brief --no-alt --steps 2 --warmup 1 2>gpurun_out/l20_keep.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], 'ms', d['value'], d['unit'], 'peak', c['peak_mem_GB'], 'GB', c['activation_recomputation'], c['final_loss'])" || tail -5 gpurun_out/l20_keep.err
  done
} > $O 2>&1
cat $O
