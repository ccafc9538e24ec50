#!/bin/bash
# DPOT-L, 20-step rollout: the (per-GPU batch, kept AR steps) plane on ONE box -> gpurun_out/r06_l20_sweep.txt
# memory model (profiles/r05_l20_keep_last.txt): ~7.03 GiB per sample with every step recomputed + ~1.072 GiB per sample and kept step
mkdir -p gpurun_out
O=gpurun_out/r06_l20_sweep.txt
{
  echo "DPOT-L 20-step rollout, bench.py --config L20 --brief --no-alt --steps 2 --warmup 1, one box; B = per-GPU batch, k = AR steps that keep their activations"
  for P in ${POINTS:-"16 8" "8 20" "8 16" "10 17" "12 13" "12 10" "16 6"}; do
    set -- $P
    echo "== B=$1 keep_last=$2"
    DPOT_BENCH_KEEP_LAST=$2 timeout 600 python bench.py --config L20 --batch $1 --brief --no-alt --steps 2 --warmup 1 2>gpurun_out/l20_sweep.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], 'ms', d['value'], d['unit'], 'peak', c['peak_mem_GB'], 'GiB', c['activation_recomputation'], c['final_loss'])" || tail -5 gpurun_out/l20_sweep.err
  done
} > $O 2>&1
cat $O
