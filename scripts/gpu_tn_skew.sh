#!/bin/bash
# three-product AFNO weight gradients at 128 channels per block: shorter token ranges for the (slower) sum-product tile
# (DPOT_TN_GAUSS_SKEW = s12/s3; 1/1 = equal ranges): parity, the launch alone, the train steps - one box
mkdir -p gpurun_out
O=gpurun_out/r05_tn_skew.txt
{
  timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "afno_wgrad2 or block_finalize" 2>&1 | tail -3
  for sk in 1/1 5/6 4/5 3/4; do
    echo "== DPOT_TN_GAUSS_SKEW=$sk"
    DPOT_TN_GAUSS_SKEW=$sk timeout 300 python scripts/tn_bench.py 2>&1 | grep "afno_wgrad2" | grep "\*\|splitk=10 \|splitk=12 \|splitk= 5 \|splitk= 6 " | grep -v "sets=1"
  done
  for rep in 1 2; do for cfg in T S M; do for sk in 1/1 5/6; do
    echo "== config $cfg DPOT_TN_GAUSS_SKEW=$sk (rep $rep)"
    DPOT_TN_GAUSS_SKEW=$sk timeout 600 python bench.py --config $cfg --brief --no-alt 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['config']['final_loss'])"
  done; done; done
} > $O 2>&1
cat $O
