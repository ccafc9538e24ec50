#!/bin/bash
# three-product 96-channel AFNO weight-gradient kernel (gemm_tn96g_kernel) against the 192 x 192 four-product kernel: parity
# tests, the launch alone, the DPOT-L step - all on one box
mkdir -p gpurun_out
O=gpurun_out/r05_tn96g.txt
{
  timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "afno_wgrad2" 2>&1 | tail -3
  for g in 0 1; do DPOT_AFNO_WGRAD_GAUSS96=$g timeout 300 python scripts/tn_bench.py L 2>&1 | grep -v amdgpu.ids; done
  for rep in 1 2; do for g in 0 1; do
    echo "== DPOT-L batch 16, DPOT_AFNO_WGRAD_GAUSS96=$g (rep $rep)"
    DPOT_AFNO_WGRAD_GAUSS96=$g timeout 600 python bench.py --config L --brief --no-alt 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['config']['final_loss'])"
  done; done
} > $O 2>&1
cat $O
