#!/usr/bin/env python
"""launch ONE form of the headline step's native fp32 GEMMs a few times (for rocprofv3 --pmc passes):
    panel  - gemm_panel_kernel<4,8,1>: 8192 x 512 x 512, bias + GELU + pre-activation saved (channel-MLP fc1 forward at DPOT-Tiny B=32)
    panel_lin - the same without the activation epilogue (fc2 forward / data gradients)
    tn     - gemm_tn_kernel via mlp_wgrad2: both weight gradients of a block (512 x 512 x 8192 each) + the reduce launch
operands rotate through 5 sets (cold, as inside the step)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpot_amd import ops
form = sys.argv[1] if len(sys.argv) > 1 else "panel"
M, N, K = 8192, 512, 512
if form.startswith("panel"):
    sets = [(torch.randn(M, K, device="cuda"),) for _ in range(5)]
    W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    pk = ops.PanelPacks([(W, N, K, K, False)]); pk.refresh()
    for i in range(15):
        A = sets[i % 5][0]
        if form == "panel":
            ops.gemm_panel(A, pk.bufs[0], N, bias=b, act=1, mode=ops.EPI_ACT, save_pre=True)
        else:
            ops.gemm_panel(A, pk.bufs[0], N)
else:
    T, E, mh = 8192, 512, 512
    sets = [[torch.randn(T, E, device="cuda"), torch.randn(T, mh, device="cuda"), torch.randn(T, E, device="cuda"),
             torch.randn(T, mh, device="cuda")] for _ in range(5)]
    dW2, dW1 = torch.empty(E, mh, device="cuda"), torch.empty(mh, E, device="cuda")
    db2, db1 = torch.empty(E, device="cuda"), torch.empty(mh, device="cuda")
    sk = ops.mlp_wgrad2_splitk(T, E, mh)
    for i in range(15):
        s = sets[i % 5]
        ops.mlp_wgrad2(s[0], s[1], s[2], s[3], dW2, db2, dW1, db1, sk)
torch.cuda.synchronize()
