#!/usr/bin/env python
"""launch ONE form of the AFNO weight-gradient kernels a few times (for rocprofv3 --pmc passes):
    T   - gemm_tn_kernel, three-product form, DPOT-Tiny B=32 (Mm 4608, 4 blocks of 128 channels)
    M   - the same at DPOT-S / -M (8 blocks of 128)
    L   - gemm_tn96g_kernel, DPOT-L B=16 (Mm 8704, 16 blocks of 96);  DPOT_TUNE=wgrad_gauss=0: gemm_tn192_kernel
operands rotate through 3 sets"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpot_amd import ops
form = sys.argv[1] if len(sys.argv) > 1 else "T"
Mm, nb, bs = {"T": (4608, 4, 128), "M": (4608, 8, 128), "L": (16 * 32 * 17, 16, 96)}[form]
E = nb * bs
sets = [[torch.randn(Mm, 2 * E, device="cuda") for _ in range(4)] for _ in range(3)]
dw1, dw2 = torch.empty(2, nb, bs, bs, device="cuda"), torch.empty(2, nb, bs, bs, device="cuda")
db1, db2 = torch.empty(2, nb, bs, device="cuda"), torch.empty(2, nb, bs, device="cuda")
sk = ops.afno_wgrad2_splitk(Mm, nb, bs)
for i in range(15):
    s = sets[i % 3]
    ops.afno_wgrad2(s[0], s[1], s[2], s[3], nb, bs, dw1, db1, dw2, db2, sk)
torch.cuda.synchronize()
print("ok", form, "splitk", sk)
