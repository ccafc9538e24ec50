"""n eager launches of the bf16x6 mixer kernel for rocprofv3 --pmc passes: python scripts/afno_mlp6_run.py l-fwd|l-bwd|l1-fwd|m-bwd [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DPOT_TUNE"] = "mixer6=2"
import torch
from dpot_amd import ops
form = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nb, bs, M = {"l": (16, 96, 8704), "l1": (16, 96, 333), "m": (8, 128, 4608)}[form.split("-")[0]]
N = 2 * bs; dev = "cuda"
S = torch.randn(M, nb * N, device=dev); pre = torch.randn(M, nb * N, device=dev)
w1 = torch.randn(2, nb, bs, bs, device=dev) * 0.05; w2 = torch.randn(2, nb, bs, bs, device=dev) * 0.05
c1 = torch.randn(2, nb, bs, device=dev) * 0.1; c2 = torch.randn(2, nb, bs, device=dev) * 0.1
ops.set_gemm_precision("auto")
it1, it2 = ops.AfnoPacks([(w1, c1), (w2, c2)]).refresh()
for _ in range(n):
    if form.endswith("fwd"):
        ops.afno_mlp2(S, it1.p6[0], it1[1], it2.p6[0], it2[1], nb, bs, 1, mode=0, want_pre=True, layout=2)
    else:
        ops.afno_mlp2(S, it2.p6[1], None, it1.p6[1], None, nb, bs, 1, mode=1, aux=pre, want_pre=True, want_mid=True, layout=2)
torch.cuda.synchronize()
