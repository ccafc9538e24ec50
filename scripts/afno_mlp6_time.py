"""times of the bf16x6 mixer kernel alone (no checks): python scripts/afno_mlp6_time.py  [DPOT_HIP_LIB=variant]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops
from afno_mlp6_check import timeit
def run(nb, bs, M):
    N = 2 * bs; dev = "cuda"
    S = torch.randn(M, nb * N, device=dev); pre = torch.randn(M, nb * N, device=dev)
    w1 = torch.randn(2, nb, bs, bs, device=dev) * 0.05; w2 = torch.randn(2, nb, bs, bs, device=dev) * 0.05
    c1 = torch.randn(2, nb, bs, device=dev) * 0.1; c2 = torch.randn(2, nb, bs, device=dev) * 0.1
    ops.set_gemm_precision("auto")
    it1, it2 = ops.AfnoPacks([(w1, c1), (w2, c2)]).refresh()
    tf = timeit(lambda: ops.afno_mlp2(S, it1.p6[0], it1[1], it2.p6[0], it2[1], nb, bs, 1, mode=0, want_pre=True, layout=2))
    tb = timeit(lambda: ops.afno_mlp2(S, it2.p6[1], None, it1.p6[1], None, nb, bs, 1, mode=1, aux=pre, want_pre=True, want_mid=True, layout=2))
    return f"nb={nb} bs={bs} M={M}: fwd {tf:6.1f} bwd {tb:6.1f}"
print(os.environ.get("DPOT_HIP_LIB", "product")[-28:], " | ".join(run(*a) for a in ((4, 128, 4608), (8, 128, 4608), (16, 96, 8704), (16, 96, 333))), flush=True)
