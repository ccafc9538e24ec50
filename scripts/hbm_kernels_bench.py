"""graph-timed micro-benchmarks of the HBM-bound kernels of a block (GroupNorm, rfft2, irfft2) at DPOT-Tiny B=32"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

B, h, E, nb, m = 32, 16, 512, 4, 32
mx, my = min(m, h), min(m, h // 2 + 1)
x = torch.randn(B, h * h, E, device="cuda"); gw = torch.randn(E, device="cuda"); gb = torch.randn(E, device="cuda")
mb = x.numel() * 4 / 1e6
t = timeit(lambda: ops.groupnorm_fwd(x, gw, gb)); print(f"groupnorm_fwd: {t:.1f} us  ({2*mb:.0f} MB -> {2*mb/t*1e-3*1e3/1e3:.2f} TB/s)".replace("TB/s", "GB/ms"))
xn, mean, rstd = ops.groupnorm_fwd(x, gw, gb)
dy = torch.randn_like(x)
t = timeit(lambda: ops.groupnorm_bwd(dy, x, mean, rstd, gw)); print(f"groupnorm_bwd (+param grads): {t:.1f} us  ({3*mb:.0f} MB)")
S = ops.rfft2(x, h, h, nb, mx, my, 0); sb = S.numel() * 4 / 1e6
t = timeit(lambda: ops.rfft2(x, h, h, nb, mx, my, 0)); print(f"rfft2: {t:.1f} us  ({mb + sb:.0f} MB -> {(mb+sb)/t:.2f} TB/s)")
t = timeit(lambda: ops.irfft2(S, B, h, h, E, nb, mx, my, 1, res=x)); print(f"irfft2 (+res): {t:.1f} us  ({2*mb + sb:.0f} MB -> {(2*mb+sb)/t:.2f} TB/s)")
y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x)); print(f"torch copy: {t:.1f} us ({2*mb:.0f} MB -> {2*mb/t:.2f} TB/s)")
