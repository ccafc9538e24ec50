"""micro-benchmark: the fused GroupNorm + DFT kernels (csrc/gn_dft.hip) against the separate kernel pairs they replace,
DPOT-Tiny (E = 512: 64 channels per group) and DPOT-S/M (E = 1024: 128 per group) at B = 32, hipGraph of 30 launches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops


def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for E, nb in ((512, 4), (1024, 8)):
    B, h = 32, 16
    mx, my = 16, 9
    x = torch.randn(B, h * h, E, device="cuda"); dy = torch.randn_like(x); dout = torch.randn_like(x)
    g1, b1 = torch.randn(E, device="cuda"), torch.randn(E, device="cuda")
    xn1, m1, r1 = ops.groupnorm_fwd(x, g1, b1)
    S = ops.rfft2(xn1, h, h, nb, mx, my, 0)
    y1 = ops.irfft2(S, B, h, h, E, nb, mx, my, 1, res=xn1)
    _, m2, r2 = ops.groupnorm_fwd(y1, g1, b1)
    rows = [
        ("K1 norm1 + rfft2", lambda: ops.gn_rfft2(x, g1, b1, h, h, nb, mx, my),
         lambda: ops.rfft2(ops.groupnorm_fwd(x, g1, b1)[0], h, h, nb, mx, my, 0)),
        ("K2 irfft2 + x_orig + norm2", lambda: ops.irfft2_gn(S, x, m1, r1, g1, b1, g1, b1, h, h, nb, mx, my),
         lambda: ops.groupnorm_fwd(ops.irfft2(S, B, h, h, E, nb, mx, my, 1, res=xn1), g1, b1)),
        ("K3 norm2 bwd + rfft2 adj", lambda: ops.gn_bwd_rfft2(dy, y1, m2, r2, g1, h, h, nb, mx, my),
         lambda: ops.rfft2(ops.groupnorm_bwd(dy, y1, m2, r2, g1, defer=True)[0], h, h, nb, mx, my, 1)),
        ("K4 irfft2 adj + skip + norm1 bwd + skip", lambda: ops.irfft2_gn_bwd(S, dy, x, m1, r1, g1, h, h, nb, mx, my, add=dout),
         lambda: ops.groupnorm_bwd(ops.irfft2(S, B, h, h, E, nb, mx, my, 0, res=dy), x, m1, r1, g1, add=dout, defer=True)),
    ]
    print(f"E={E} ({E // 8} channels per group), B={B}")
    for name, fused, sep in rows:
        tf, ts = timeit(fused), timeit(sep)
        print(f"  {name:<42s} fused {tf:6.1f} us   separate pair {ts:6.1f} us", flush=True)
