"""micro-benchmark of the 32 x 32 register-FFT launches of DPOT-L (csrc/dft_fast.h): rfft2 (plain / GroupNorm on load) and irfft2
(plain residual / GroupNorm(res) residual), batch 16 and 4, E = 1536, 16 blocks, all modes; hipGraph of launches on rotating
buffers (cold operands, as in the step), event timed; GB/s = algorithmic bytes (field + spectrum [+ residual]) / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops


def timeit(fns, reps=24):
    for f in fns[:2]: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fns[i % len(fns)]()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / reps)
    return min(ts)


h = w = 32; E, nb, mx, my, G = 1536, 16, 32, 17, 8
for B in (16, 4):
    nbuf = 4
    xs = [torch.randn(B, h * w, E, device="cuda") for _ in range(nbuf)]
    sp = [torch.randn(B * mx * my, 2 * E, device="cuda") for _ in range(nbuf)]
    mean, rstd = torch.randn(B, G, device="cuda") * 0.1, torch.rand(B, G, device="cuda") + 0.5
    ga, be = torch.rand(E, device="cuda") + 0.5, torch.randn(E, device="cuda") * 0.1
    fld, spc = xs[0].numel() * 4, sp[0].numel() * 4
    rows = [("rfft2", [lambda i=i: ops.rfft2(xs[i], h, w, nb, mx, my, 0) for i in range(nbuf)], fld + spc),
            ("rfft2 of GroupNorm(x)", [lambda i=i: ops.rfft2(xs[i], h, w, nb, mx, my, 0, norm=(mean, rstd, ga, be)) for i in range(nbuf)], fld + spc),
            ("irfft2 + residual", [lambda i=i: ops.irfft2(sp[i], B, h, w, E, nb, mx, my, 1, res=xs[i]) for i in range(nbuf)], spc + 2 * fld),
            ("irfft2 + GroupNorm(res)", [lambda i=i: ops.irfft2(sp[i], B, h, w, E, nb, mx, my, 1, res=xs[i], res_norm=(mean, rstd, ga, be)) for i in range(nbuf)], spc + 2 * fld),
            ("irfft2 (no residual)", [lambda i=i: ops.irfft2(sp[i], B, h, w, E, nb, mx, my, 0) for i in range(nbuf)], spc + fld)]
    for name, fns, by in rows:
        t = timeit(fns)
        print(f"B={B:2d} {name:<26s} {t * 1e6:7.1f} us  {by / t / 1e9:7.0f} GB/s", flush=True)
