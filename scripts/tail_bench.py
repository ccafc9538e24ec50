"""graph-timed micro-benchmark of the fused out-layer tail (csrc/tail.hip) at DPOT-Tiny B=32: forward and backward"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

B, h, P, co = 32, 16, 8, 4
npx = B * h * h * P * P
bufs = [torch.randn(npx, 32, device="cuda") for _ in range(3)]
douts = [torch.randn(B, h * P, h * P, co, device="cuda") for _ in range(3)]
w2 = torch.randn(32, 32, device="cuda") * 0.2; b2 = torch.randn(32, device="cuda") * 0.1
w4 = torch.randn(co, 32, device="cuda") * 0.2; b4 = torch.randn(co, device="cuda") * 0.1
w4p, b4p = ops.out_tail_pad(w4, b4, co)
i = [0]
def fwd():
    i[0] = (i[0] + 1) % 3
    return ops.out_tail_fwd(bufs[i[0]], w2, b2, w4p, b4p, B, h, h, P, co, 1)
def bwd():
    i[0] = (i[0] + 1) % 3
    return ops.out_tail_bwd(bufs[i[0]], douts[i[0]], w2, b2, w4p, B, h, h, P, co, 1)
mb_f = (npx * 32 + npx * co) * 4 / 1e6
mb_b = (2 * npx * 32 + npx * co) * 4 / 1e6
t = timeit(fwd); print(f"out_tail_fwd: {t:.1f} us  ({mb_f:.0f} MB -> {mb_f / t:.2f} TB/s = {mb_f / t / 8:.2f} of 8 TB/s)")
t = timeit(bwd); print(f"out_tail_bwd (+ partial-row reduction): {t:.1f} us  ({mb_b:.0f} MB -> {mb_b / t:.2f} TB/s = {mb_b / t / 8:.2f} of 8 TB/s)")
