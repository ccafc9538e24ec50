"""graph-timed micro-benchmarks of the batch-only head of the Tiny step (noise, patch embedding) - sustained clocks"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops, functional as F

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

B, X, T, Cc, P, hid = 32, 128, 10, 4, 8, 35
x = torch.randn(B, X, X, T, Cc, device="cuda")
mb = x.numel() * 4 / 1e6
t = timeit(lambda: ops.noise_inject(x, None, 0.0005))
print(f"noise_inject (rng): {t:.1f} us   ({2*mb:.0f} MB r+w + {mb:.0f} MB norm pass -> {3*mb/t*1e-6*1e6/1e6:.2f} TB/s)")
y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x)); print(f"torch copy {mb:.0f} MB: {t:.1f} us -> {2*mb/t/1e6*1e6/1e6:.2f} TB/s")
t = timeit(lambda: (x * x).sum()); print(f"torch x*x sum: {t:.1f} us")
gx = torch.linspace(0, 1, X).cuda(); gt = torch.linspace(0, 1, T).cuda()
K0 = (Cc + 3) * P * P; hidp = 36
w0 = torch.randn(hid, Cc + 3, P, P, device="cuda") / math.sqrt(K0); b0 = torch.randn(hid, device="cuda")
w0p = ops.copy2d_pad(w0, hid, K0, hidp, K0); b0p = ops.copy2d_pad(b0, 1, hid, 1, hidp).view(hidp)
grid = F.embed_grid_matrix(gx, gx, gt, X, X, T, Cc, P)
wfrag = ops.embed_pack_w0(w0)
bt = torch.empty(grid.shape[0], hidp, device="cuda")
ops.gemm(grid, w0p[:, Cc * P * P:], bt, grid.shape[0], hidp, grid.shape[1], transB=True, lda=grid.shape[1], ldb=K0, ldc=hidp, bias=b0p)
t = timeit(lambda: ops.embed_fwd(x, wfrag, bt, hidp, 1)); print(f"embed_fwd implicit: {t:.1f} us")
def old():
    A0 = ops.patchify(x, gx, gx, gt, P)
    return ops.linear_fwd(A0, w0p, b0p, act=1, save_pre=True)
t = timeit(old); print(f"patchify + GEMM: {t:.1f} us")
dH = torch.randn(B * 256 * T, hidp, device="cuda")
dw0 = torch.empty(hid, K0, device="cuda")
t = timeit(lambda: ops.embed_wgrad(x, dH, dw0, hid)); print(f"embed_wgrad implicit (+reduce): {t:.1f} us")
A0 = ops.patchify(x, gx, gx, gt, P)
t = timeit(lambda: ops.linear_bwd_weight(dH, A0)); print(f"wgrad GEMM on the patch matrix: {t:.1f} us")
