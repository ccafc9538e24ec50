"""micro-benchmark: fp32 panel GEMM vs the generic 64x64-tile kernel at the DPOT-Tiny channel-MLP shape, per epilogue"""
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
    return e0.elapsed_time(e1) * 1e-3 / reps

shapes = [(8192, 512, 512), (8192, 2048, 512), (8192, 512, 2048)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for M, N, K in shapes:
    # rotate over several operand sets so that the inputs do not sit in L2 from the previous repetition
    sets = []
    for i in range(6):
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
        R = torch.randn(M, N, device="cuda"); X = torch.randn(M, N, device="cuda")
        pf = ops.PanelPacks([(W, N, K, K, False)]); pf.refresh()
        sets.append((A, W, b, R, X, pf))
    it = [0]
    def nxt():
        it[0] = (it[0] + 1) % len(sets); return sets[it[0]]
    fl = 2.0 * M * N * K
    rows = []
    def both(name, fp, fg):
        tp, tg = timeit(fp), timeit(fg)
        print(f"M={M} N={N} K={K} {name:18s}: panel {tp*1e6:6.1f} us {fl/tp/1e12:6.1f} TF | generic {tg*1e6:6.1f} us {fl/tg/1e12:6.1f} TF", flush=True)
    def p_lin():
        A, W, b, R, X, pf = nxt(); ops.gemm_panel(A, pf.bufs[0], N, bias=b)
    def g_lin():
        A, W, b, R, X, pf = nxt(); ops.linear_fwd(A, W, b, precision=ops.GEMM_F32)
    both("linear+bias", p_lin, g_lin)
    def p_act():
        A, W, b, R, X, pf = nxt(); ops.gemm_panel(A, pf.bufs[0], N, bias=b, act=1, mode=ops.EPI_ACT, save_pre=True)
    def g_act():
        A, W, b, R, X, pf = nxt(); ops.linear_fwd(A, W, b, act=1, save_pre=True, precision=ops.GEMM_F32)
    both("gelu+save_pre", p_act, g_act)
    def p_res():
        A, W, b, R, X, pf = nxt(); ops.gemm_panel(A, pf.bufs[0], N, bias=b, res=R)
    def g_res():
        A, W, b, R, X, pf = nxt(); ops.linear_fwd(A, W, b, res=R, precision=ops.GEMM_F32)
    both("linear+res", p_res, g_res)
    def p_dact():
        A, W, b, R, X, pf = nxt(); ops.gemm_panel(A, pf.bufs[0], N, act=1, mode=ops.EPI_DACT, aux=X)
    both("dact(aux)", p_dact, p_dact)
