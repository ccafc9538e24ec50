"""micro-benchmark: bf16 panel GEMM (csrc/gemm_bf16p.hip) incl. its activation pack vs the fp32 panel / generic kernels"""
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

for M, N, K in ((8192, 512, 512), (8192, 1024, 1024), (8192, 4096, 1024), (8192, 1024, 4096), (4096, 6144, 1536), (4096, 1536, 6144)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    pb = ops.PanelPacks([(W, N, K, K, False)], bf16=True); pb.refresh()
    pf = ops.PanelPacks([(W, N, K, K, False)]); pf.refresh()
    Ap = ops.bf16_pack_rows(A)
    t_g = timeit(lambda: ops.gemm_bf16p(Ap, pb.bufs[0], M, N, K, bias=b, act=1, mode=ops.EPI_ACT, save_pre=True))
    t_p = timeit(lambda: ops.bf16_pack_rows(A))
    t_f = timeit(lambda: ops.gemm_panel(A, pf.bufs[0], N, bias=b, act=1, mode=ops.EPI_ACT, save_pre=True))
    t_x = timeit(lambda: ops.linear_fwd(A, W, b, act=1, save_pre=True, precision=ops.GEMM_BF16X6))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: bf16p {t_g*1e6:7.1f} us {fl/t_g/1e12:6.1f} TF (+ pack A {t_p*1e6:6.1f} us -> {fl/(t_g+t_p)/1e12:6.1f} TF) | "
          f"fp32 panel {t_f*1e6:7.1f} us {fl/t_f/1e12:6.1f} TF | bf16x6 {t_x*1e6:7.1f} us {fl/t_x/1e12:6.1f} TF", flush=True)

print("weight-gradient form: dW[n,k] = dy^T x, tokens = GEMM k-dim, both operands packed transposed, split-K", flush=True)
for T, N, K in ((8192, 4096, 1024), (8192, 1024, 4096), (4096, 6144, 1536), (4096, 1536, 6144), (8192, 1024, 1024)):
    dy = torch.randn(T, N, device="cuda"); x = torch.randn(T, K, device="cuda")
    dyp, xp = ops.bf16_pack_rows(dy, trans=True), ops.bf16_pack_rows(x, trans=True)
    out = torch.empty(N, K, device="cuda")
    sk = ops._lib.load().dpot_gemm_bf16p_splitk(N, K, T)
    t_g = timeit(lambda: ops.gemm_bf16p(dyp, xp, N, K, T, out=out))
    t_p = timeit(lambda: (ops.bf16_pack_rows(dy, trans=True), ops.bf16_pack_rows(x, trans=True)))
    t_x = timeit(lambda: ops.linear_bwd_weight(dy, x, precision=ops.GEMM_BF16X6))
    t_f = timeit(lambda: ops.linear_bwd_weight(dy, x))
    fl = 2.0 * T * N * K
    print(f"tokens={T} n={N} k={K} splitk={sk}: bf16p {t_g*1e6:7.1f} us {fl/t_g/1e12:6.1f} TF (+ packs {t_p*1e6:6.1f} us -> "
          f"{fl/(t_g+t_p)/1e12:6.1f} TF) | fp32 split {t_f*1e6:7.1f} us {fl/t_f/1e12:6.1f} TF | bf16x6 {t_x*1e6:7.1f} us "
          f"{fl/t_x/1e12:6.1f} TF", flush=True)

print("bf16x6 (three planes, fp32-accurate) panel vs the register-staged bf16x6 kernel and the fp32 panel", flush=True)
for M, N, K in ((8192, 512, 512), (8192, 1024, 1024), (8192, 4096, 1024), (8192, 1024, 4096), (4096, 6144, 1536), (4096, 1536, 6144)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    pb = ops.PanelPacks([(W, N, K, K, False)], bf16=True, planes=3); pb.refresh()
    pf = ops.PanelPacks([(W, N, K, K, False)]); pf.refresh()
    Ap = ops.bf16_pack_rows(A, planes=3)
    t_g = timeit(lambda: ops.gemm_bf16p(Ap, pb.bufs[0], M, N, K, bias=b, act=1, mode=ops.EPI_ACT, save_pre=True, planes=3))
    t_p = timeit(lambda: ops.bf16_pack_rows(A, planes=3))
    t_f = timeit(lambda: ops.gemm_panel(A, pf.bufs[0], N, bias=b, act=1, mode=ops.EPI_ACT, save_pre=True))
    t_x = timeit(lambda: ops.linear_fwd(A, W, b, act=1, save_pre=True, precision=ops.GEMM_BF16X6))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: x6 panel {t_g*1e6:7.1f} us {fl/t_g/1e12:6.1f} TF-equiv (+ pack A {t_p*1e6:6.1f} us -> {fl/(t_g+t_p)/1e12:6.1f}) | "
          f"fp32 panel {t_f*1e6:7.1f} us {fl/t_f/1e12:6.1f} TF | bf16x6 split kernel {t_x*1e6:7.1f} us {fl/t_x/1e12:6.1f} TF", flush=True)
for T, N, K in ((8192, 4096, 1024), (8192, 512, 512), (4096, 6144, 1536)):
    dy = torch.randn(T, N, device="cuda"); x = torch.randn(T, K, device="cuda")
    dyp, xp = ops.bf16_pack_rows(dy, trans=True, planes=3), ops.bf16_pack_rows(x, trans=True, planes=3)
    out = torch.empty(N, K, device="cuda")
    t_g = timeit(lambda: ops.gemm_bf16p(dyp, xp, N, K, T, out=out, planes=3))
    t_p = timeit(lambda: (ops.bf16_pack_rows(dy, trans=True, planes=3), ops.bf16_pack_rows(x, trans=True, planes=3)))
    t_x = timeit(lambda: ops.linear_bwd_weight(dy, x, precision=ops.GEMM_BF16X6))
    fl = 2.0 * T * N * K
    print(f"wgrad tokens={T} n={N} k={K}: x6 panel {t_g*1e6:7.1f} us {fl/t_g/1e12:6.1f} TF-equiv (+ packs {t_p*1e6:6.1f} us -> "
          f"{fl/(t_g+t_p)/1e12:6.1f}) | bf16x6 split kernel {t_x*1e6:7.1f} us {fl/t_x/1e12:6.1f} TF", flush=True)
