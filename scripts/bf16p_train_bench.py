"""micro-benchmark: the bf16 channel-MLP GEMMs in the forms the DPOT-M / -L train step launches them (packed outputs),
per shape: fc1 fwd (act + packs + act' pack | old: fp32 pre-activation), fc2 fwd (+ residual), fc2 dgrad (act' product +
packs + column sums), fc1 dgrad."""
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


shapes = [("M B=32", 8192, 1024, 4096), ("L B=4", 4096, 1536, 6144), ("L B=16", 16384, 1536, 6144), ("S B=32", 8192, 1024, 1024)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0].split()[0] in sys.argv[1:]]
for name, M, E, mh in shapes:
    x = torch.randn(M, E, device="cuda"); do = torch.randn(M, E, device="cuda")
    W1 = torch.randn(mh, E, device="cuda") * 0.03; W2 = torch.randn(E, mh, device="cuda") * 0.03
    b1 = torch.randn(mh, device="cuda") * 0.1; b2 = torch.randn(E, device="cuda") * 0.1
    pk = ops.PanelPacks([(W1, mh, E, E, False), (W1, E, mh, E, True), (W2, E, mh, mh, False), (W2, mh, E, mh, True)], bf16=True)
    pk.refresh()
    xp, xpT, _ = ops.bf16_pack_both(x)
    dop, dopT, _ = ops.bf16_pack_both(do)
    fl = 2.0 * M * E * mh
    rep = lambda tag, t: print(f"  {tag:<46s} {t*1e6:7.1f} us  {fl/t/1e12:7.1f} TF", flush=True)
    print(f"{name}: tokens {M}, E {E}, hidden {mh}  ({fl/1e9:.1f} GFLOP per GEMM)")
    t = timeit(lambda: ops.gemm_bf16p_packed(xp, pk.bufs[0], M, mh, E, bias=b1, act=1, mode=ops.EPI_ACT, save_pre=True,
                                             pack_rows=True, pack_trans=True, store=False))
    rep("fc1 fwd, fp32 pre-activation saved (round 2)", t)
    t = timeit(lambda: ops.gemm_bf16p_packed(xp, pk.bufs[0], M, mh, E, bias=b1, act=1, mode=ops.EPI_ACT, save_dact=True,
                                             pack_rows=True, pack_trans=True, store=False))
    rep("fc1 fwd, bf16 act' pack saved", t)
    t = timeit(lambda: ops.gemm_bf16p_packed(xp, pk.bufs[0], M, mh, E, bias=b1, act=1, mode=ops.EPI_ACT,
                                             pack_rows=True, store=False))
    rep("fc1 fwd, inference form (row pack only)", t)
    _, Hpre, hp, hpT, _ = ops.gemm_bf16p_packed(xp, pk.bufs[0], M, mh, E, bias=b1, act=1, mode=ops.EPI_ACT, save_pre=True,
                                                pack_rows=True, pack_trans=True, store=False)
    _, D, _, _, _ = ops.gemm_bf16p_packed(xp, pk.bufs[0], M, mh, E, bias=b1, act=1, mode=ops.EPI_ACT, save_dact=True,
                                          pack_rows=True, pack_trans=True, store=False)
    t = timeit(lambda: ops.gemm_bf16p(hp, pk.bufs[2], M, E, mh, bias=b2, res=x))
    rep("fc2 fwd (+ bias, residual; fp32 out)", t)
    t = timeit(lambda: ops.gemm_bf16p_packed(dop, pk.bufs[3], M, mh, E, act=1, mode=ops.EPI_DACT, aux=Hpre, pack_rows=True,
                                             pack_trans=True, colsum=True, store=False))
    rep("fc2 dgrad, act'(fp32 pre) (round 2)", t)
    t = timeit(lambda: ops.gemm_bf16p_packed(dop, pk.bufs[3], M, mh, E, act=1, mode=ops.EPI_DACT, dact=D, pack_rows=True,
                                             pack_trans=True, colsum=True, store=False))
    rep("fc2 dgrad, bf16 act' pack", t)
    _, _, dhp, dhpT, _ = ops.gemm_bf16p_packed(dop, pk.bufs[3], M, mh, E, act=1, mode=ops.EPI_DACT, dact=D, pack_rows=True,
                                               pack_trans=True, colsum=True, store=False)
    t = timeit(lambda: ops.gemm_bf16p(dhp, pk.bufs[1], M, E, mh))
    rep("fc1 dgrad (fp32 out)", t)
    if ops.gemm_bf16p_pair_wanted(E, mh, mh, E, M):
        o0, o1 = torch.empty(E, mh, device="cuda"), torch.empty(mh, E, device="cuda")
        t = timeit(lambda: ops.gemm_bf16p_pair(dopT, hpT, E, mh, dhpT, xpT, mh, E, M, out0=o0, out1=o1))
        print(f"  {'fc2 + fc1 weight gradients, pair launch':<46s} {t*1e6:7.1f} us  {2*fl/t/1e12:7.1f} TF", flush=True)
    t = timeit(lambda: ops.bf16_pack_both(x))
    print(f"  {'pack_both of a [tokens, E] activation':<46s} {t*1e6:7.1f} us  {M*E*8/t/1e12:7.2f} TB/s", flush=True)
    del x, do, W1, W2, pk, xp, xpT, dop, dopT, Hpre, hp, hpT, D, dhp, dhpT
    torch.cuda.empty_cache()
