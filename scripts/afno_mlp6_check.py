"""the bf16x6 mixer kernel (csrc/afno_mlp6.hip) against the fp32 matrix-core kernel and a float64 torch product: errors and times"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops

def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

def rel(a, b):
    return ((a.double() - b).norm() / b.norm()).item()

def run(nb, bs, M, seed=0):
    torch.manual_seed(seed)
    N = 2 * bs
    dev = "cuda"
    S = torch.randn(M, nb * N, device=dev)
    pre = torch.randn(M, nb * N, device=dev)
    w1 = torch.randn(2, nb, bs, bs, device=dev) * 0.05; w2 = torch.randn(2, nb, bs, bs, device=dev) * 0.05
    c1 = torch.randn(2, nb, bs, device=dev) * 0.1; c2 = torch.randn(2, nb, bs, device=dev) * 0.1
    ops.set_gemm_precision("auto")
    pk = ops.AfnoPacks([(w1, c1), (w2, c2)])
    it1, it2 = pk.refresh()
    (wbig1, bb1, f1, k1), (wbig2, bb2, f2, k2) = it1, it2
    assert it1.p6 is not None
    # float64 reference
    Sd = S.double().view(M, nb, N); W1 = wbig1.double(); W2 = wbig2.double()
    pre_ref = torch.einsum("mbk,bkn->mbn", Sd, W1) + bb1.double()[None]
    g = torch.nn.functional.gelu(pre_ref)
    y_ref = (torch.einsum("mbk,bkn->mbn", g, W2) + bb2.double()[None]).reshape(M, nb * N)
    pre_ref = pre_ref.reshape(M, nb * N)
    out = []
    for name, lay, wa, wb in (("fp32 3-product", 1, f1, f2), ("bf16x6", 2, it1.p6[0], it2.p6[0])):
        Y, P, Mid = ops.afno_mlp2(S, wa, bb1, wb, bb2, nb, bs, 1, mode=0, want_pre=True, want_mid=True, layout=lay)
        t = timeit(lambda: ops.afno_mlp2(S, wa, bb1, wb, bb2, nb, bs, 1, mode=0, want_pre=True, want_mid=False, layout=lay))
        out.append(f"{name}: fwd {t:6.1f} us  err Y {rel(Y, y_ref):.2e} pre {rel(P, pre_ref):.2e} mid {rel(Mid, g.reshape(M, nb * N)):.2e}")
    # backward data: mid = (X W2^T) * act'(aux), Y = mid W1^T, pre-out = act(aux)
    ad = pre.double().view(M, nb, N).requires_grad_(True)
    ga = torch.nn.functional.gelu(ad)
    dact = torch.autograd.grad(ga.sum(), ad)[0]
    t1 = torch.einsum("mbn,bkn->mbk", Sd, W2) * dact
    ds_ref = torch.einsum("mbn,bkn->mbk", t1, W1).reshape(M, nb * N)
    for name, lay, wa, wb in (("fp32 3-product", 1, k2, k1), ("bf16x6", 2, it2.p6[1], it1.p6[1])):
        Y, P, Mid = ops.afno_mlp2(S, wa, None, wb, None, nb, bs, 1, mode=1, aux=pre, want_pre=True, want_mid=True, layout=lay)
        t = timeit(lambda: ops.afno_mlp2(S, wa, None, wb, None, nb, bs, 1, mode=1, aux=pre, want_pre=True, want_mid=True, layout=lay))
        out.append(f"{name}: bwd {t:6.1f} us  err dS {rel(Y, ds_ref):.2e} dO1pre {rel(Mid, t1.reshape(M, nb * N)):.2e} O1 {rel(P, ga.detach().reshape(M, nb * N)):.2e}")
    print(f"nb={nb} bs={bs} M={M}:\n  " + "\n  ".join(out), flush=True)

if __name__ == "__main__":
    run(4, 128, 4608)          # DPOT-Tiny, batch 32
    run(8, 128, 4608)          # DPOT-S / -M
    run(16, 96, 8704)          # DPOT-L, batch 16
    run(8, 128, 4608 - 37)     # ragged last row tile
    run(16, 96, 333)
