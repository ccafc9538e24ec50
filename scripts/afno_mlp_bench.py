"""micro-benchmark of the fused AFNO MLP kernel (csrc/afno_mlp.hip): hipGraph of 50 launches, event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops

def bench(nb, bs, M, reps=50):
    N = 2 * bs
    S = torch.randn(M, nb * N, device="cuda")
    W1 = torch.randn(nb, N, N, device="cuda") * 0.05
    W2 = torch.randn(nb, N, N, device="cuda") * 0.05
    b1 = torch.randn(nb, N, device="cuda") * 0.1
    b2 = torch.randn(nb, N, device="cuda") * 0.1
    W1f, W1b = ops.afno_block_weights(W1)
    W2f, W2b = ops.afno_block_weights(W2)
    pre = torch.randn(M, nb * N, device="cuda")
    forms = {"fwd-train": lambda: ops.afno_mlp2(S, W1f, b1, W2f, b2, nb, bs, 1, mode=0, want_pre=True, want_mid=True),
             "fwd-infer": lambda: ops.afno_mlp2(S, W1f, b1, W2f, b2, nb, bs, 1, mode=0),
             "bwd-data": lambda: ops.afno_mlp2(S, W2b, None, W1b, None, nb, bs, 1, mode=1, aux=pre, want_mid=True)}
    if ops.afno_mlp3_supported(nb, bs):
        # the three-product kernel on (Wr, Wi) fragment packs
        w1 = torch.randn(2, nb, bs, bs, device="cuda") * 0.05; w2 = torch.randn(2, nb, bs, bs, device="cuda") * 0.05
        c1 = torch.randn(2, nb, bs, device="cuda") * 0.1; c2 = torch.randn(2, nb, bs, device="cuda") * 0.1
        pk = ops.AfnoPacks([(w1, c1), (w2, c2)])
        (_, bb1, f1, k1), (_, bb2, f2, k2) = pk.refresh()
        forms.update({"3mult fwd-train": lambda: ops.afno_mlp2(S, f1, bb1, f2, bb2, nb, bs, 1, mode=0, want_pre=True, want_mid=False, layout=1),
                      "3mult fwd-train(r2: +mid)": lambda: ops.afno_mlp2(S, f1, bb1, f2, bb2, nb, bs, 1, mode=0, want_pre=True, want_mid=True, layout=1),
                      "3mult fwd-infer": lambda: ops.afno_mlp2(S, f1, bb1, f2, bb2, nb, bs, 1, mode=0, layout=1),
                      "3mult bwd-data": lambda: ops.afno_mlp2(S, k2, None, k1, None, nb, bs, 1, mode=1, aux=pre, want_mid=True, want_pre=True, layout=1),
                      "3mult bwd-data(r2: no act out)": lambda: ops.afno_mlp2(S, k2, None, k1, None, nb, bs, 1, mode=1, aux=pre, want_mid=True, layout=1)})
    flops = 2 * 2.0 * M * N * N * nb
    out = []
    for name, fn in forms.items():
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
        t = e0.elapsed_time(e1) * 1e-3 / reps
        out.append(f"{name} {t*1e6:7.1f} us {flops/t/1e12:6.1f} TF")
    print(f"nb={nb} bs={bs} M={M}: " + " | ".join(out), flush=True)

def tiny_train(n=30):
    """the DPOT-Tiny (B=32) mixer launch in its training form, eagerly, n times (for rocprofv3 --pmc passes)"""
    nb, bs, M = 4, 128, 4608
    N = 2 * bs
    S = torch.randn(M, nb * N, device="cuda")
    W1f, _ = ops.afno_block_weights(torch.randn(nb, N, N, device="cuda") * 0.05)
    W2f, _ = ops.afno_block_weights(torch.randn(nb, N, N, device="cuda") * 0.05)
    b1 = torch.randn(nb, N, device="cuda") * 0.1
    b2 = torch.randn(nb, N, device="cuda") * 0.1
    if ops.afno_mlp3_supported(nb, bs):            # what the model runs for bs = 128: the three-product kernel
        pk = ops.AfnoPacks([(torch.randn(2, nb, bs, bs, device="cuda") * 0.05, torch.randn(2, nb, bs, device="cuda") * 0.1),
                            (torch.randn(2, nb, bs, bs, device="cuda") * 0.05, torch.randn(2, nb, bs, device="cuda") * 0.1)])
        (_, c1, f1, _), (_, c2, f2, _) = pk.refresh()
        for _ in range(n):
            ops.afno_mlp2(S, f1, c1, f2, c2, nb, bs, 1, mode=0, want_pre=True, want_mid=False, layout=1)
        torch.cuda.synchronize()
        return
    for _ in range(n):
        ops.afno_mlp2(S, W1f, b1, W2f, b2, nb, bs, 1, mode=0, want_pre=True, want_mid=True)
    torch.cuda.synchronize()


def tiny_bwd(n=30):
    """the DPOT-Tiny (B=32) mixer DATA-GRADIENT launch (three-product kernel, mode 1), eagerly, n times"""
    nb, bs, M = 4, 128, 4608
    N = 2 * bs
    S = torch.randn(M, nb * N, device="cuda")
    pre = torch.randn(M, nb * N, device="cuda")
    pk = ops.AfnoPacks([(torch.randn(2, nb, bs, bs, device="cuda") * 0.05, torch.randn(2, nb, bs, device="cuda") * 0.1),
                        (torch.randn(2, nb, bs, bs, device="cuda") * 0.05, torch.randn(2, nb, bs, device="cuda") * 0.1)])
    (_, _, _, k1), (_, _, _, k2) = pk.refresh()
    for _ in range(n):
        ops.afno_mlp2(S, k2, None, k1, None, nb, bs, 1, mode=1, aux=pre, want_mid=True, want_pre=True, layout=1)
    torch.cuda.synchronize()


def fused_fwd(n=30, E=1024, nb=8, B=32):
    """the one-launch AFNO layer (csrc/afno_fused.hip), training form, DPOT-S / -M at batch 32 (256 workgroups)"""
    os.environ["DPOT_TUNE"] = "afno_layer=1"
    h, mx, my, bs = 16, 16, 9, E // nb
    x = torch.randn(B, h * h, E, device="cuda")
    g1, b1 = torch.rand(E, device="cuda") + 0.5, torch.randn(E, device="cuda") * 0.1
    pk = ops.AfnoPacks([(torch.randn(2, nb, bs, bs, device="cuda") * 0.05, torch.randn(2, nb, bs, device="cuda") * 0.1),
                        (torch.randn(2, nb, bs, bs, device="cuda") * 0.05, torch.randn(2, nb, bs, device="cuda") * 0.1)])
    p = pk.refresh()
    for _ in range(n):
        ops.afno_fused_fwd(x, g1, b1, p[0][2], p[0][1], p[1][2], p[1][1], g1, b1, h, h, nb, mx, my, 1, save=True, want_y1=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "tiny-train":
        tiny_train()
    elif len(sys.argv) > 1 and sys.argv[1] == "tiny-bwd":
        tiny_bwd()
    elif len(sys.argv) > 1 and sys.argv[1] == "fused-fwd":
        fused_fwd()
    else:
        for nb, bs, M in ((4, 128, 4608), (8, 128, 2304), (16, 96, 2176), (16, 96, 8704)):
            bench(nb, bs, M)
