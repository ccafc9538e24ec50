"""SURVEY 8 f4 A/B on ONE box: one AFNO layer's forward as ONE launch (csrc/afno_fused.hip) against the three launches it
replaces (gn_rfft2 -> afno_mlp3 -> irfft2_gn), training form (S / pre-activation / y1 / xn2 written) and inference form
(xn2 only), DPOT-Tiny (E = 512, nb = 4) and DPOT-S/M (E = 1024, nb = 8) at several batches.  hipGraph of `reps` layers,
event timed; every layer of the graph works on its own buffers (cold operands, as inside the model)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops

os.environ["DPOT_TUNE"] = "afno_layer=1"


def timeit(fns, reps):
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
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return min(ts)


def main():
    h, mx, my, act = 16, 16, 9, 1
    reps = 24
    for E, nb, Bs in ((512, 4, (32, 64)), (1024, 8, (16, 32, 64))):
        bs = E // nb
        for B in Bs:
            nbuf = 6
            xs = [torch.randn(B, h * h, E, device="cuda") for _ in range(nbuf)]
            g1, b1 = torch.rand(E, device="cuda") + 0.5, torch.randn(E, device="cuda") * 0.1
            g2, b2 = torch.rand(E, device="cuda") + 0.5, torch.randn(E, device="cuda") * 0.1
            pks = []
            for i in range(nbuf):
                pk = ops.AfnoPacks([(torch.randn(2, nb, bs, bs, device="cuda") * 0.05, torch.randn(2, nb, bs, device="cuda") * 0.1),
                                    (torch.randn(2, nb, bs, bs, device="cuda") * 0.05, torch.randn(2, nb, bs, device="cuda") * 0.1)])
                pks.append((pk, pk.refresh()))

            def three(i, train):
                x, p = xs[i], pks[i][1]
                def f():
                    S, m1, r1 = ops.gn_rfft2(x, g1, b1, h, h, nb, mx, my)
                    O2, pre, _ = ops.afno_mlp2(S, p[0][2], p[0][1], p[1][2], p[1][1], nb, bs, act, mode=0, want_pre=train, layout=1)
                    return ops.irfft2_gn(O2, x, m1, r1, g1, b1, g2, b2, h, h, nb, mx, my)
                return f

            def one(i, train):
                x, p = xs[i], pks[i][1]
                return lambda: ops.afno_fused_fwd(x, g1, b1, p[0][2], p[0][1], p[1][2], p[1][1], g2, b2, h, h, nb, mx, my, act,
                                                  save=train, want_y1=train)

            def mix(i):
                x, p = xs[i], pks[i][1]
                S = torch.randn(B * mx * my, 2 * E, device="cuda")
                return lambda: ops.afno_mlp2(S, p[0][2], p[0][1], p[1][2], p[1][1], nb, bs, act, mode=0, want_pre=True, layout=1)

            t3 = timeit([three(i, True) for i in range(nbuf)], reps)
            t1 = timeit([one(i, True) for i in range(nbuf)], reps)
            t3i = timeit([three(i, False) for i in range(nbuf)], reps)
            t1i = timeit([one(i, False) for i in range(nbuf)], reps)
            tm = timeit([mix(i) for i in range(nbuf)], reps)
            fl = 2 * 4 * 2.0 * B * mx * my * bs * bs * nb
            print(f"E={E} nb={nb} B={B} ({B * nb} workgroups): train  three launches {t3:6.1f} us | one launch {t1:6.1f} us "
                  f"({fl / t1 / 1e6:5.1f} TF algorithmic) || inference  three {t3i:6.1f} | one {t1i:6.1f} || mixer launch alone {tm:6.1f}",
                  flush=True)


if __name__ == "__main__":
    main()
