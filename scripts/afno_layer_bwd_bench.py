"""A/B on ONE box: the backward of one AFNO layer (GroupNorm2 backward ... GroupNorm1 backward + outer skip, data path only -
the weight-gradient launch is the same on both sides) as ONE launch (csrc/afno_fused.hip afno_fused_bwd_kernel) against the
launches it replaces: gn_bwd_rfft2 -> afno_mlp3 (mode 1) -> irfft2 + groupnorm_bwd (128 channels per group: DPOT-S / -M) or
-> irfft2_gn_bwd (64 channels per group: DPOT-Tiny).  hipGraph of `reps` layers over rotating buffers, event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops

os.environ["DPOT_AFNO_LAYER"] = "1"


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
            Mm = B * mx * my
            bufs = []
            for i in range(nbuf):
                x = torch.randn(B, h * h, E, device="cuda")
                g1, b1 = torch.rand(E, device="cuda") + 0.5, torch.randn(E, device="cuda") * 0.1
                g2, b2 = torch.rand(E, device="cuda") + 0.5, torch.randn(E, device="cuda") * 0.1
                pk = ops.AfnoPacks([(torch.randn(2, nb, bs, bs, device="cuda") * 0.05, torch.randn(2, nb, bs, device="cuda") * 0.1),
                                    (torch.randn(2, nb, bs, bs, device="cuda") * 0.05, torch.randn(2, nb, bs, device="cuda") * 0.1)])
                p = pk.refresh()
                S, pre, y1, xn2, m1, r1, m2, r2 = ops.afno_fused_fwd(x, g1, b1, p[0][2], p[0][1], p[1][2], p[1][1], g2, b2, h, h, nb,
                                                                     mx, my, act)
                bufs.append(dict(x=x, g1=g1, g2=g2, pk=pk, p=p, pre=pre, y1=y1, m1=m1, r1=r1, m2=m2, r2=r2,
                                 dxn2=torch.randn(B, h * h, E, device="cuda"), dout=torch.randn(B, h * h, E, device="cuda")))

            def multi(d):
                def f():
                    dy1, part2, dO2 = ops.gn_bwd_rfft2(d["dxn2"], d["y1"], d["m2"], d["r2"], d["g2"], h, h, nb, mx, my, col_weights=1)
                    dS, O1, dPre = ops.afno_mlp2(dO2, d["p"][1][3], None, d["p"][0][3], None, nb, bs, act, mode=1, aux=d["pre"],
                                                 want_mid=True, want_pre=True, layout=1)
                    if E // 8 <= 64:
                        return ops.irfft2_gn_bwd(dS, dy1, d["x"], d["m1"], d["r1"], d["g1"], h, h, nb, mx, my, add=d["dout"], col_weights=0)
                    dxn1 = ops.irfft2(dS, B, h, h, E, nb, mx, my, 0, res=dy1)
                    return ops.groupnorm_bwd(dxn1, d["x"], d["m1"], d["r1"], d["g1"], add=d["dout"], defer=True)
                return f

            def one(d):
                return lambda: ops.afno_fused_bwd(d["dxn2"], d["y1"], d["m2"], d["r2"], d["g2"], d["pre"], d["p"][1][3], d["p"][0][3],
                                                  d["x"], d["m1"], d["r1"], d["g1"], d["dout"], h, h, nb, mx, my, act)

            tm = timeit([multi(d) for d in bufs], reps)
            t1 = timeit([one(d) for d in bufs], reps)
            fl = 2 * 4 * 2.0 * B * mx * my * bs * bs * nb
            print(f"E={E} nb={nb} B={B} ({B * nb} workgroups): backward data path  separate launches {tm:6.1f} us | one launch {t1:6.1f} us "
                  f"({fl / t1 / 1e6:5.1f} TF algorithmic)", flush=True)


if __name__ == "__main__":
    main()
