"""where the one-launch AFNO layer (csrc/afno_fused.hip) spends its time: needs the -DAF_TIMING variant build
(scripts/build_variant_src.sh aftiming afno_fused -DAF_TIMING; DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_aftiming.so),
whose waves 0 and 7 leave shader-clock stamps in y1.  Prints the median duration of every phase over the workgroups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops

os.environ["DPOT_AFNO_LAYER"] = "1"
NAMES = ["x loads", "rfft2 (registers)", "GN1 statistics (barrier) + scale", "publish S (+ global S) + barrier",
         "layer 1 MFMA", "barrier + pre store + GELU + publish + barrier", "layer 2 MFMA", "irfft2 + x reload + GN2 row sums",
         "GN2 statistics (barrier) + stores issued", "store drain"]


def run(E, nb, B, train):
    h, mx, my = 16, 16, 9
    bs = E // nb
    x = torch.randn(B, h * h, E, device="cuda")
    g1, b1 = torch.rand(E, device="cuda") + 0.5, torch.randn(E, device="cuda") * 0.1
    pk = ops.AfnoPacks([(torch.randn(2, nb, bs, bs, device="cuda") * 0.05, torch.randn(2, nb, bs, device="cuda") * 0.1),
                        (torch.randn(2, nb, bs, bs, device="cuda") * 0.05, torch.randn(2, nb, bs, device="cuda") * 0.1)])
    p = pk.refresh()
    for _ in range(3):
        out = ops.afno_fused_fwd(x, g1, b1, p[0][2], p[0][1], p[1][2], p[1][1], g1, b1, h, h, nb, mx, my, 1, save=train)
    torch.cuda.synchronize()
    y1 = out[2]
    st = y1.view(B, h * h, E)[:, 0, :].contiguous().view(B, nb, bs).view(torch.int64).view(B * nb, bs // 2)[:, :24].cpu()
    for w, nm in ((0, "wave 0"), (12, "wave 7")):
        t = st[:, w:w + 11]
        d = (t[:, 1:] - t[:, :-1]).double()
        med = d.median(dim=0).values
        tot = (t[:, 10] - t[:, 0]).double().median().item()
        print(f"E={E} nb={nb} B={B} train={train} {nm}: total {tot:9.0f} cycles; " +
              " | ".join(f"{n}: {v:7.0f}" for n, v in zip(NAMES, med.tolist())), flush=True)
    span = (st[:, [10, 22]].max() - st[:, [0, 12]].min()).item()
    print(f"   first start -> last end over all workgroups: {span} cycles")


if __name__ == "__main__":
    for E, nb, B in ((1024, 8, 32), (512, 4, 32)):
        for train in (True, False):
            run(E, nb, B, train)
