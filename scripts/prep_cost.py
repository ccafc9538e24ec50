"""What do the weight-only prep launches cost INSIDE the replayed graph?  DPOT-Tiny, batch 32: the normal step against a
step whose derived weights (embed fold, AFNO / panel packs, head layouts: ~20 launches, ~126 us as eager kernel time) are
computed once OUTSIDE the graph (stale after the first update - a timing experiment, not a training mode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dpot_amd import DPOTNet
from dpot_amd.train import FlatParams, FusedAdam, GraphedTrainStep
from bench import TINY

B = 32
xx = torch.randn(B, 128, 128, 10, 4, device="cuda"); yy = torch.randn(B, 128, 128, 1, 4, device="cuda")
msk = torch.ones(B, 128, 128, 1, 4, device="cuda")


def run(stale, noise=0.0005, steps=200):
    torch.manual_seed(0)
    model = DPOTNet(**TINY).cuda()
    opt = FusedAdam(FlatParams(model), lr=1e-3, betas=(0.9, 0.9), weight_decay=1e-6, max_norm=1e4)
    if stale:
        cached = model._derived_weights()
        model._derived_weights = lambda: cached
    g = GraphedTrainStep(model, opt, xx, yy, msk, noise_scale=noise)
    for _ in range(30):
        g.replay(1e-4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay(1e-4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(2):
    a, b, c = run(False), run(True), run(False, noise=0.0)
    print(f"normal {a:.4f} ms   prep outside the graph {b:.4f} ms (prep costs {1e3 * (a - b):.0f} us in replay)   "
          f"no noise injection {c:.4f} ms (noise costs {1e3 * (a - c):.0f} us)", flush=True)
