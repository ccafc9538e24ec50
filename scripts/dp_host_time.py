"""Host cost of the data-parallel step's driver on ONE process: SegmentedTrainStep (the hipGraph chain cut at the gradient
buckets, dpot_amd/train.py) at DPOT-Tiny, batch 32, with the reducer in dry-run mode (every bucket's stream hand-off and one
device operation on the side stream, no collective - there is no second GPU here).  Prints the wall time per step, the host
time spent inside replay() and the number of graph launches / bucket hand-offs per step; compare with GraphedTrainStep (one
graph, no buckets).  The host time must stay well below the step time, or the GPU starves at N > 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import DPOTNet
from dpot_amd.dp import BucketedGradReducer
from dpot_amd.train import FlatParams, FusedAdam, GraphedTrainStep, SegmentedTrainStep
from bench import TINY

torch.manual_seed(0)
B = 32
xx = torch.randn(B, 128, 128, 10, 4, device="cuda"); yy = torch.randn(B, 128, 128, 1, 4, device="cuda")
msk = torch.ones(B, 128, 128, 1, 4, device="cuda")


def run(kind, n_buckets=4, steps=200):
    model = DPOTNet(**TINY).cuda()
    fp = FlatParams(model)
    opt = FusedAdam(fp, lr=1e-3, betas=(0.9, 0.9), weight_decay=1e-6, max_norm=1e4, update_tail=kind != "single")
    if kind == "single":
        g = GraphedTrainStep(model, opt, xx, yy, msk, noise_scale=0.0005)
        nseg, nb = 1, 0
    else:
        red = BucketedGradReducer(fp, n_buckets=n_buckets, overlap=True)       # n_buckets None: dp.auto_n_buckets (by bytes)
        red.dry_run = True
        g = SegmentedTrainStep(model, opt, red, xx, yy, msk, noise_scale=0.0005)
        nseg, nb = len(g.graphs) + 1, red.n_buckets
    for _ in range(20):
        g.replay(1e-4)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        th = time.perf_counter()
        g.replay(1e-4)
        host += time.perf_counter() - th
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    # host-only rate: the same calls with the GPU idle at the start of each (sync before, not timed)
    hs = 0.0
    for _ in range(50):
        torch.cuda.synchronize()
        th = time.perf_counter()
        g.replay(1e-4)
        hs += time.perf_counter() - th
    torch.cuda.synchronize()
    print(f"{kind:<22s} graphs/step {nseg}  buckets {nb}  wall {wall / steps * 1e3:.3f} ms/step  host inside replay() "
          f"{host / steps * 1e6:.0f} us/step (queue full)  {hs / 50 * 1e6:.0f} us/step (GPU idle at call)", flush=True)


run("single")
run("segmented, auto buckets", None)
for nb in (2, 4, 8):
    run(f"segmented, {nb} buckets", nb)
