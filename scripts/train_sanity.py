#!/usr/bin/env python
"""does it train?  DPOT-Tiny on a synthetic but learnable task (predict the last input frame of smooth random fields),
300 graph-replayed steps with the fused optimiser; prints the relative-L2 loss per sample-channel every 50 steps.
Usage: python scripts/train_sanity.py [f32|auto|bf16x6]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpot_amd import DPOTNet, ops                                        # noqa: E402
from dpot_amd.train import FlatParams, FusedAdam, GraphedTrainStep, one_cycle_lr  # noqa: E402

TINY = dict(img_size=128, patch_size=8, in_channels=4, out_channels=4, in_timesteps=10, out_timesteps=1, n_blocks=4,
            embed_dim=512, out_layer_dim=32, depth=4, modes=32, mlp_ratio=1, n_cls=12)


def fields(B, seed):
    """smooth travelling waves: sum of 6 low-frequency sinusoids per channel, advected in time"""
    g = torch.Generator().manual_seed(seed)
    x = torch.linspace(0, 1, 128)
    X, Y = torch.meshgrid(x, x, indexing="ij")
    out = torch.zeros(B, 128, 128, 11, 4)
    for b in range(B):
        for c in range(4):
            for _ in range(6):
                kx, ky = torch.randint(1, 5, (2,), generator=g).tolist()
                ph, sp = torch.rand(2, generator=g).tolist()
                amp = torch.randn(1, generator=g).item()
                for t in range(11):
                    out[b, :, :, t, c] += amp * torch.sin(2 * math.pi * (kx * X + ky * Y + ph + 0.03 * sp * t))
    return out


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
    ops.set_gemm_precision(prec)
    torch.manual_seed(0)
    model = DPOTNet(**TINY).cuda()
    opt = FusedAdam(FlatParams(model), lr=1e-3, betas=(0.9, 0.9), weight_decay=1e-6, max_norm=10000.0)
    B, steps = 32, 300
    data = [fields(B, s) for s in range(4)]                              # 4 fixed batches, cycled
    xx = data[0][..., :10, :].contiguous().cuda()
    yy = data[0][..., 10:, :].contiguous().cuda()
    msk = torch.ones(B, 128, 128, 1, 4, device="cuda")
    g = GraphedTrainStep(model, opt, xx, yy, msk, noise_scale=0.0005, warmup=1)
    hist = []
    for it in range(steps):
        d = data[it % 4]
        g.stage(d[..., :10, :].contiguous().cuda(), d[..., 10:, :].contiguous().cuda(), msk)
        loss = g.replay(one_cycle_lr(it, steps, 1e-3, pct_start=0.2))
        if it % 50 == 0 or it == steps - 1:
            hist.append((it, loss.item() / (B * 4)))
    print(prec, " ".join(f"{i}:{v:.4f}" for i, v in hist), flush=True)
    assert all(math.isfinite(v) for _, v in hist) and hist[-1][1] < 0.5 * hist[0][1], "loss did not go down"


if __name__ == "__main__":
    main()
