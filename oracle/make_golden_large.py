#!/usr/bin/env python
"""Golden numbers for BASELINE.json configs[4] from the REFERENCE implementation.  TEST INFRASTRUCTURE ONLY.

DPOT-Large (embed 1536, depth 24, 16 blocks, mlp_ratio 4, out_layer_dim 128) on 256x256 fields, modes 64, a
20-step auto-regressive rollout (configs/pretrain_large.yaml; train_temporal.py:201-230), batch 1, recipe weights
and inputs (oracle.dpot_ref.recipe_*: closed form, rebuilt identically on the GPU box).  The imported reference model
is driven by the reference's own loop body; every AR step runs under torch.utils.checkpoint (autograd memory
management only - the arithmetic is the reference's) because 20 steps of DPOT-L activations do not fit this
container's 62 GB.  Writes tests/golden/g11_large_rollout.npz: loss, total and per-tensor gradient norms (float64
accumulation) and a subsample of the 20-step prediction.  Runs ~10 minutes on 8 cores.

    python oracle/make_golden_large.py [T_ar]
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch
from torch.utils.checkpoint import checkpoint

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("DPOT_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import dpot_ref as R  # noqa: E402
from oracle.make_golden import ref_model, save, sub  # noqa: E402   (imports the reference modules)
from utils.criterion import SimpleLpLoss  # noqa: E402               (reference)


def main():
    T_ar = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = R.DPOTConfig(**R.LARGE)
    sd0 = R.recipe_state_dict(cfg, salt=6)
    m = ref_model(cfg, sd0)
    m.train()
    B, S = 1, cfg.img_size
    xx = R.recipe_input((B, S, S, cfg.in_timesteps, cfg.in_channels), salt=81)
    yy = R.recipe_input((B, S, S, T_ar, cfg.out_channels), salt=82)
    msk = torch.ones(B, S, S, 1, cfg.out_channels)
    crit = SimpleLpLoss(size_average=False)
    t0 = time.time()
    cur, loss, chunks = xx, 0.0, []
    for t in range(T_ar):                                   # train_temporal.py:201-219 (T_bundle = 1, no noise)
        im = checkpoint(lambda x: m(x)[0], cur, use_reentrant=False)
        loss = loss + crit(im, yy[..., t:t + 1, :], mask=msk)
        chunks.append(im.detach())
        cur = torch.cat((cur[..., 1:, :], im), dim=-2)
        print(f"  AR step {t + 1}/{T_ar}  loss so far {float(loss):.6f}  ({time.time() - t0:.0f} s)", flush=True)
    loss.backward()
    pred = torch.cat(chunks, dim=-2)
    names, norms = [], []
    for k, p in m.named_parameters():
        if p.grad is not None:
            names.append(k)
            norms.append(p.grad.double().norm().item())
    total = float(np.sqrt(np.sum(np.square(norms))))
    print(f"  g11: loss={loss.item():.6f} |g|={total:.6e}  ({time.time() - t0:.0f} s)")
    save("g11_large_rollout", loss=np.float64(loss.item()), grad_norm=np.float64(total), names=np.array(names),
         grad_norms=np.array(norms, dtype=np.float64), pred=sub(pred, 97), T_ar=np.int64(T_ar), B=np.int64(B))


if __name__ == "__main__":
    main()
