#!/usr/bin/env python
"""Golden numbers for DPOT-Large at the BENCHMARKED per-GPU batch (16) from the REFERENCE implementation.
TEST INFRASTRUCTURE ONLY.

bench.py --config L quotes DPOT-L (embed 1536, depth 24, 16 blocks, mlp_ratio 4, out_layer_dim 128, 256x256, modes 64;
configs/pretrain_large.yaml:63-87) at batch 16; the kernel selection there (two-workgroup bf16 GEMM for every launch
with >= 512 tiles, 128 x 192 tiles, pair-grid rules, panel heights, split-K factors) depends on the batch, so the
parity suite needs a reference answer AT that batch.  50 TFLOP of CPU work - too slow to run live in a GPU test and
too large for this container's 62 GB in one piece, hence a committed fixture computed in micro-batches:

    loss_like = sum_b <y_b, up_y_b> + <cls_b, up_c_b>     (the samples of a batch are independent: models/dpot.py:364-403
                                                           has no cross-sample operation with normalize = False)

so the parameter gradient of the batch is the SUM of the micro-batch gradients, which is how autograd accumulates
`.grad` over several backward() calls of the imported reference model.  Inputs / weights / upstream gradients are the
recipe tensors of tests/test_gpu_sizes.py::_oracle_case("LARGE", 16) (salts 71 / 72 / 73, weights salt 4).

Writes tests/golden/g13_large_b16.npz: prediction and dx subsamples + checksums, cls, and for EVERY parameter the
float64 gradient norm plus a strided subsample of the gradient.  ~15 minutes on 8 cores.

    python oracle/make_golden_large_b16.py [B] [micro]
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("DPOT_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import dpot_ref as R  # noqa: E402
from oracle.make_golden import ref_model, save, sub  # noqa: E402   (imports the reference modules)

GRAD_SUB = 1024            # elements kept per parameter gradient


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    micro = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    name = sys.argv[3] if len(sys.argv) > 3 else "LARGE"
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = R.DPOTConfig(**getattr(R, name))
    m = ref_model(cfg, R.recipe_state_dict(cfg, salt=4))
    m.train()
    S = cfg.img_size
    x = R.recipe_input((B, S, S, cfg.in_timesteps, cfg.in_channels), salt=71)
    up_y = R.recipe_input((B, S, S, cfg.out_timesteps, cfg.out_channels), salt=72) * 0.3
    up_c = R.recipe_input((B, cfg.n_cls), salt=73) * 0.3
    t0 = time.time()
    ys, cs, dxs = [], [], []
    for lo in range(0, B, micro):
        xo = x[lo:lo + micro].clone().requires_grad_(True)
        y, c = m(xo)                                              # the reference forward (models/dpot.py:364-403)
        ((y * up_y[lo:lo + micro]).sum() + (c * up_c[lo:lo + micro]).sum()).backward()
        ys.append(y.detach()), cs.append(c.detach()), dxs.append(xo.grad)
        print(f"  samples {lo}..{lo + micro - 1} done ({time.time() - t0:.0f} s)", flush=True)
    y, c, dx = torch.cat(ys), torch.cat(cs), torch.cat(dxs)
    out = dict(B=np.int64(B), y=sub(y, 1009), c=c.numpy(), dx=sub(dx, 4099))
    names, norms = [], []
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        names.append(k)
        norms.append(p.grad.double().norm().item())
        out[f"g/{k}"] = sub(p.grad, max(1, p.numel() // GRAD_SUB))
    out["names"] = np.array(names)
    out["grad_norms"] = np.array(norms, dtype=np.float64)
    print(f"  g13: |g| = {float(np.sqrt(np.sum(np.square(norms)))):.6e}  ({time.time() - t0:.0f} s)")
    save(f"g13_{name.lower()}_b{B}", **out)


if __name__ == "__main__":
    main()
