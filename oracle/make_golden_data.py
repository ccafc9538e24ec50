#!/usr/bin/env python
"""Golden vectors of the input pipeline's per-sample transform.  TEST INFRASTRUCTURE ONLY.

utils/griddataset.py cannot be imported in this image (h5py is missing), so - like the train_temporal.py loop body in
make_golden.py - the lines of MixedTemporalDataset.pad_data (griddataset.py:94-101) and of the training window
(:150-153) are transcribed literally below and run on recipe samples; the oracle (oracle/data_ref.py) must agree before
anything is written.  Writes tests/golden/g12_data.npz.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import data_ref as D  # noqa: E402

CASES = [  # (H, W, T, C), res, n_channels, t_in, t_ar, t0
    ((16, 16, 8, 1), 32, 4, 4, 2, 1),          # 2x up-sampling, 1 -> 4 channels (the ns2d_fno pattern, scaled down)
    ((32, 32, 7, 3), 32, 3, 4, 1, 2),          # identity resolution, no channel padding
    ((48, 40, 6, 2), 24, 4, 3, 2, 0),          # down-sampling, non-square source
    ((13, 21, 9, 4), 32, 5, 5, 3, 1),          # odd sizes, non-integer ratios
]


def reference_pad_data(x, res, n_channels):
    # ---- griddataset.py:94-101, verbatim (self.res -> res, self.n_channels -> n_channels)
    H, W, T, C = x.shape
    x = x.view(H, W, -1).permute(2, 0, 1)  # Cmax, H, W
    x = F.interpolate(x.unsqueeze(0), size=(res, res), mode='bilinear').squeeze(0).permute(1, 2, 0)
    x = x.view(*x.shape[:2], T, C)
    x_new = torch.ones([*x.shape[:-1], n_channels])
    x_new[..., :x.shape[-1]] = x  # H, W, T, Cmax
    return x_new


def main():
    out = {}
    for k, (shape, res, nc, t_in, t_ar, t0) in enumerate(CASES):
        raw = D.recipe_sample(shape, salt=100 + k).contiguous()
        sample = reference_pad_data(raw, res, nc)
        # ---- griddataset.py:152 with start_idx = t0
        x, y = sample[..., t0: t0 + t_in, :], sample[..., t0 + t_in: min(t0 + t_in + t_ar, sample.shape[-2]), :]
        xo, yo = D.window(D.pad_data(raw, res, nc), t0, t_in, t_ar)
        assert torch.equal(x, xo) and torch.equal(y, yo), f"oracle disagrees with the reference lines on case {k}"
        out[f"c{k}.x"], out[f"c{k}.y"] = x.numpy(), y.numpy()
        out[f"c{k}.meta"] = np.array(list(shape) + [res, nc, t_in, t_ar, t0], dtype=np.int64)
        print(f"case {k}: raw {shape} -> x {tuple(x.shape)}, y {tuple(y.shape)}")
    path = os.path.join(ROOT, "tests", "golden", "g12_data.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


if __name__ == "__main__":
    main()
