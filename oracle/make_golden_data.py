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


# test-mode / down-sampling cases (griddataset.py:159-174): (H, W, T, C), res, n_channels, t_in, t_test, downsample,
# pred_channels (None: all the dataset's channels)
TEST_CASES = [
    ((16, 16, 12, 1), 32, 4, 4, 6, (1, 1), None),     # test trajectory, mask on every 2nd grid point, 1 of 4 channels
    ((32, 32, 7, 3), 32, 3, 4, 10, (1, 1), 2),        # t_test runs past the trajectory (slice clips), pred_channels 2
    ((48, 40, 9, 2), 24, 4, 3, 4, (2, 3), None),      # target resolution below the data's (k = 0 -> 1), downsample (2, 3)
    ((16, 16, 8, 2), 32, 2, 4, 3, (4, 4), None),      # downsample 4
]


def reference_test_item(sample_raw, res, n_channels, t_in, t_test, down, pred_channels):
    # ---- griddataset.py:143-174 with train = False, verbatim where the lines do not touch files / self.*
    sample = sample_raw
    orig_size = list(sample.shape)
    orig_size[-1] = pred_channels if pred_channels is not None else orig_size[-1]
    sample = reference_pad_data(sample, res, n_channels)
    start_idx = 0
    x, y = sample[..., start_idx:start_idx + t_in, :], sample[..., t_in:t_in + t_test, :]
    # get_target_mask (griddataset.py:103-117)
    msk = torch.zeros(*sample.shape[:2], 1, sample.shape[-1])    ## target mask shape H,W,1,C
    kx, ky = sample.shape[0] // orig_size[0], sample.shape[1] // orig_size[1]
    if kx == 0 or ky == 0:
        kx = 1 if kx == 0 else kx
        ky = 1 if ky == 0 else ky
    msk[::kx, ::ky, :, :orig_size[-1]] = 1
    ### downsample
    if down != (1, 1):
        x, y = x[::down[0], ::down[1]], y[::down[0], ::down[1]]
    return x, y, msk


def main():
    out = {}
    for k, (shape, res, nc, t_in, t_test, down, pc) in enumerate(TEST_CASES):
        raw = D.recipe_sample(shape, salt=200 + k).contiguous()
        x, y, msk = reference_test_item(raw, res, nc, t_in, t_test, down, pc)
        padded = D.pad_data(raw, res, nc)
        xo, yo = D.test_window(padded, t_in, t_test)
        xo, yo = D.downsample(xo, yo, down)
        mo = D.target_mask(padded, list(shape[:3]) + [pc if pc is not None else shape[3]])
        assert torch.equal(x, xo) and torch.equal(y, yo) and torch.equal(msk, mo), f"oracle disagrees on test case {k}"
        out[f"t{k}.x"], out[f"t{k}.y"], out[f"t{k}.msk"] = x.numpy(), y.numpy(), msk.numpy()
        out[f"t{k}.meta"] = np.array(list(shape) + [res, nc, t_in, t_test, down[0], down[1], -1 if pc is None else pc],
                                     dtype=np.int64)
        print(f"test case {k}: raw {shape} -> x {tuple(x.shape)}, y {tuple(y.shape)}, msk {tuple(msk.shape)}")
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
