#!/usr/bin/env python
"""Golden vectors of the input pipeline's per-sample transform.  TEST INFRASTRUCTURE ONLY.

The vectors come from the reference's OWN `utils.griddataset.MixedTemporalDataset.__getitem__` (griddataset.py:125-175,
which calls `pad_data` :88-101 and `get_target_mask` :103-117), imported from /root/reference and run here:

  * `h5py` is absent from this image, so an in-memory stand-in module is put into `sys.modules` BEFORE the import; its
    `File(path)['data'][...]` hands back recipe trajectories (the reference only ever does `File(path, 'r')['data'][idx]`
    / `[:]` on it, griddataset.py:64,74,143).  No reference arithmetic is replaced: resize, padding, window, mask and
    down-sampling all execute in the reference's code;
  * `utils/make_master_file.py:324` writes a CSV into the CWD at import -> the import runs from a temporary directory;
  * synthetic entries are added to the reference's `DATASET_DICT` (sizes, `t_test`, `downsample`, `pred_channels`,
    `scatter_storage`) - that table is the reference's own configuration mechanism for a new dataset;
  * the random window start of training items (`np.random.randint`, :151) is pinned by seeding and re-deriving it.

The oracle (oracle/data_ref.py) must agree bit for bit before anything is written.  Writes tests/golden/g12_data.npz.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import data_ref as D  # noqa: E402

CASES = [  # (H, W, T, C), res, n_channels, t_in, t_ar, t0
    ((16, 16, 8, 1), 32, 4, 4, 2, 1),          # 2x up-sampling, 1 -> 4 channels (the ns2d_fno pattern, scaled down)
    ((32, 32, 7, 3), 32, 3, 4, 1, 2),          # identity resolution, no channel padding
    ((48, 40, 6, 2), 24, 4, 3, 2, 0),          # down-sampling, non-square source
    ((13, 21, 9, 4), 32, 5, 5, 3, 1),          # odd sizes, non-integer ratios
]

# test-mode / down-sampling cases (griddataset.py:159-174): (H, W, T, C), res, n_channels, t_in, t_test, downsample,
# pred_channels (None: all the dataset's channels)
TEST_CASES = [
    ((16, 16, 12, 1), 32, 4, 4, 6, (1, 1), None),     # test trajectory, mask on every 2nd grid point, 1 of 4 channels
    ((32, 32, 7, 3), 32, 3, 4, 10, (1, 1), 2),        # t_test runs past the trajectory (slice clips), pred_channels 2
    ((48, 40, 9, 2), 24, 4, 3, 4, (2, 3), None),      # target resolution below the data's (k = 0 -> 1), downsample (2, 3)
    ((16, 16, 8, 2), 32, 2, 4, 3, (4, 4), None),      # downsample 4
]

_STORE = {}          # path -> numpy array [n_samples, H, W, T, C]   (what the stand-in h5py serves)


class _FakeH5File(dict):
    def __init__(self, path, mode="r"):
        super().__init__(data=_STORE[path])


def import_reference_dataset():
    """-> (MixedTemporalDataset, DATASET_DICT) of the reference, imported with the stand-in h5py from a temp CWD"""
    fake = types.ModuleType("h5py")
    fake.File = _FakeH5File
    sys.modules["h5py"] = fake
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            from utils.griddataset import MixedTemporalDataset
            from utils.make_master_file import DATASET_DICT
        finally:
            os.chdir(cwd)
    return MixedTemporalDataset, DATASET_DICT


def register(DATASET_DICT, name, raw, n_channels, t_test, down, pred_channels, scatter):
    """one synthetic dataset holding the single trajectory `raw` (stored 3 times: index 1 is the one read)"""
    path = f"/synthetic/{name}"
    arr = np.stack([np.zeros_like(raw), raw, np.zeros_like(raw)])
    entry = {"train_path": path, "test_path": path, "train_size": 3, "test_size": 3, "scatter_storage": scatter,
             "t_test": t_test, "t_in": 10, "in_size": tuple(raw.shape[:2]), "n_channels": n_channels,
             "downsample": down}
    if pred_channels is not None:
        entry["pred_channels"] = pred_channels
    if scatter:                                   # griddataset.py:63-64: one file per sample, whole 'data' array
        for i in range(3):
            _STORE[f"{path}/data_{i}.hdf5"] = arr[i]
    else:
        _STORE[path] = arr
    DATASET_DICT[name] = entry


def main():
    MixedTemporalDataset, DATASET_DICT = import_reference_dataset()
    out = {}
    for k, (shape, res, nc, t_in, t_test, down, pc) in enumerate(TEST_CASES):
        raw = D.recipe_sample(shape, salt=200 + k).contiguous()
        rawnp = raw.numpy() if shape[3] > 1 else raw.numpy()[..., 0]       # 3-D storage exercises the `ndim == 3` branch (:145)
        name = f"g12_test{k}"
        register(DATASET_DICT, name, rawnp, nc, t_test, down, pc, scatter=(k % 2 == 1))
        ds = MixedTemporalDataset([name], res=res, t_in=t_in, t_ar=1, n_channels=nc, train=False)
        x, y, msk, idx_cls = ds[1]
        assert idx_cls.tolist() == [0]
        padded = D.pad_data(raw, res, nc)
        assert torch.equal(ds.pad_data(raw.clone()), padded), f"oracle pad_data disagrees on test case {k}"
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
        name = f"g12_train{k}"
        register(DATASET_DICT, name, raw.numpy(), nc, 1, (1, 1), None, scatter=(k % 2 == 0))
        ds = MixedTemporalDataset([name], res=res, t_in=t_in, t_ar=t_ar, n_channels=nc, train=True)
        # griddataset.py:151: start_idx = np.random.randint(max(T - (t_in + t_ar) + 1, 1)); find a seed that draws t0
        hi = max(shape[2] - (t_in + t_ar) + 1, 1)
        seed = next(s for s in range(10000) if np.random.RandomState(s).randint(hi) == t0)
        np.random.seed(seed)
        x, y, msk, idx_cls = ds[1]
        assert torch.equal(msk, torch.ones(res, res, 1, nc))                 # :153
        xo, yo = D.window(D.pad_data(raw, res, nc), t0, t_in, t_ar)
        assert torch.equal(x, xo) and torch.equal(y, yo), f"oracle disagrees with the reference on case {k}"
        out[f"c{k}.x"], out[f"c{k}.y"] = x.numpy(), y.numpy()
        out[f"c{k}.meta"] = np.array(list(shape) + [res, nc, t_in, t_ar, t0], dtype=np.int64)
        print(f"case {k}: raw {shape} -> x {tuple(x.shape)}, y {tuple(y.shape)} (seed {seed} -> start {t0})")
    # dataset mixing (griddataset.py:133-141): two datasets, data_weights (1, 2): global index -> (dataset, local index)
    ds = MixedTemporalDataset(["g12_train0", "g12_train1"], n_list=[3, 3], res=32, t_in=4, t_ar=1, n_channels=4,
                              train=True, data_weights=[1, 2])
    out["mix.len"] = np.array([len(ds)], dtype=np.int64)
    out["mix.cls"] = np.array([int(ds[i][3]) for i in range(len(ds))], dtype=np.int64)
    path = os.path.join(ROOT, "tests", "golden", "g12_data.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


if __name__ == "__main__":
    main()
