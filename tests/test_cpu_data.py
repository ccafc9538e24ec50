"""CPU tests of the input pipeline's host logic and of its oracle against the golden vectors (g12: written by the
reference's own `MixedTemporalDataset.__getitem__`, utils/griddataset.py:125-175, run in the build container with an
in-memory stand-in for h5py - oracle/make_golden_data.py)."""
import numpy as np
import torch

from helpers import load
from oracle import data_ref as D


def test_data_oracle_matches_golden():
    fx = load("g12_data")
    k = 0
    while f"c{k}.meta" in fx.files:
        H, W, T, Cc, res, nc, t_in, t_ar, t0 = (int(v) for v in fx[f"c{k}.meta"])
        raw = D.recipe_sample((H, W, T, Cc), salt=100 + k)
        x, y = D.window(D.pad_data(raw, res, nc), t0, t_in, t_ar)
        assert np.array_equal(x.numpy(), fx[f"c{k}.x"]) and np.array_equal(y.numpy(), fx[f"c{k}.y"])
        assert (x[..., Cc:] == 1).all()                       # padded channels are ones
        k += 1
    assert k == 4


def test_mixed_index_and_window_rules():
    from dpot_amd.data import MixedIndex, random_window_start, target_mask
    mi = MixedIndex([5, 3, 4], [1, 2, 1])                     # dataset 1 is repeated twice (data_weights)
    assert len(mi) == 5 + 6 + 4
    # griddataset.py:133-141
    cum = np.cumsum([5, 6, 4])
    for idx in range(len(mi)):
        d = int(np.searchsorted(cum, idx + 1))
        local = idx if d == 0 else idx - cum[d - 1]
        assert mi.locate(idx) == (d, int(local // [1, 2, 1][d]))
    assert mi.locate(5) == (1, 0) and mi.locate(6) == (1, 0) and mi.locate(7) == (1, 1) and mi.locate(11) == (2, 0)
    rng = np.random.default_rng(0)
    assert all(0 <= random_window_start(20, 10, 1, rng) <= 9 for _ in range(200))
    assert {random_window_start(11, 10, 1, rng) for _ in range(20)} == {0}
    assert random_window_start(5, 10, 1, rng) == 0            # short trajectory: max(.., 1) -> start 0
    # griddataset.py:103-117 transcribed
    res, size_orig, nc = 16, [4, 8, 20, 2], 3
    msk = torch.zeros(res, res, 1, nc)
    kx, ky = res // size_orig[0], res // size_orig[1]
    msk[::kx, ::ky, :, :size_orig[-1]] = 1
    assert torch.equal(target_mask(res, size_orig, nc), msk)
    assert target_mask(8, [16, 16, 5, 1], 2)[..., 0].sum() == 64          # target coarser than the data: k -> 1


def test_data_oracle_test_mode_and_downsample_match_golden():
    """g12 t0..t3: the reference's `__getitem__` with train=False (test window, get_target_mask, downsample;
    griddataset.py:143-174) against the oracle's restatement and the product's host helpers (eval_window, target_mask)"""
    from dpot_amd.data import eval_window, target_mask
    fx = load("g12_data")
    k = 0
    while f"t{k}.meta" in fx.files:
        H, W, T, Cc, res, nc, t_in, t_test, dh, dw, pc = (int(v) for v in fx[f"t{k}.meta"])
        raw = D.recipe_sample((H, W, T, Cc), salt=200 + k)
        padded = D.pad_data(raw, res, nc)
        x, y = D.downsample(*D.test_window(padded, t_in, t_test), (dh, dw))
        size_orig = [H, W, T, Cc if pc < 0 else pc]
        msk = D.target_mask(padded, size_orig)
        assert np.array_equal(x.numpy(), fx[f"t{k}.x"]) and np.array_equal(y.numpy(), fx[f"t{k}.y"])
        assert np.array_equal(msk.numpy(), fx[f"t{k}.msk"])
        t0, t_ar = eval_window(T, t_in, t_test)
        assert t0 == 0 and t_ar == fx[f"t{k}.y"].shape[2]
        assert np.array_equal(target_mask(res, size_orig, nc).numpy(), fx[f"t{k}.msk"])
        k += 1
    assert k == 4


def test_mixed_index_matches_reference_dataset_mixing():
    """g12 mix.*: `idx_cls` of every global index of a two-dataset mix with data_weights (1, 2), as the reference's
    `__getitem__` returned it (griddataset.py:133-141,174)"""
    from dpot_amd.data import MixedIndex
    fx = load("g12_data")
    mi = MixedIndex([3, 3], [1, 2])
    assert len(mi) == int(fx["mix.len"][0])
    assert [mi.locate(i)[0] for i in range(len(mi))] == fx["mix.cls"].tolist()
