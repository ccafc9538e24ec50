"""GPU tests of the device-side input pipeline (csrc/data.hip, dpot_amd/data.py): bilinear resize + channel pad +
temporal window against the golden vectors / the CPU oracle, and the double-buffered DeviceBatcher end to end."""
import numpy as np
import pytest
import torch

from helpers import assert_close, load
from oracle import data_ref as D

pytestmark = pytest.mark.gpu


def test_resize_pad_window_golden_and_batched():
    from dpot_amd.data import resize_pad_window
    fx = load("g12_data")
    for k in range(4):
        H, W, T, Cc, res, nc, t_in, t_ar, t0 = (int(v) for v in fx[f"c{k}.meta"])
        raw = D.recipe_sample((H, W, T, Cc), salt=100 + k).contiguous().cuda()
        xx, yy = resize_pad_window([raw], [t0], res, t_in, t_ar, nc)
        assert_close(xx[0], fx[f"c{k}.x"], f"case {k} x", rtol=1e-6, atol_scale=1e-6)
        assert_close(yy[0], fx[f"c{k}.y"], f"case {k} y", rtol=1e-6, atol_scale=1e-6)
    # one launch, samples of DIFFERENT datasets (shapes) in one batch, at the BASELINE resolution
    shapes = [(64, 64, 20, 1), (128, 128, 14, 3), (256, 256, 12, 4), (100, 60, 16, 2)] * 20      # 80 samples: 2 launches
    starts = [min(i % 3, s[2] - 11) for i, s in enumerate(shapes)]
    raws = [D.recipe_sample(s, salt=7 + i) for i, s in enumerate(shapes)]
    xx, yy = resize_pad_window([r.cuda() for r in raws], starts, 128, 10, 1, 4)
    for i in (0, 1, 2, 3, 64, 79):
        xr, yr = D.window(D.pad_data(raws[i], 128, 4), starts[i], 10, 1)
        assert_close(xx[i], xr, f"sample {i} x", rtol=1e-6, atol_scale=1e-6)
        assert_close(yy[i], yr, f"sample {i} y", rtol=1e-6, atol_scale=1e-6)
    with pytest.raises(Exception):
        resize_pad_window([raws[0].cuda()], [15], 128, 10, 1, 4)          # window runs past the trajectory


def test_device_batcher_double_buffer_overlaps_and_is_exact():
    from dpot_amd.data import DeviceBatcher
    B, res, t_in, t_ar, nc = 4, 64, 6, 2, 3
    shapes = [(32, 32, 12, 1), (64, 64, 10, 3), (48, 40, 9, 2), (32, 32, 12)]
    db = DeviceBatcher(B, res, t_in, t_ar, nc, max_raw_floats_per_sample=64 * 64 * 10 * 3)
    rng = np.random.default_rng(1)
    batches = []
    for it in range(5):                                   # more batches than slots: slots are recycled
        raws = [D.recipe_sample(s if len(s) == 4 else s + (1,), salt=20 * it + i) for i, s in enumerate(shapes)]
        raws = [r if len(s) == 4 else r[..., 0] for r, s in zip(raws, shapes)]           # a [H,W,T] dataset
        starts = [int(rng.integers(0, 2)) for _ in shapes]
        batches.append((raws, starts))
    db.submit(batches[0][0], batches[0][1])
    for it in range(5):
        if it + 1 < 5:
            db.submit(batches[it + 1][0], batches[it + 1][1])             # next batch in flight while this one is read
        xx, yy, msk = db.get()
        got_x, got_y = xx.clone(), yy.clone()                             # "the step": reads the slot on this stream
        db.release()
        raws, starts = batches[it]
        for i, r in enumerate(raws):
            r4 = r if r.dim() == 4 else r.unsqueeze(-1)
            xr, yr = D.window(D.pad_data(r4, res, nc), starts[i], t_in, t_ar)
            assert_close(got_x[i], xr, f"batch {it} sample {i} x", rtol=1e-6, atol_scale=1e-6)
            assert_close(got_y[i], yr, f"batch {it} sample {i} y", rtol=1e-6, atol_scale=1e-6)
        assert tuple(msk.shape) == (B, res, res, 1, nc) and bool((msk == 1).all())
    assert db.h2d_bytes == sum(int(np.prod(r.shape)) * 4 for raws, _ in batches for r in raws)


def test_test_mode_windows_downsample_and_idx_cls_golden():
    """SURVEY f2 remainder (griddataset.py:159-174): test-mode window (x = first t_in frames, y = the following t_test
    frames, clipped at the trajectory's end), strided down-sampling of the resized fields, idx_cls - the device kernel
    against the golden vectors written by the reference dataset class itself (g12 t0..t3), and through DeviceBatcher"""
    from dpot_amd.data import DeviceBatcher, eval_window, resize_pad_window
    fx = load("g12_data")
    for k in range(4):
        H, W, T, Cc, res, nc, t_in, t_test, dh, dw, pc = (int(v) for v in fx[f"t{k}.meta"])
        raw = D.recipe_sample((H, W, T, Cc), salt=200 + k).contiguous()
        t0, t_ar = eval_window(T, t_in, t_test)
        xx, yy = resize_pad_window([raw.cuda()], [t0], res, t_in, t_ar, nc, downsample=(dh, dw))
        assert_close(xx[0], fx[f"t{k}.x"], f"test case {k} x", rtol=1e-6, atol_scale=1e-6)
        assert_close(yy[0], fx[f"t{k}.y"], f"test case {k} y", rtol=1e-6, atol_scale=1e-6)
        # the same through the double-buffered batcher (batch of 2 identical samples, dataset ids 3 and 5)
        db = DeviceBatcher(2, res, t_in, t_ar, nc, max_raw_floats_per_sample=raw.numel(), downsample=(dh, dw))
        db.submit([raw.numpy(), raw.numpy()], [t0, t0], dataset_ids=[3, 5])
        bx, by, _ = db.get()
        cls = db.last_cls.clone()
        gx, gy = bx.clone(), by.clone()
        db.release()
        torch.cuda.synchronize()
        assert cls.dtype == torch.int64 and cls.cpu().tolist() == [[3], [5]]
        assert_close(gx[1], fx[f"t{k}.x"], f"batcher test case {k} x", rtol=1e-6, atol_scale=1e-6)
        assert_close(gy[0], fx[f"t{k}.y"], f"batcher test case {k} y", rtol=1e-6, atol_scale=1e-6)


def test_resize_pad_window_rejects_wrongly_shaped_output_buffers():
    """caller-supplied out_xx / out_yy must have the down-sampled shape the kernel writes"""
    from dpot_amd import _lib
    from dpot_amd.data import resize_pad_window
    raw = [torch.rand(16, 16, 8, 1, device="cuda")]
    ok_x = torch.empty(1, 16, 16, 4, 4, device="cuda")
    with pytest.raises(_lib.DpotHipError, match="out_xx"):
        resize_pad_window(raw, [0], 32, 4, 2, 4, out_xx=torch.empty(1, 32, 32, 4, 4, device="cuda"), downsample=(2, 2))
    with pytest.raises(_lib.DpotHipError, match="out_yy"):
        resize_pad_window(raw, [0], 32, 4, 2, 4, out_xx=ok_x, out_yy=torch.empty(1, 16, 16, 1, 4, device="cuda"),
                          downsample=(2, 2))
    with pytest.raises(_lib.DpotHipError, match="out_xx"):
        resize_pad_window(raw, [0], 32, 4, 2, 4, out_xx=torch.empty(1, 16, 16, 4, 8, device="cuda")[..., ::2],
                          downsample=(2, 2))
    xx, yy = resize_pad_window(raw, [0], 32, 4, 2, 4, out_xx=ok_x, downsample=(2, 2))
    assert xx is ok_x and tuple(yy.shape) == (1, 16, 16, 2, 4)
