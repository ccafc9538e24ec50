"""CPU oracle of the input pipeline's sample transform.  TEST INFRASTRUCTURE ONLY.

Restates utils/griddataset.py:88-101 (`pad_data`: bilinear resize of every (t, c) image with
F.interpolate(mode='bilinear'), channel pad with ones) and :150-153 (training window) for one raw sample.
Pinned by tests/golden/g12_data.npz, written by the reference's own `MixedTemporalDataset.__getitem__` imported from
/root/reference (oracle/make_golden_data.py: in-memory stand-in for the absent h5py, temporary CWD).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def pad_data(x: Tensor, res: int, n_channels: int) -> Tensor:
    """[H,W,T,C] -> [res,res,T,n_channels]   (griddataset.py:88-101)"""
    H, W, T, Cc = x.shape
    img = x.reshape(H, W, T * Cc).permute(2, 0, 1).unsqueeze(0)                # [1, T*C, H, W]
    img = F.interpolate(img, size=(res, res), mode="bilinear").squeeze(0).permute(1, 2, 0)
    img = img.reshape(res, res, T, Cc)
    out = torch.ones(res, res, T, n_channels)
    out[..., :Cc] = img
    return out


def window(sample: Tensor, t0: int, t_in: int, t_ar: int):
    """griddataset.py:152: x = sample[..., t0:t0+t_in, :], y = sample[..., t0+t_in : t0+t_in+t_ar, :]"""
    return sample[..., t0:t0 + t_in, :], sample[..., t0 + t_in:min(t0 + t_in + t_ar, sample.shape[-2]), :]


def test_window(sample: Tensor, t_in: int, t_test: int):
    """griddataset.py:159-161 (test datasets): x = the first t_in frames, y = sample[..., t_in : t_in + t_test, :]"""
    return sample[..., 0:t_in, :], sample[..., t_in:t_in + t_test, :]


def target_mask(sample: Tensor, size_orig) -> Tensor:
    """griddataset.py:103-117 get_target_mask: ones on the grid points / channels the dataset itself has"""
    msk = torch.zeros(*sample.shape[:2], 1, sample.shape[-1])
    kx, ky = sample.shape[0] // size_orig[0], sample.shape[1] // size_orig[1]
    kx, ky = (1 if kx == 0 else kx), (1 if ky == 0 else ky)
    msk[::kx, ::ky, :, :size_orig[-1]] = 1
    return msk


def downsample(x: Tensor, y: Tensor, d):
    """griddataset.py:170-172"""
    return x[::d[0], ::d[1]], y[::d[0], ::d[1]]


def recipe_sample(shape, salt: int) -> Tensor:
    """closed-form raw trajectory (smooth field + a hash texture), identical wherever it is evaluated"""
    from oracle.dpot_ref import recipe_tensor
    H, W, T, Cc = shape
    u = recipe_tensor("sample", shape, salt).float()
    gx = torch.linspace(0, 1, H).view(H, 1, 1, 1)
    gy = torch.linspace(0, 1, W).view(1, W, 1, 1)
    gt = torch.linspace(0, 1, T).view(1, 1, T, 1)
    c = torch.arange(1, Cc + 1).view(1, 1, 1, Cc).float()
    return torch.sin(6.0 * gx * c + 3.0 * gt) * torch.cos(4.0 * gy + c) + 0.25 * (u - 0.5)
