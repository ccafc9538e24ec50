"""Input pipeline for the DPOT step, MI355X side (reference: utils/griddataset.py:27-174 and the
DataLoader of train_temporal.py:106-109).

The reference resizes / pads / windows every sample on CPU workers (F.interpolate per sample) and ships finished
[B,res,res,T,C] batches (2.9 MB per sample at 128^2) through a DataLoader - at the GPU step's ~9 k samples/s that
pipeline is the bottleneck.  Here the CPU side only hands over the RAW trajectories ([H,W,T,C] as stored, e.g. 64x64
for ns2d_fno: 16x fewer bytes than the padded 128x128x4 batch):

    host:    raw samples -> one pinned staging buffer -> ONE async H2D copy on a copy stream
    device:  dpot_resize_pad_window (csrc/data.hip): bilinear resize to res x res, channel pad with ones, temporal
             window [t0, t0+t_in+t_ar) -> xx, yy written straight into one of two (n_buffers) batch slots
    step:    the compute stream waits on the slot's event; while it trains on slot k the copy stream fills slot k+1

HDF5 reading (h5py is not available in this image) stays with the caller: `DeviceBatcher.submit` takes numpy arrays /
CPU tensors.  Dataset mixing and index sharding are host logic (`MixedIndex`, dp.shard_indices).

Also covered (griddataset.py:159-174): test-mode windows (`train=False`: x = the first t_in frames, y = the following
t_test frames, the resolution / channel mask of `get_target_mask`), the per-dataset strided `downsample` of the resized
fields, and `idx_cls` (the dataset index of every sample, int64 [B, 1]).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import SampleDesc, check

Tensor = torch.Tensor


def random_window_start(T: int, t_in: int, t_ar: int, rng: np.random.Generator) -> int:
    """griddataset.py:151  start_idx = np.random.randint(max(T - (t_in + t_ar) + 1, 1))"""
    return int(rng.integers(max(T - (t_in + t_ar) + 1, 1)))


def eval_window(T: int, t_in: int, t_test: int) -> Tuple[int, int]:
    """griddataset.py:159-161 (test datasets return the whole trajectory): (t0, t_ar) = (0, number of frames of
    sample[..., t_in : t_in + t_test, :] that exist)"""
    return 0, max(0, min(t_test, T - t_in))


def target_mask(res: int, size_orig: Sequence[int], n_channels: int) -> Tensor:
    """griddataset.py:103-117: evaluation mask [res, res, 1, n_channels] - ones on the grid points that exist at the
    dataset's own resolution and on its own channels (size_orig = [H, W, T, C_pred])"""
    msk = torch.zeros(res, res, 1, n_channels)
    kx, ky = res // size_orig[0], res // size_orig[1]
    kx, ky = max(kx, 1), max(ky, 1)
    msk[::kx, ::ky, :, :size_orig[-1]] = 1
    return msk


class MixedIndex:
    """index arithmetic of MixedTemporalDataset (griddataset.py:50-56,133-141): datasets are concatenated, dataset d
    is repeated data_weights[d] times; global index -> (dataset, sample)"""

    def __init__(self, sizes: Sequence[int], weights: Optional[Sequence[int]] = None):
        self.sizes = list(sizes)
        self.weights = list(weights) if weights is not None else [1] * len(self.sizes)
        self.cumulative = np.cumsum([s * w for s, w in zip(self.sizes, self.weights)])

    def __len__(self) -> int:
        return int(self.cumulative[-1])

    def locate(self, idx: int) -> Tuple[int, int]:
        d = int(np.searchsorted(self.cumulative, idx + 1))
        local = idx if d == 0 else idx - int(self.cumulative[d - 1])
        return d, int(local // self.weights[d])


DESC_BYTES = C.sizeof(SampleDesc)          # 32
DESC_FLOATS = DESC_BYTES // 4


def _fill_table(table: np.ndarray, ptrs: Sequence[int], shapes: Sequence[Sequence[int]], starts: Sequence[int],
                t_in: int, t_ar: int, n_channels: int) -> None:
    """write the dpot_sample_desc records into `table` (uint8 view of host memory), validating what the kernel assumes"""
    descs = (SampleDesc * len(ptrs)).from_buffer(table)
    for i, (ptr, (H, W, T, Cc), t0) in enumerate(zip(ptrs, shapes, starts)):
        if Cc > n_channels or t0 < 0 or t0 + t_in + t_ar > T:
            raise ValueError(f"sample {i}: shape {(H, W, T, Cc)}, window [{t0}, {t0 + t_in + t_ar}) - needs C <= "
                             f"{n_channels} and the window inside its {T} frames")
        descs[i].data, descs[i].H, descs[i].W, descs[i].T, descs[i].C, descs[i].t0 = ptr, H, W, T, Cc, int(t0)


def _down_shape(res: int, downsample: Tuple[int, int]) -> Tuple[int, int]:
    dh, dw = int(downsample[0]), int(downsample[1])
    if dh < 1 or dw < 1:
        raise ValueError(f"downsample {downsample}: factors must be >= 1")
    return (res + dh - 1) // dh, (res + dw - 1) // dw


def resize_pad_window(samples: Sequence[Tensor], starts: Sequence[int], res: int, t_in: int, t_ar: int,
                      n_channels: int, out_xx: Optional[Tensor] = None, out_yy: Optional[Tensor] = None,
                      downsample: Tuple[int, int] = (1, 1)):
    """samples: CUDA tensors [H,W,T,C] (fp32, contiguous), one per batch entry -> (xx [B,res,res,t_in,Cmax],
    yy [B,res,res,t_ar,Cmax]) through ONE launch of csrc/data.hip.  downsample = (dh, dw): the reference's
    x[::dh, ::dw] on the resized fields (griddataset.py:170-172) - the outputs then have ceil(res/d) points per axis"""
    B = len(samples)
    dev = samples[0].device
    rh, rw = _down_shape(res, downsample)
    xx = out_xx if out_xx is not None else torch.empty(B, rh, rw, t_in, n_channels, dtype=torch.float32, device=dev)
    yy = out_yy if out_yy is not None else (
        torch.empty(B, rh, rw, t_ar, n_channels, dtype=torch.float32, device=dev) if t_ar > 0 else None)
    for name, buf, t in (("out_xx", out_xx, t_in), ("out_yy", out_yy, t_ar)):
        # the kernel writes a dense [B, ceil(res/dh), ceil(res/dw), t, C] block: any other caller-supplied buffer would
        # be garbled or overrun
        if buf is not None and not (tuple(buf.shape) == (B, rh, rw, t, n_channels) and buf.is_contiguous()
                                    and buf.dtype == torch.float32 and buf.device == dev):
            raise _lib.DpotHipError(f"resize_pad_window: {name} must be a contiguous float32 tensor of shape "
                                    f"{(B, rh, rw, t, n_channels)} on {dev}, got {tuple(buf.shape)} {buf.dtype} "
                                    f"{buf.device}")
    for s in samples:
        if not (s.is_cuda and s.dtype == torch.float32 and s.is_contiguous() and s.dim() == 4):
            raise _lib.DpotHipError("resize_pad_window: samples must be contiguous float32 CUDA tensors [H,W,T,C]")
    host = np.zeros(B * DESC_BYTES, dtype=np.uint8)
    _fill_table(host, [s.data_ptr() for s in samples], [tuple(s.shape) for s in samples], starts, t_in, t_ar, n_channels)
    table = torch.from_numpy(host).to(dev)
    check(_lib.load().dpot_resize_pad_window(table.data_ptr(), B, xx.data_ptr(), yy.data_ptr() if yy is not None else None,
                                             res, t_in, t_ar, n_channels, int(downsample[0]), int(downsample[1]),
                                             torch.cuda.current_stream().cuda_stream),
          "resize_pad_window")
    return xx, yy


class DeviceBatcher:
    """Double-buffered batch producer: raw samples in, device-resident (xx, yy, msk) out, overlapped with the step."""

    def __init__(self, batch: int, res: int, t_in: int, t_ar: int, n_channels: int, max_raw_floats_per_sample: int,
                 device="cuda", n_buffers: int = 2, downsample: Tuple[int, int] = (1, 1)):
        """t_ar: frames of y per sample (training: the rollout length; test mode: t_test of the dataset, with
        starts = 0 - see `eval_window`).  downsample: one factor pair per batcher (a batch is one tensor)."""
        self.B, self.res, self.t_in, self.t_ar, self.C = batch, res, t_in, t_ar, n_channels
        self.down = (int(downsample[0]), int(downsample[1]))
        rh, rw = _down_shape(res, self.down)
        self.dev = torch.device(device)
        self.n = n_buffers
        self.head = batch * DESC_FLOATS                # the descriptor table rides at the head of the staging buffer
        cap = self.head + batch * max_raw_floats_per_sample
        self.stream = torch.cuda.Stream(device=self.dev)
        self.host = [torch.empty(cap, dtype=torch.float32).pin_memory() for _ in range(n_buffers)]
        self.raw = [torch.empty(cap, dtype=torch.float32, device=self.dev) for _ in range(n_buffers)]
        self.xx = [torch.empty(batch, rh, rw, t_in, n_channels, device=self.dev) for _ in range(n_buffers)]
        self.yy = [torch.empty(batch, rh, rw, t_ar, n_channels, device=self.dev) for _ in range(n_buffers)]
        # training mask (griddataset.py:157: ones at the window's resolution, i.e. BEFORE the down-sampling - the reference
        # does not sub-sample msk; test-mode masks: `target_mask` per dataset)
        self.msk = torch.ones(batch, res, res, 1, n_channels, device=self.dev)
        self.cls = [torch.zeros(batch, 1, dtype=torch.int64, device=self.dev) for _ in range(n_buffers)]
        self.cls_host = [torch.zeros(batch, 1, dtype=torch.int64).pin_memory() for _ in range(n_buffers)]
        self.ready = [torch.cuda.Event() for _ in range(n_buffers)]       # slot filled (recorded on the copy stream)
        self.consumed = [None] * n_buffers                                # slot's last reader (recorded on compute)
        self.k = 0
        self.pending: List[int] = []
        self.used = [False] * n_buffers                                   # slot has been filled at least once
        self._out = set()                                                 # slots handed out by get(), not yet released
        self._last: Optional[int] = None
        self.h2d_bytes = 0

    def submit(self, samples: Sequence, starts: Sequence[int], dataset_ids: Optional[Sequence[int]] = None) -> None:
        """enqueue one batch: samples = numpy arrays / CPU tensors [H,W,T,C] (or [H,W,T]); returns immediately.
        dataset_ids: the dataset index of every sample (griddataset.py:174 idx_cls; default zeros)"""
        assert len(samples) == self.B and len(starts) == self.B
        assert dataset_ids is None or len(dataset_ids) == self.B
        slot = self.k % self.n
        # validate BEFORE anything is overwritten (a bad batch must leave the slot and the counters untouched)
        arrs, offs, shapes, off = [], [], [], self.head
        for i, (s, t0) in enumerate(zip(samples, starts)):
            a = np.asarray(s, dtype=np.float32) if not torch.is_tensor(s) else s.detach().float().numpy()
            if a.ndim == 3:
                a = a[..., None]                       # griddataset.py:145 "augment channel dim"
            H, W, T, Cc = a.shape
            if Cc > self.C or t0 < 0 or t0 + self.t_in + self.t_ar > T:
                raise ValueError(f"sample {i}: shape {(H, W, T, Cc)}, window [{t0}, {t0 + self.t_in + self.t_ar}) - needs "
                                 f"C <= {self.C} and the window inside its {T} frames")
            if off + a.size > self.host[slot].numel():
                raise ValueError("DeviceBatcher: raw samples exceed max_raw_floats_per_sample")
            arrs.append(a)
            offs.append(off)
            shapes.append(a.shape)
            off += a.size
        # the slot is reused: its pinned staging buffer, raw buffer and xx / yy must be free.  The previous H2D copy +
        # transform of this slot (copy stream) and the step that read it (recorded by release()) are waited for; a slot
        # that was handed out by get() but never release()d has an unknown reader - wait for the whole device then
        if slot in self.pending or slot in self._out:
            if slot in self.pending:
                raise RuntimeError(f"DeviceBatcher: all {self.n} slots hold batches that were never fetched with get()")
            torch.cuda.synchronize(self.dev)           # unreleased consumer: conservative, correct
            self._out.discard(slot)
        if self.used[slot]:
            self.ready[slot].synchronize()
        if self.consumed[slot] is not None:
            self.consumed[slot].synchronize()          # host buffer / device slot are free again (normally long since)
            self.consumed[slot] = None
        self.k += 1
        self.used[slot] = True
        host = self.host[slot]
        hnp = host.numpy()
        for a, o in zip(arrs, offs):
            hnp[o:o + a.size] = a.reshape(-1)
        base = self.raw[slot].data_ptr()
        _fill_table(hnp[:self.head].view(np.uint8), [base + 4 * o for o in offs], shapes, starts, self.t_in, self.t_ar,
                    self.C)
        self.cls_host[slot].zero_()
        if dataset_ids is not None:
            self.cls_host[slot][:, 0] = torch.as_tensor(list(dataset_ids), dtype=torch.int64)
        with torch.cuda.stream(self.stream):
            self.raw[slot][:off].copy_(host[:off], non_blocking=True)            # ONE H2D copy: table + raw samples
            self.cls[slot].copy_(self.cls_host[slot], non_blocking=True)
            check(_lib.load().dpot_resize_pad_window(base, self.B, self.xx[slot].data_ptr(), self.yy[slot].data_ptr(),
                                                     self.res, self.t_in, self.t_ar, self.C, self.down[0], self.down[1],
                                                     self.stream.cuda_stream),
                  "resize_pad_window")
            self.ready[slot].record(self.stream)
        self.h2d_bytes += (off - self.head) * 4
        self.pending.append(slot)

    def get(self) -> Tuple[Tensor, Tensor, Tensor]:
        """the oldest submitted batch; the CURRENT stream waits for it (no host synchronisation).  The tensors stay
        valid until `n_buffers` further submits; call `release()` (or `release(slot)` with `last_slot`) after the step that
        reads them was enqueued - a slot that is re-used without having been released costs a device synchronise."""
        slot = self.pending.pop(0)
        torch.cuda.current_stream().wait_event(self.ready[slot])
        self._last = slot
        self._out.add(slot)
        return self.xx[slot], self.yy[slot], self.msk

    @property
    def last_cls(self) -> Optional[Tensor]:
        """idx_cls [B, 1] int64 of the batch the latest get() returned (griddataset.py:174; train_temporal.py:199)"""
        return None if self._last is None else self.cls[self._last]

    @property
    def last_slot(self) -> Optional[int]:
        """token of the batch the latest get() returned (pass it to release() when several batches are in flight)"""
        return self._last

    def release(self, slot: Optional[int] = None) -> None:
        """the step reading batch `slot` (default: the latest get()) has been enqueued on the current stream"""
        slot = self._last if slot is None else slot
        if slot is None or slot not in self._out:
            raise RuntimeError("DeviceBatcher.release: no batch outstanding for this slot")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.consumed[slot] = ev
        self._out.discard(slot)
