"""A second HIP stream for the side branches of backward.

The backward of a block has a serial critical path (dgrad GEMMs, GroupNorm, DFTs) and a crowd of work that hangs off it
without feeding it: weight-gradient GEMMs, their split-K reductions, bias-gradient column sums, weight un-packing.
Those ~16 launches per block include many tiny kernels that cost ~5 us of wall time each while using a sliver of the
256 CUs.  Running the side work on a second stream (fork after the producer, join at the end of the stage) lets the
hardware overlap it with the critical path; inside a captured hipGraph this becomes a graph with parallel branches.

Memory safety: every tensor the side stream reads is kept alive by the stage's backward frame until the join, and the
main stream only continues past the join after the side work has finished - so the caching allocator never hands a
buffer that the side stream still uses to a later main-stream allocation.

Measured on MI355X (DPOT-Tiny, B=32, hipGraph replay): 3.94 ms/step with the side stream vs 3.84 ms without - the wgrad
GEMMs already fill all 256 CUs, so the branches contend instead of overlapping.  It is therefore OFF by default
(DPOT_SIDE_STREAM=1 turns it on); kept because it is the right structure once the critical path is shorter
(smaller batches / more GPUs per model).
"""
from __future__ import annotations

import os
from contextlib import contextmanager

import torch

_ENABLED = os.environ.get("DPOT_SIDE_STREAM", "0") == "1"
_streams = {}


def enabled() -> bool:
    return _ENABLED


def set_enabled(flag: bool) -> None:
    global _ENABLED
    _ENABLED = bool(flag)


def _side(device) -> "torch.cuda.Stream":
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _streams.get(idx)
    if s is None:
        s = torch.cuda.Stream(device=idx)
        _streams[idx] = s
    return s


@contextmanager
def side(device):
    """run the body on the side stream, ordered after everything enqueued on the current stream so far"""
    if not _ENABLED:
        yield
        return
    s = _side(device)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        yield


def join(device) -> None:
    """make the current stream wait for the side stream"""
    if _ENABLED and (device.index if device.index is not None else torch.cuda.current_device()) in _streams:
        torch.cuda.current_stream().wait_stream(_side(device))


# ------------------------------------------------------------------------------------------------------
# Prep stream: the weight-only products of a step (packed / padded / folded weights, ~20 tiny launches, ~100 us of wall
# time in a captured graph) do not depend on the batch.  DPOTNet.weights_scope() enqueues them on this stream while the
# main stream runs the batch-only head of the step (noise injection, patch gathering - HBM-bound kernels that leave the
# small launches room); the first consumer (EmbedFn) joins.
# Measured on MI355X (DPOT-Tiny, B=32, hipGraph replay): 3.75 ms/step with the prep branch vs 3.67 ms without - like the
# side stream above, a second branch in the captured graph costs more in cross-branch dependencies than the ~100 us of
# small launches it hides.  OFF by default (DPOT_PREP_STREAM=1 turns it on).
_PREP = os.environ.get("DPOT_PREP_STREAM", "0") == "1"
_prep_streams = {}
_prep_pending = set()


def _idx(device) -> int:
    return device.index if device.index is not None else torch.cuda.current_device()


@contextmanager
def prep(device):
    if not _PREP:
        yield
        return
    idx = _idx(device)
    s = _prep_streams.get(idx)
    if s is None:
        s = _prep_streams[idx] = torch.cuda.Stream(device=idx)
    s.wait_stream(torch.cuda.current_stream(idx))
    _prep_pending.add(idx)
    with torch.cuda.stream(s):
        yield


def prep_join(device) -> None:
    """the current stream waits for the prep stream if it has un-joined work (idempotent)"""
    idx = _idx(device)
    if idx in _prep_pending:
        _prep_pending.discard(idx)
        torch.cuda.current_stream(idx).wait_stream(_prep_streams[idx])
