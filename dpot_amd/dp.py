"""Data parallelism for the DPOT step: one process per GPU, RCCL all-reduce of the flat gradient buffer over xGMI,
overlapped with backward.  Replaces what the reference gets from HF-accelerate -> torch DDP -> NCCL
(train_temporal_parallel.py:102,185,243-244).

Semantics kept from the reference:
  * every rank holds a full replica; parameters are broadcast from rank 0 once at start-up;
  * the loss is a per-rank batch SUM and DDP AVERAGES gradients over ranks -> here: all-reduce(SUM) of the flat
    buffer and a 1/world_size ``grad_scale`` folded into the fused clip+Adam kernel (no extra pass over memory);
  * clip + Adam run redundantly on every rank after the reduction (replicated optimiser, no ZeRO);
  * per-rank batch = configured batch size (Accelerator(split_batches=False)): ``shard_indices`` hands rank r the
    batches r, r+W, ... of the shuffled batch list; the last round wraps around to the start of the epoch
    (accelerate's even_batches=True with drop_last=False), nothing is dropped.

MI355X-specific choices: xGMI is point-to-point (7 links x ~153 GB/s per GPU), ring collectives are per-link
bound and latency matters for the 30 MB Tiny gradient, so the buffer is cut into FEW LARGE contiguous buckets
(default 4) in reverse execution order; each bucket is reduced on a side HIP stream as soon as autograd has
produced its last gradient, and the compute stream only waits for the side stream right before the optimiser.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .train import FlatParams


def auto_n_buckets(grad_bytes: int) -> int:
    """bucket count by gradient BYTES (round 5; a fixed 4 before): ~32 MB per bucket, at least 2 (one bucket would leave
    nothing to overlap), at most 8 (every bucket boundary is a graph cut: ~40 us of host + launch gap per cut,
    profiles/r05_dp_host_time.txt, and ring all-reduces over xGMI want large messages - 7 links x ~153 GB/s per GPU).
    DPOT-Tiny (30 MB) -> 2, DPOT-S (123 MB) -> 4, DPOT-M (489 MB) / DPOT-L (2 GB) -> 8; SURVEY 8e: 2-4 for Tiny.
    The reference leaves this to torch-DDP's 25 MB default (train_temporal_parallel.py:185 via accelerate)."""
    return max(2, min(8, -(-int(grad_bytes) // (32 << 20))))


class BucketedGradReducer:
    def __init__(self, flat: FlatParams, process_group=None, n_buckets: Optional[int] = None, overlap: bool = True):
        self.fp = flat
        if n_buckets is None:
            n_buckets = auto_n_buckets(flat.n_head * flat.grad.element_size())
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.grad_scale = 1.0 / self.world
        self.is_cuda = flat.grad.is_cuda
        self.overlap = overlap and self.is_cuda
        self.stream = torch.cuda.Stream() if self.overlap else None
        # contiguous buckets over the flat buffer (execution order), roughly equal bytes, cut only BETWEEN STAGES
        # (embed+time-agg | block 0 | ... | block N-1 | out layer): a bucket is then final exactly when the backward
        # of its first stage has run, which is where the segmented hipGraph step (train.SegmentedTrainStep) cuts
        n_params = len(flat.params)
        ends = [flat.offsets[i + 1] if i + 1 < n_params else flat.total for i in range(n_params)]
        self.stage_of: List[int] = [self._stage(n) for n in flat.names]       # -1 = cls_head tail
        target = flat.n_head / max(1, n_buckets)
        stage_bytes = {}
        for i in range(n_params):
            lo = flat.offsets[i]
            stage_bytes[self.stage_of[i]] = stage_bytes.get(self.stage_of[i], 0) + ends[i] - lo
        self.bucket_of: List[int] = [0] * n_params
        self.ranges: List[List[int]] = []
        start, b = 0, 0
        for i in range(n_params):
            self.bucket_of[i] = b
            last_of_stage = i == n_params - 1 or self.stage_of[i + 1] != self.stage_of[i]
            nxt = stage_bytes[self.stage_of[i + 1]] if (last_of_stage and i + 1 < n_params) else 0
            cur = ends[i] - start
            # close at a stage end once the bucket is big enough, or when swallowing the next stage would overshoot;
            # the cls_head tail (no gradient in single-loss training) always gets a bucket of its own
            if i == n_params - 1 or ends[i] == flat.n_head or \
                    (last_of_stage and (cur >= target or cur + nxt > 1.5 * target)):
                self.ranges.append([start, ends[i]])
                start = ends[i]
                b += 1
        self.n_buckets = len(self.ranges)
        self.tail_bucket = self.bucket_of[-1] if flat.n_head < flat.total else -1
        self.members = [[i for i in range(n_params) if self.bucket_of[i] == k] for k in range(self.n_buckets)]
        self._pending = [0] * self.n_buckets
        self._seen: List[bool] = [False] * n_params
        self._work = []
        self._launched = [False] * self.n_buckets
        self._hooks = []
        # two notification sources: (a) the HIP stage backwards write gradients straight into the flat buffer and
        # call FlatParams.fire(i); (b) ordinary autograd accumulation (foreign graphs) -> post-accumulate hooks
        self._attached = True
        self.dry_run = False                 # world 1 only: go through the motions of the bucket all-reduces (see _launch)
        # world 1 only: issue the collectives for real on a one-rank process group (identity; RCCL enqueues no device work for
        # one rank) - exercises the communicator, ProcessGroupNCCL's host path and the stream hand-offs on a 1-GPU box
        # (tests/test_gpu_train2.py, bench.py DPOT_BENCH_FORCE_DP=1)
        self.single_rank_collective = False
        self.skip_zero_tail = True
        flat.callbacks.append(self._on_ready)
        for i, p in enumerate(flat.params):
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    # -- hipGraph capture ----------------------------------------------------------------------------------
    def detach(self) -> None:
        """ignore gradient-ready notifications (while a backward is being CAPTURED: a collective must never be
        issued from inside a stream capture; the segmented step launches the buckets itself between replays)"""
        self._attached = False

    def attach(self) -> None:
        self._attached = True

    @staticmethod
    def _stage(name: str) -> int:
        """execution-order stage of a parameter: 0 = embed (patch_embed, pos_embed, time_agg_layer, scale_feats),
        1 + i = blocks.i, 10**6 = out_layer, -1 = cls_head"""
        if name.startswith("blocks."):
            return 1 + int(name.split(".")[1])
        if name.startswith("out_layer."):
            return 10 ** 6
        if name.startswith("cls_head."):
            return -1
        return 0

    def first_stage_of_bucket(self, k: int) -> int:
        return self.stage_of[self.members[k][0]]

    # -- set-up ------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src: int = 0) -> None:
        """rank 0 -> all, one collective over the flat fp32 parameter buffer (DDP's initial broadcast)."""
        if self.world > 1 or (self.single_rank_collective and dist.is_initialized()):
            dist.broadcast(self.fp.flat, src=src, group=self.pg)
            self.fp.epoch += 1           # parameters replaced: derived / packed weight copies (PanelPacks.fresh_for) are stale

    # -- per step ----------------------------------------------------------------------------------------
    def begin_step(self) -> None:
        self._pending = [len(m) for m in self.members]
        self._seen = [False] * len(self.fp.params)
        self._launched = [False] * self.n_buckets
        self._work = []

    def _on_ready(self, i: int) -> None:
        if not self._attached or self._seen[i]:
            return
        self._seen[i] = True
        k = self.bucket_of[i]
        self._pending[k] -= 1
        if self._pending[k] == 0:
            self._launch(k)

    def _make_hook(self, i: int):
        def hook(_param):
            self._on_ready(i)
        return hook

    def _launch(self, k: int) -> None:
        single = self.world == 1 and not (self.single_rank_collective and dist.is_initialized())
        if self._launched[k] or (single and not self.dry_run):
            self._launched[k] = True
            return
        if single:
            # dry run on one process (scripts/dp_host_time.py): the stream choreography of a bucket all-reduce - side
            # stream waits for the compute stream, one device operation over the bucket on the side stream - without a
            # collective, to time the HOST side of the segmented step where no second GPU exists
            self._launched[k] = True
            lo, hi = self.ranges[k]
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.fp.grad[lo:hi].mul_(1.0)
            return
        self._launched[k] = True
        lo, hi = self.ranges[k]
        buf = self.fp.grad[lo:hi]
        if self.overlap:
            self.stream.wait_stream(torch.cuda.current_stream())     # gradients of this bucket are complete
            with torch.cuda.stream(self.stream):
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
        else:
            self._work.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def reduce_bucket(self, k: int) -> None:
        """explicitly launch the all-reduce of bucket k (segmented hipGraph step: called after the graph segment
        that finalises the bucket's gradients has been enqueued)"""
        self._launch(k)

    def finish(self) -> None:
        """reduce whatever has not been reduced yet (parameters that received no gradient this step still take
        part: their slice is zero, as with DDP + `0.0 * cls_loss`), then make the compute stream wait."""
        for k in range(self.n_buckets):
            if not self._launched[k]:
                if k == self.tail_bucket and self.skip_zero_tail and not any(
                        self._seen[i] or self.fp.filled[i] for i in self.members[k]):
                    # cls_head received no gradient on this rank - and (same program on every rank) on no rank:
                    # its slice is zero everywhere, SUM of zeros is zero: no collective needed (DDP + `0.0*cls_loss`
                    # all-reduces those zeros, train_temporal_parallel.py:243; the result is identical)
                    self._launched[k] = True
                    continue
                self._launch(k)
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self.stream)
        for w in self._work:
            w.wait()
        self._work = []

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


def dp_lr_step_index(opt_step: int, world: int) -> int:
    """accelerate's AcceleratedScheduler (split_batches=False) steps the wrapped LR scheduler `world` times per
    optimiser step, and train_temporal_parallel.py:150 sizes OneCycleLR with the UNSHARDED loader length
    (epochs * len(train_loader) before accelerator.prepare) - so the schedule still spans the run.  Index into that
    schedule for the lr used by optimiser step `opt_step` (0-based): scheduler.step() has been called opt_step*world
    times when the step's optimizer.step() runs."""
    return opt_step * world


def dp_one_cycle_lr(opt_step: int, world: int, total_steps_unsharded: int, max_lr: float, **kw) -> float:
    from .train import one_cycle_lr
    return one_cycle_lr(min(dp_lr_step_index(opt_step, world), total_steps_unsharded - 1), total_steps_unsharded,
                        max_lr, **kw)


def gather_eval_metric(value: torch.Tensor, process_group=None) -> torch.Tensor:
    """train_temporal_parallel.py:294-295 `accelerator.gather_for_metrics((loss,))`: every rank's scalar, in rank
    order, on every rank ([world] tensor; the reference then sums it)."""
    value = value.detach().reshape(1)
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        out = [torch.empty_like(value) for _ in range(dist.get_world_size(process_group))]
        dist.all_gather(out, value, group=process_group)
        return torch.cat(out)
    return value.clone()


def shard_batches(order: Sequence[int], batch_size: int, rank: int, world: int) -> List[List[int]]:
    """The batches of rank `rank` for one epoch over the common sample order `order`, with the semantics the reference
    gets from `accelerator.prepare(train_loader)` (train_temporal_parallel.py:120,185: `DataLoader(batch_size,
    shuffle=True)`, i.e. drop_last=False, under `Accelerator(split_batches=False)` (:102) whose default is
    even_batches=True -> accelerate's `BatchSamplerShard`): batch i of the unsharded loader goes to rank i % W, and the
    LAST round is completed by WRAPPING AROUND to the samples at the start of the epoch - first the ragged final batch
    is filled up to `batch_size`, then whole batches are appended until every rank has the same number of full
    batches.  Nothing is dropped; a few early samples are seen twice.  Every rank runs ceil(ceil(n / bs) / W) steps.
    (tests/test_cpu_host.py compares with accelerate's class itself.)"""
    n = len(order)
    if n == 0:
        return []
    per_round = batch_size * world
    n_batches = -(-n // batch_size)
    rounds = -(-n_batches // world)
    head = list(order[:per_round])                    # what the wrap-around draws from: the first W batches
    while len(head) < per_round:                      # fewer than one round of samples: cycle them
        head += head
    ext = list(order) + head[:rounds * per_round - n]
    return [ext[(k * world + rank) * batch_size:(k * world + rank + 1) * batch_size] for k in range(rounds)]


def shard_indices(n_samples: int, batch_size: int, rank: int, world: int, epoch: int, seed: int = 0,
                  shuffle: bool = True) -> List[List[int]]:
    """all ranks draw the same shuffled order (common seed + epoch, like accelerate's synchronised
    SeedableRandomSampler) and take their `shard_batches` of it."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    order = torch.randperm(n_samples, generator=g).tolist() if shuffle else list(range(n_samples))
    return shard_batches(order, batch_size, rank, world)


def dp_steps_per_epoch(n_samples: int, batch_size: int, world: int) -> Tuple[int, int]:
    """-> (optimiser steps per epoch on every rank, length of the UNSHARDED loader).  The second number is what the
    reference sizes OneCycleLR with (train_temporal_parallel.py:150: `epochs * len(train_loader)` before
    `accelerator.prepare`), the first is how many steps really run (see `dp_lr_step_index`)."""
    unsharded = -(-n_samples // batch_size)
    return -(-unsharded // world), unsharded


def all_reduce_scalar(value: torch.Tensor, process_group=None) -> torch.Tensor:
    """sum of a per-rank scalar (eval metrics; train_temporal_parallel.py:294-297 gathers them)."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        value = value.clone()
        dist.all_reduce(value, op=dist.ReduceOp.SUM, group=process_group)
    return value
