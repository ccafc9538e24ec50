"""Rollout driver of the DPOT pre-training step (reference: train_temporal.py:189-230), MI355X style:

  * the auto-regressive rollout, the masked relative-L2 loss, backward, global-norm clip and Adam are all
    enqueued on one HIP stream with NO host synchronisation (the reference does three ``.item()`` per step);
  * parameters, gradients and Adam moments live in flat fp32 buffers: one fused kernel does clip + Adam for the
    whole model (the reference loops over 61 tensors in Python, utils/optimizer.py:26-52), and the flat gradient
    buffer is what the data-parallel reducer all-reduces in buckets (dp.py);
  * learning rate / bias corrections are read from device memory, so a fixed-shape step can be captured in a
    hipGraph and replayed (``GraphedTrainStep``).
"""
from __future__ import annotations

import contextlib
import math
import os
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops
from .functional import clear_grad_packs, rel_l2_loss

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------------
class FlatParams:
    """Re-homes the parameters of ``model`` into one flat fp32 buffer (and their .grad into another).

    ``order`` puts parameters whose names start with one of ``tail_prefixes`` (cls_head: it gets no gradient in
    single-GPU training, train_temporal.py:226) at the end, so the optimiser can address 'everything with a
    gradient' as one contiguous prefix."""

    # forward execution order of DPOTNet's sub-modules (registration order puts time_agg_layer after the blocks):
    # gradients then become final from the END of the buffer towards its start, so contiguous buckets can be
    # all-reduced while the rest of the backward still runs (dp.py)
    EXEC_ORDER = ("patch_embed.", "pos_embed", "time_agg_layer.", "scale_feats_", "blocks.", "out_layer.")

    @classmethod
    def _exec_rank(cls, name: str):
        for i, pre in enumerate(cls.EXEC_ORDER):
            if name.startswith(pre):
                return (i, int(name.split(".")[1])) if pre == "blocks." else (i, 0)
        return (len(cls.EXEC_ORDER), 0)

    def __init__(self, model: nn.Module, tail_prefixes: Sequence[str] = ("cls_head.",)):
        self.model = model
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        head = [(n, p) for n, p in named if not any(n.startswith(t) for t in tail_prefixes)]
        head.sort(key=lambda np_: self._exec_rank(np_[0]))                     # stable: registration order within a module
        tail = [(n, p) for n, p in named if any(n.startswith(t) for t in tail_prefixes)]
        self.names = [n for n, _ in head + tail]
        self.params = [p for _, p in head + tail]
        dev, dt = self.params[0].device, torch.float32
        # every tensor starts on a 4-float boundary so float4 kernels and bucket slices stay aligned
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.total = off
        self.n_head = self.offsets[len(head)] if tail else off
        self.flat = torch.zeros(self.total, dtype=dt, device=dev)
        self.grad = torch.zeros(self.total, dtype=dt, device=dev)
        self.grad_views: List[Tensor] = []
        n_p = len(self.params)
        self.filled: List[bool] = [False] * n_p      # slot already holds a gradient contribution this step
        self.pending: List[int] = [0] * n_p          # forward uses whose backward has not delivered yet
        self.callbacks: List[Callable[[int], None]] = []   # fired with the parameter index when its grad is final
        self.epoch = 0                               # bumped by the optimiser whenever it changes the parameters
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                self.flat[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                self.grad_views.append(self.grad[o:o + n].view(p.shape))
                p.grad = self.grad_views[i]
                # The stage backwards (functional._Sink) write parameter gradients STRAIGHT into these slots (first
                # contribution of a step: the kernel's output buffer is the slot; later ones - AR rollout, the model
                # runs T_ar times - are added in place) and return None to autograd, so there is no per-parameter
                # accumulate kernel.  p.grad stays bound to the slot, so foreign autograd graphs still accumulate
                # into the flat buffer the ordinary way.
                p._dpot_grad_slot = (self, i)

    def zero_grad(self) -> None:
        self.grad.zero_()
        for i, p in enumerate(self.params):
            self.filled[i] = False
            self.pending[i] = 0
            if p.grad is None or p.grad.data_ptr() != self.grad_views[i].data_ptr():
                p.grad = self.grad_views[i]

    def fire(self, i: int) -> None:
        for cb in self.callbacks:
            cb(i)


class FusedAdam:
    """Adam with L2 weight decay folded into the gradient (utils/optimizer.py:9-52 semantics) + global-norm clip
    (train_temporal.py:228), as ONE kernel over the flat buffer.  ``update_tail=False`` reproduces the single-GPU
    reference where cls_head has no gradient and is therefore skipped by the optimiser.

    Host/device protocol of a step: ``stage_hyper(lr)`` enqueues a one-thread kernel that receives lr / betas / eps /
    weight decay / max_norm BY VALUE, advances the DEVICE-side step counter and derives the bias corrections from
    it; ``launch()`` (capturable) enqueues ||g||^2 + the fused clip+Adam kernel, which read those values from device
    memory.  There is no host staging buffer, so a host that runs many (graph-replayed) steps ahead of the GPU
    cannot disturb a step that is still queued."""

    def __init__(self, flat: FlatParams, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0, max_norm: Optional[float] = None,
                 update_tail: bool = False):
        self.fp = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_norm = max_norm
        self.update_tail = update_tail
        dev = flat.flat.device
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)     # the step counter the kernels use
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._part = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.step_count = 0                       # host mirror of step_dev (number of steps ENQUEUED)
        self.param_groups = [{"lr": lr}]          # minimal torch.optim surface for LR schedulers / logging

    @property
    def n_active(self) -> int:
        return self.fp.total if self.update_tail else self.fp.n_head

    def zero_grad(self) -> None:
        self.fp.zero_grad()

    def stage_hyper(self, lr: Optional[float] = None, advance: int = 1) -> None:
        """host side of a step (NOT capturable by design: call it right before a graph replay)"""
        if lr is not None:
            self.lr = lr
            self.param_groups[0]["lr"] = lr
        self.step_count += advance
        ops.adam_stage(self.hyper, self.step_dev, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                       self.max_norm if self.max_norm is not None else 0.0, advance)

    def launch(self, grad_scale: float = 1.0) -> None:
        """device side of a step (capturable): ||g||^2 then fused clip + Adam."""
        n = self.n_active
        g = self.fp.grad[:n]
        use_clip = self.max_norm is not None
        if use_clip:
            ops.sumsq(g, self.sumsq, self._part)
        plan = self._pack_plan()
        if plan is not None:
            # round 6: the channel-MLP weights leave the optimiser ALSO as their two bf16 packs - the next forward's pack
            # launch (a second read of every weight written here) disappears (PanelPacks.refresh sees them fresh)
            ops.adam_step_packs(plan, self.fp.flat, self.fp.grad, self.exp_avg, self.exp_avg_sq, self.hyper,
                                self.sumsq if use_clip else None, grad_scale)
        else:
            ops.adam_step(self.fp.flat[:n], g, self.exp_avg[:n], self.exp_avg_sq[:n], self.hyper,
                          self.sumsq if use_clip else None, grad_scale)
        self.fp.epoch += 1            # parameters changed: a backward of an EARLIER forward must not run any more
        if plan is not None:
            plan.pp.mark_fresh(self.fp)

    def mark_packs_fresh(self) -> None:
        """after a graph replay whose captured Adam launch wrote the weight packs (the host-side record cannot see it)"""
        plan = self._pack_plan()
        if plan is not None:
            plan.pp.mark_fresh(self.fp)

    def _pack_plan(self):
        """the dpot_adam_step_packs tables for the model's plain-bf16 channel-MLP weight packs (DPOTNet._panel_packs_bf16, made
        by its first forward in that mode), or None: no such packs / a weight that does not tile / DPOT_TUNE packs=0"""
        if ops.tune("packs") == 0:
            return None
        pp = getattr(self.fp.model, "_panel_packs_bf16", None)
        if pp is None:
            self._plan = None
            return None
        key = (self.fp.flat.data_ptr(), self.n_active, id(pp))
        plan = getattr(self, "_plan", None)
        if plan is None or plan.key != key:
            plan = self._plan = ops.AdamPackPlan.build(self.fp.flat, self.n_active, pp) or False
            if plan:
                plan.key = key
        return plan or None

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0) -> None:
        self.stage_hyper(lr)
        self.launch(grad_scale)

    def grad_norm(self, grad_scale: float = 1.0) -> Tensor:
        """global norm of the gradients currently in the flat buffer (device tensor; reading it synchronises):
        after a step, the value its clip used (train_temporal.py:228 returns it from clip_grad_norm_)"""
        out = torch.empty_like(self.sumsq)
        ops.sumsq(self.fp.grad[:self.n_active], out, self._part)
        return out.sqrt() * grad_scale

    # -- snapshot / restore (graph warm-up, tests) --------------------------------------------------------
    def snapshot(self):
        return (self.fp.flat.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.step_dev.clone(),
                self.step_count, self.lr)

    def restore(self, snap) -> None:
        flat, m, v, sd, sc, lr = snap
        self.fp.flat.copy_(flat)
        self.exp_avg.copy_(m)
        self.exp_avg_sq.copy_(v)
        self.step_dev.copy_(sd)
        self.step_count, self.lr = sc, lr
        self.param_groups[0]["lr"] = lr
        self.fp.epoch += 1                       # parameters changed behind the packs' back ...
        plan = self._pack_plan()
        if plan is not None:                     # ... so bring the optimiser-owned weight packs back in line right away: a graph
            plan.pp.refresh(force=True)          # captured after this (GraphedTrainStep warm-up) relies on them being fresh
            plan.pp.mark_fresh(self.fp)

    # -- checkpoint format of the reference: torch.optim state_dict (train_temporal.py:244,281) ------------
    def state_dict(self, model: nn.Module) -> dict:
        """{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} with i = index of the parameter
        in ``model.parameters()`` order - what ``torch.save({'optimizer': optimizer.state_dict()})`` of the
        reference's Adam (utils/optimizer.py:101-164) holds.  Parameters the optimiser never updates (cls_head in
        single-GPU training) have no state entry, as in the reference."""
        order = {id(p): i for i, p in enumerate(model.parameters())}
        n_act = self.n_active
        state = {}
        step = int(self.step_dev.item())
        for p, off in zip(self.fp.params, self.fp.offsets):
            if off >= n_act or step == 0:
                continue
            n = p.numel()
            state[order[id(p)]] = {"step": step,
                                   "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                                   "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict, model: nn.Module) -> None:
        order = {id(p): i for i, p in enumerate(model.parameters())}
        steps = set()
        with torch.no_grad():
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            for p, off in zip(self.fp.params, self.fp.offsets):
                st = sd["state"].get(order[id(p)])
                if st is None:
                    continue
                n = p.numel()
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused optimiser keeps one")
        step = steps.pop() if steps else 0
        self.step_dev.fill_(step)
        self.step_count = step
        g = sd.get("param_groups", [{}])[0]
        self.lr = g.get("lr", self.lr)
        self.betas = tuple(g.get("betas", self.betas))
        self.eps = g.get("eps", self.eps)
        self.weight_decay = g.get("weight_decay", self.weight_decay)
        self.param_groups[0]["lr"] = self.lr


# ------------------------------------------------------------------------------------------------------
def one_cycle_lr(step: int, total_steps: int, max_lr: float, pct_start: float = 0.3, div_factor: float = 1e4,
                 final_div_factor: float = 1e4) -> float:
    """torch.optim.lr_scheduler.OneCycleLR (cos annealing, two phases) as a pure function of the step index;
    train_temporal.py:140 uses div_factor=1e4, final_div_factor=1e4, pct_start=warmup_epochs/epochs."""
    initial = max_lr / div_factor
    min_lr = initial / final_div_factor
    up_end = float(pct_start * total_steps) - 1.0
    down_end = float(total_steps) - 1.0

    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    if step <= up_end:
        return cos(initial, max_lr, step / up_end if up_end > 0 else 1.0)
    return cos(max_lr, min_lr, (step - up_end) / (down_end - up_end))


# ------------------------------------------------------------------------------------------------------
def rollout(model: nn.Module, xx: Tensor, yy: Tensor, msk: Optional[Tensor], T_bundle: int = 1,
            noise_scale: float = 0.0, noise: Optional[List[Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """auto-regressive rollout with summed per-step loss (train_temporal.py:201-219)"""
    scope = model.weights_scope() if hasattr(model, "weights_scope") else contextlib.nullcontext()
    with scope:
        return _rollout(model, xx, yy, msk, T_bundle, noise_scale, noise)


def _rollout(model, xx, yy, msk, T_bundle, noise_scale, noise):
    loss = None
    preds = []
    T_ar = yy.shape[-2]
    xx = ops._req(xx.contiguous(), "xx")
    n_steps = len(range(0, T_ar, T_bundle))
    tell = hasattr(model, "_ar_pos")
    for k, t in enumerate(range(0, T_ar, T_bundle)):
        if tell:
            model._ar_pos = (k, n_steps)            # DPOTNet.recompute_keep_last: which AR steps keep their activations
        y = yy[..., t:t + T_bundle, :]
        if noise_scale != 0.0:
            if noise is None and not xx.requires_grad and xx.numel() % 4 == 0:
                # first AR step of training: nothing to differentiate -> draw the noise inside the kernel
                xx = ops.noise_inject(xx, None, noise_scale)
            else:
                xx = _NoiseFn.apply(xx, noise[k] if noise is not None else None, noise_scale)
        try:
            im, _ = model(xx)
        finally:
            if tell:
                model._ar_pos = None                # (never left behind: a later plain forward must not see a stale position)
        l = rel_l2_loss(im, y, msk)
        loss = l if loss is None else loss + l
        preds.append(im)
        if t + T_bundle < T_ar:
            xx = _SlideFn.apply(xx, im)
    pred = preds[0] if len(preds) == 1 else torch.cat(preds, dim=-2)
    return loss, pred


class _NoiseFn(torch.autograd.Function):
    """xx + noise_scale * ||xx|| * eps (train_temporal.py:205) for AR steps whose input carries a gradient.  Like
    the reference's autograd, the backward differentiates through the norm as well:
    d/dxx = g + s * xx / ||xx|| * sum(g * eps)  (csrc/loss_opt.hip noise_bwd_*).  eps=None: drawn by the in-kernel
    generator in forward and RE-drawn in backward from a 16-byte copy of the generator state (no eps tensor)."""

    @staticmethod
    def forward(ctx, xx, eps, noise_scale):
        xx = xx.contiguous()
        rng = None
        if eps is None and xx.numel() % 4 != 0:
            eps = torch.randn_like(xx)                      # shapes the in-kernel generator does not cover
        if eps is not None:
            eps = eps.contiguous()
        out, norms = ops.noise_inject(xx, eps, noise_scale, return_norms=True)
        if eps is None:
            rng = ops.rng_state(xx.device).clone()         # {seed, offset} this call drew from
        ctx.save_for_backward(xx, eps if eps is not None else xx.new_empty(0), norms,
                              rng if rng is not None else xx.new_empty(0))
        ctx.noise_scale, ctx.has_eps = noise_scale, eps is not None
        return out

    @staticmethod
    def backward(ctx, g):
        xx, eps, norms, rng = ctx.saved_tensors
        dx = ops.noise_inject_bwd(xx, eps if ctx.has_eps else None, None if ctx.has_eps else rng, g.contiguous(),
                                  norms, ctx.noise_scale)
        return dx, None, None


class _SlideFn(torch.autograd.Function):
    """xx <- cat(xx[..., T_bundle:, :], im) (train_temporal.py:219) as one kernel each way"""

    @staticmethod
    def forward(ctx, xx, im):
        ctx.Tb = im.shape[-2]
        return ops.window_slide(ops._req(xx.contiguous(), "xx"), ops._req(im.contiguous(), "im"))

    @staticmethod
    def backward(ctx, dout):
        dxx, dim = ops.window_slide_bwd(dout.contiguous(), ctx.Tb, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dxx, dim


def train_step(model: nn.Module, opt: FusedAdam, xx: Tensor, yy: Tensor, msk: Optional[Tensor], T_bundle: int = 1,
               noise_scale: float = 0.0, lr: Optional[float] = None, reducer=None,
               grad_scale: float = 1.0) -> Tuple[Tensor, Tensor]:
    """one optimisation step; returns (loss, pred) as device tensors - no host sync."""
    opt.zero_grad()
    if reducer is not None:
        reducer.begin_step()
    loss, pred = rollout(model, xx, yy, msk, T_bundle, noise_scale)
    loss.backward()
    clear_grad_packs()
    if reducer is not None:
        reducer.finish()
    opt.step(lr, grad_scale)
    return loss.detach(), pred.detach()


# ------------------------------------------------------------------------------------------------------
def _retire_collective_watchdog_work(seconds: float = 0.4) -> None:
    """Before a capture that will contain collectives: let the process group's watchdog thread retire the work items of the
    LIVE collectives issued so far (the eager warm-up).  torch's nccl backend keeps every eager collective on a list that its
    watchdog polls (hipEventQuery on the collective's end event, every ~100 ms) until the event has completed; the capture then
    pulls the backend's internal communication stream into capture mode, and HIP refuses a query of an event on a capturing
    stream (hipErrorCapturedEvent) - the watchdog thread throws and the PROCESS aborts.  Seen once in ~10 runs of the one-rank
    RCCL test (round 6).  With the device idle the watchdog needs one of its periods to drop the finished items; there is no
    API to ask it, so: synchronise, then sleep a few periods.  (Inside the capture torch does not enqueue work items at all.)"""
    import time
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    torch.cuda.synchronize()
    time.sleep(seconds)


class GraphedTrainStep:
    """Captures forward + loss + backward + clip + Adam of a fixed-shape batch into one hipGraph.

    ``stage(xx, yy, msk)`` copies a batch into the static input buffers, ``replay(lr)`` runs the step.  The
    ~500 kernel launches of a DPOT-Tiny step then cost one graph launch instead of ~3 ms of Python/HIP launch
    overhead (the GPU work itself is ~4 ms at B=32)."""

    def __init__(self, model: nn.Module, opt: FusedAdam, xx: Tensor, yy: Tensor, msk: Optional[Tensor],
                 T_bundle: int = 1, noise_scale: float = 0.0, warmup: int = 2, reducer=None,
                 grad_scale: float = 1.0, capture_collectives: bool = False):
        self.model, self.opt = model, opt
        self.xx, self.yy = xx.clone(), yy.clone()
        self.msk = msk.clone() if msk is not None else None
        if reducer is not None and getattr(reducer, "world", 1) > 1 and not capture_collectives:
            raise ValueError("GraphedTrainStep captures the WHOLE step into one graph; pass capture_collectives=True to "
                             "capture the bucket all-reduces with it, or use SegmentedTrainStep (collectives between graphs)")
        # capture_collectives (opt-in): the hook-driven bucket all-reduces are issued INSIDE the capture - the reducer's side
        # stream forks from / joins the capturing stream, RCCL's kernels become nodes of the one graph; no cut in the
        # autograd graph, no graph launches between the buckets
        self.reducer = reducer if capture_collectives else None
        if self.reducer is not None:
            if grad_scale != 1.0 and grad_scale != self.reducer.grad_scale:
                raise ValueError(f"GraphedTrainStep(capture_collectives=True) scales the gradient by the reducer's 1 / world = "
                                 f"{self.reducer.grad_scale}; a different grad_scale ({grad_scale}) was passed")
            grad_scale = self.reducer.grad_scale
            # the capture bakes in WHICH buckets are all-reduced: always include the cls_head tail (a zero slice when the loss
            # ignores it - the SUM of zeros - so the replayed graph stays right if the head starts to receive gradients later)
            self.reducer.skip_zero_tail = False
        self.grad_scale = grad_scale
        self.T_bundle, self.noise_scale = T_bundle, noise_scale
        # eager warm-up on a side stream (allocator + lazy inits).  The warm-up iterations are NOT training steps:
        # parameters, Adam moments and the step counter are restored afterwards (a fine-tune of a pretrained
        # checkpoint must not receive unscheduled full-lr updates before its first replay)
        snap = opt.snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body(stage=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.reducer is not None:
            _retire_collective_watchdog_work()
        opt.restore(snap)
        self.graph = torch.cuda.CUDAGraph()
        opt.zero_grad()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss, self.pred = self._body(stage=False)
        self.warmup_steps = warmup

    def _body(self, stage: bool):
        opt = self.opt
        if stage:
            opt.stage_hyper(opt.lr)
        opt.zero_grad()
        if self.reducer is not None:
            self.reducer.begin_step()
        loss, pred = rollout(self.model, self.xx, self.yy, self.msk, self.T_bundle, self.noise_scale)
        loss.backward()
        clear_grad_packs()
        if self.reducer is not None:
            self.reducer.finish()
        opt.launch(self.grad_scale)
        return loss.detach(), pred.detach()

    def stage(self, xx: Tensor, yy: Tensor, msk: Optional[Tensor] = None) -> None:
        self.xx.copy_(xx, non_blocking=True)
        self.yy.copy_(yy, non_blocking=True)
        if msk is not None and self.msk is not None:
            self.msk.copy_(msk, non_blocking=True)

    def replay(self, lr: Optional[float] = None) -> Tensor:
        self.opt.stage_hyper(lr)
        self.graph.replay()
        self.opt.fp.epoch += 1        # the captured Adam + pack launches rewrote the parameters and their packed copies
        self.opt.mark_packs_fresh()   # (the captured Adam wrote the weight packs too, if it owned them at capture time)
        return self.loss


class SegmentedTrainStep:
    """The data-parallel train step (any T_ar, as train_temporal_parallel.py:214-244 runs it under DDP) as a CHAIN of
    hipGraphs, cut where a gradient bucket becomes final, so that the RCCL all-reduce of bucket k runs on the reducer's side stream while the compute stream
    replays the backward of the earlier stages - the overlap torch-DDP gets from autograd hooks
    (train_temporal_parallel.py:244), without a collective inside a capture and without eager launch overhead.

        segment 0      zero_grad, forward, loss, backward of [out layer .. first stage of the LAST bucket]
        segment j      backward of the stages of the j-th bucket from the end
        optimiser      ||g||^2 + fused clip + Adam          (after the compute stream has joined the side stream)

    The autograd graph is cut at the bucket boundaries (``DPOTNet._boundary_hook``): each cut replaces the latent by a
    detached leaf; the next segment continues with ``out.backward(leaf.grad)``.  All segments share one memory pool
    and are always replayed in capture order.

    Auto-regressive rollouts (T_ar > T_bundle): every AR step adds to every parameter gradient, and the backward visits
    the AR steps last to first - a bucket is final only once the backward of the FIRST AR step has passed it.  The cuts
    are therefore made in the first forward call only; segment 0 holds the whole rollout forward, the loss, the complete
    backward of AR steps T-1 .. 1 and the backward of AR step 0 down to its last cut."""

    def __init__(self, model: nn.Module, opt: FusedAdam, reducer, xx: Tensor, yy: Tensor, msk: Optional[Tensor],
                 noise_scale: float = 0.0, warmup: int = 2, T_bundle: int = 1):
        self.T_bundle = T_bundle
        self.model, self.opt, self.reducer = model, opt, reducer
        self.xx, self.yy = xx.clone(), yy.clone()
        self.msk = msk.clone() if msk is not None else None
        self.noise_scale = noise_scale
        self.grad_scale = reducer.grad_scale
        depth = len(model.blocks)
        norm = lambda st: depth + 1 if st == 10 ** 6 else st          # stage number -> boundary index
        # buckets in backward order (tail excluded); a bucket whose first stage is b > 0 needs a cut at boundary b
        self.bwd_buckets = [k for k in range(reducer.n_buckets - 1, -1, -1) if k != reducer.tail_bucket]
        self.cut_at = sorted({norm(reducer.first_stage_of_bucket(k)) for k in self.bwd_buckets} - {0})
        snap = opt.snapshot()
        reducer.detach()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    opt.stage_hyper(opt.lr)
                    for fn in self._segment_fns():
                        fn()
                    opt.launch(self.grad_scale)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            opt.restore(snap)
            self.graphs: List[torch.cuda.CUDAGraph] = []
            pool = torch.cuda.graph_pool_handle()
            for fn in self._segment_fns():
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                    fn()
                self.graphs.append(g)
            self.opt_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.opt_graph, pool=pool, capture_error_mode="thread_local"):
                opt.launch(self.grad_scale)
        finally:
            reducer.attach()
            model._boundary_hook = None
        self.tail_has_grad = reducer.tail_bucket >= 0 and any(opt.fp.filled[i]
                                                              for i in reducer.members[reducer.tail_bucket])
        assert len(self.graphs) == len(self.bwd_buckets), (len(self.graphs), self.bwd_buckets)

    def _segment_fns(self):
        """generator of the segment bodies; segment j+1 may only be built after segment j has run"""
        cuts = []
        cut_at = set(self.cut_at)
        calls = [0]                                  # forward calls of this rollout seen so far (hook(1, .) opens one)

        def hook(b, lat):
            if b == 1:
                calls[0] += 1
            if calls[0] == 1 and b in cut_at and lat.requires_grad:
                leaf = lat.detach().requires_grad_(True)
                cuts.append((lat, leaf))
                return leaf
            return lat

        def first():
            self.opt.zero_grad()
            self.model._boundary_hook = hook
            calls[0] = 0
            try:
                loss, pred = rollout(self.model, self.xx, self.yy, self.msk, self.T_bundle, self.noise_scale)
            finally:
                self.model._boundary_hook = None
            assert len(cuts) == len(self.cut_at), f"expected {len(self.cut_at)} graph cuts, the forward made {len(cuts)}"
            loss.backward()
            clear_grad_packs()
            self.loss, self.pred = loss.detach(), pred.detach()

        yield first
        # one further segment per cut, from the last cut to the first
        n = len(self.cut_at)
        for j in range(n - 1, -1, -1):
            def seg(j=j):
                out, leaf = cuts[j]
                out.backward(leaf.grad)
                clear_grad_packs()
                leaf.grad = None
            yield seg

    def stage(self, xx: Tensor, yy: Tensor, msk: Optional[Tensor] = None) -> None:
        self.xx.copy_(xx, non_blocking=True)
        self.yy.copy_(yy, non_blocking=True)
        if msk is not None and self.msk is not None:
            self.msk.copy_(msk, non_blocking=True)

    def replay(self, lr: Optional[float] = None) -> Tensor:
        red = self.reducer
        self.opt.stage_hyper(lr)
        red.begin_step()
        for g, k in zip(self.graphs, self.bwd_buckets):
            g.replay()
            red.reduce_bucket(k)              # side stream waits for the compute stream, then all-reduces bucket k
        if self.tail_has_grad:
            red.reduce_bucket(red.tail_bucket)
        elif red.tail_bucket >= 0:
            red._launched[red.tail_bucket] = True
        red.finish()                          # compute stream joins the side stream
        self.opt_graph.replay()
        self.opt.fp.epoch += 1                # as FusedAdam.launch: an eager backward of an earlier forward must fail
        self.opt.mark_packs_fresh()
        return self.loss


# ------------------------------------------------------------------------------------------------------
def _grad_checksum(g: Tensor) -> Tensor:
    """order-independent 64-bit checksum of the BIT patterns of a flat fp32 buffer (int64[2]: sum and sum of squares of the
    words mod 2^64) - equal buffers give equal checksums; used across ranks (after an all-reduce every rank must hold the same
    gradient bit for bit)"""
    w = g.view(torch.int32).to(torch.int64)
    return torch.stack([w.sum(), (w * w).sum()])


def _end_stray_captures(streams) -> None:
    """best effort after a FAILED stream capture: a stream that joined the capture (the reducer's side stream) can be left in
    capture mode when the capture is torn down by an exception; end it through the HIP runtime and clear the sticky error"""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return
    for st in streams:
        if st is None:
            continue
        g = ctypes.c_void_p()
        hip.hipStreamEndCapture(ctypes.c_void_p(st.cuda_stream), ctypes.byref(g))
        if g.value:
            hip.hipGraphDestroy(g)
    hip.hipGetLastError()
    try:
        torch.cuda.synchronize()
    except Exception:                                          # pragma: no cover
        pass


def make_dp_step(model: nn.Module, opt: FusedAdam, reducer, xx: Tensor, yy: Tensor, msk: Optional[Tensor],
                 noise_scale: float = 0.0, warmup: int = 2, T_bundle: int = 1, try_one_graph: bool = True):
    """The N > 1 train step with the FAST mode chosen by verification (round 6): returns (step, info).

    Fast mode = ``GraphedTrainStep(capture_collectives=True)``: the whole step, bucket all-reduces included, as ONE hipGraph
    (+0.4 % over the single-GPU graph at DPOT-Tiny against +4 % for the chain).  Fallback = ``SegmentedTrainStep`` (graph
    segments, the all-reduces issued between them: collectives never sit inside a capture).  Whether the communication
    library's kernels replay correctly as graph nodes across ranks cannot be known in advance, so the one-graph step is
    captured and ONE trial step is run twice from the same snapshot (parameters, Adam state, noise generator): eagerly with the
    hook-driven reducer (``train_step`` - the same kernels in the same order, collectives issued live) and as a replay of the
    graph.  The reduced flat gradient must agree BIT FOR BIT between the two and across the ranks (checksums through the process
    group); only then is the one-graph step returned, otherwise the segmented chain is built and ``info`` says why.  (The chain
    itself is not the reference of the comparison: with the bf16 channel MLP its graph cuts change which kernel packs a
    Block's incoming gradient, and six bias gradients differ in the last bit - tests/test_gpu_train2.py.)  Parameters,
    optimiser state and the noise generator are left exactly as found.  Mirrors train_temporal_parallel.py:102,185,243-244."""
    import torch.distributed as dist

    def chain(why):
        reducer.skip_zero_tail = tail_flag
        seg = SegmentedTrainStep(model, opt, reducer, xx, yy, msk, noise_scale=noise_scale, warmup=warmup, T_bundle=T_bundle)
        return seg, {"mode": "segmented", "why": why}

    tail_flag = reducer.skip_zero_tail
    if not try_one_graph:
        return chain("one-graph mode not tried")
    multi = dist.is_available() and dist.is_initialized()
    backend = dist.get_backend(reducer.pg) if multi else "none"
    if multi and backend != "nccl":
        # only RCCL's collectives are device work that a stream capture can record; a gloo collective synchronises the stream
        # from the host - attempting the capture would leave the streams that joined it in a broken capture state
        return chain(f"one-graph capture failed: backend '{backend}' collectives cannot be recorded in a hipGraph (RCCL only)")
    one, err = None, None
    try:
        one = GraphedTrainStep(model, opt, xx, yy, msk, T_bundle=T_bundle, noise_scale=noise_scale, warmup=max(1, warmup),
                               reducer=reducer, capture_collectives=True)
    except Exception as e:                                    # an RCCL build / driver that cannot capture its collectives
        err = f"{type(e).__name__}: {e}"[:160]
        _end_stray_captures([torch.cuda.current_stream(), reducer.stream])
    # every rank must take the same branch: agree on whether the capture worked everywhere
    ok = torch.tensor([1 if one is not None else 0], device=xx.device, dtype=torch.int64)
    if multi:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=reducer.pg)
    if int(ok.item()) == 0:
        del one
        return chain("one-graph capture failed" + (f" here ({err})" if err else " on another rank"))
    snap = opt.snapshot()
    rng = ops.rng_state(xx.device).clone()
    grads = []
    for mode in ("eager", "graph"):
        opt.restore(snap)
        ops.rng_state(xx.device).copy_(rng)
        reducer.skip_zero_tail = False                        # (as the capture: the cls_head tail is always reduced)
        if mode == "eager":
            train_step(model, opt, one.xx, one.yy, one.msk, T_bundle=T_bundle, noise_scale=noise_scale, lr=opt.lr,
                       reducer=reducer, grad_scale=reducer.grad_scale)
        else:
            one.replay(opt.lr)
        torch.cuda.synchronize()
        grads.append(opt.fp.grad[:opt.n_active].clone())
    opt.restore(snap)
    ops.rng_state(xx.device).copy_(rng)
    same_modes = bool(torch.equal(grads[0], grads[1]))
    cs = _grad_checksum(grads[1])
    verdict = torch.tensor([1 if same_modes else 0], device=xx.device, dtype=torch.int64)
    same_ranks = True
    if multi:
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=reducer.pg)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=reducer.pg)
        same_ranks = bool(torch.equal(lo, hi))
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=reducer.pg)
    if int(verdict.item()) == 1 and same_ranks:
        return one, {"mode": "one-graph", "why": "reduced gradient of a trial step bit-identical to the eager hook-driven step's "
                                                  "and across the ranks"}
    del one
    return chain("one-graph trial step " + ("differs from the eager step's gradient" if int(verdict.item()) == 0 else
                                            "gave different gradients on different ranks"))
