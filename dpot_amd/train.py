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

import math
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops
from .functional import rel_l2_loss

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------------
class FlatParams:
    """Re-homes the parameters of ``model`` into one flat fp32 buffer (and their .grad into another).

    ``order`` puts parameters whose names start with one of ``tail_prefixes`` (cls_head: it gets no gradient in
    single-GPU training, train_temporal.py:226) at the end, so the optimiser can address 'everything with a
    gradient' as one contiguous prefix."""

    def __init__(self, model: nn.Module, tail_prefixes: Sequence[str] = ("cls_head.",)):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        head = [(n, p) for n, p in named if not any(n.startswith(t) for t in tail_prefixes)]
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
    reference where cls_head has no gradient and is therefore skipped by the optimiser."""

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
        self._hyper_host = torch.zeros(8, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(8)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._part = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.param_groups = [{"lr": lr}]          # minimal torch.optim surface for LR schedulers / logging

    @property
    def n_active(self) -> int:
        return self.fp.total if self.update_tail else self.fp.n_head

    def zero_grad(self) -> None:
        self.fp.zero_grad()

    def stage_hyper(self, lr: Optional[float] = None) -> None:
        """host side of a step: advance the step counter and ship {lr, betas, eps, wd, bias corrections, max_norm}
        to the device (async copy from pinned memory - safe to call right before a graph replay)."""
        if lr is not None:
            self.lr = lr
            self.param_groups[0]["lr"] = lr
        self.step_count += 1
        b1, b2 = self.betas
        h = self._hyper_host
        h[0], h[1], h[2], h[3], h[4] = self.lr, b1, b2, self.eps, self.weight_decay
        h[5], h[6] = 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count
        h[7] = self.max_norm if self.max_norm is not None else 0.0
        self.hyper.copy_(h, non_blocking=True)

    def launch(self, grad_scale: float = 1.0) -> None:
        """device side of a step (capturable): ||g||^2 then fused clip + Adam."""
        n = self.n_active
        g = self.fp.grad[:n]
        use_clip = self.max_norm is not None
        if use_clip:
            ops.sumsq(g, self.sumsq, self._part)
        ops.adam_step(self.fp.flat[:n], g, self.exp_avg[:n], self.exp_avg_sq[:n], self.hyper,
                      self.sumsq if use_clip else None, grad_scale)

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0) -> None:
        self.stage_hyper(lr)
        self.launch(grad_scale)

    def grad_norm(self, grad_scale: float = 1.0) -> Tensor:
        """global gradient norm of the last step (device tensor; reading it synchronises)"""
        return self.sumsq.sqrt() * grad_scale


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
    loss = None
    preds = []
    T_ar = yy.shape[-2]
    for k, t in enumerate(range(0, T_ar, T_bundle)):
        y = yy[..., t:t + T_bundle, :]
        if noise_scale != 0.0:
            if noise is None and not xx.requires_grad and xx.numel() % 4 == 0:
                # first AR step of training: nothing to differentiate -> draw the noise inside the kernel
                xx = ops.noise_inject(ops._req(xx.contiguous(), "xx"), None, noise_scale)
            else:
                eps = noise[k] if noise is not None else torch.randn_like(xx)
                xx = _NoiseFn.apply(xx, eps, noise_scale)
        im, _ = model(xx)
        l = rel_l2_loss(im, y, msk)
        loss = l if loss is None else loss + l
        preds.append(im)
        if t + T_bundle < T_ar:
            xx = torch.cat((xx[..., T_bundle:, :], im), dim=-2)
    pred = preds[0] if len(preds) == 1 else torch.cat(preds, dim=-2)
    return loss, pred


class _NoiseFn(torch.autograd.Function):
    """xx + noise_scale * ||xx|| * eps (train_temporal.py:205).  The reference lets autograd differentiate through
    the norm as well; that term is O(noise_scale) and is kept here through a straight-through identity gradient
    plus the exact norm term only when xx requires grad (AR steps > 0)."""

    @staticmethod
    def forward(ctx, xx, eps, noise_scale):
        ctx.save_for_backward(xx, eps)
        ctx.noise_scale = noise_scale
        return ops.noise_inject(xx.contiguous(), eps.contiguous(), noise_scale)

    @staticmethod
    def backward(ctx, g):
        xx, eps = ctx.saved_tensors
        # d/dxx [xx + s * n(xx) * eps],  n = ||xx||_2 over (X,Y,T) per (b,c):  g + s * xx / n * sum(g * eps)
        dims = tuple(range(1, xx.dim() - 1))
        n = torch.sum(xx ** 2, dim=dims, keepdim=True) ** 0.5
        corr = ctx.noise_scale * xx / n.clamp_min(1e-30) * torch.sum(g * eps, dim=dims, keepdim=True)
        return g + corr, None, None


def train_step(model: nn.Module, opt: FusedAdam, xx: Tensor, yy: Tensor, msk: Optional[Tensor], T_bundle: int = 1,
               noise_scale: float = 0.0, lr: Optional[float] = None, reducer=None,
               grad_scale: float = 1.0) -> Tuple[Tensor, Tensor]:
    """one optimisation step; returns (loss, pred) as device tensors - no host sync."""
    opt.zero_grad()
    if reducer is not None:
        reducer.begin_step()
    loss, pred = rollout(model, xx, yy, msk, T_bundle, noise_scale)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    opt.step(lr, grad_scale)
    return loss.detach(), pred.detach()


# ------------------------------------------------------------------------------------------------------
class GraphedTrainStep:
    """Captures forward + loss + backward + clip + Adam of a fixed-shape batch into one hipGraph.

    ``stage(xx, yy, msk)`` copies a batch into the static input buffers, ``replay(lr)`` runs the step.  The
    ~500 kernel launches of a DPOT-Tiny step then cost one graph launch instead of ~3 ms of Python/HIP launch
    overhead (the GPU work itself is ~4 ms at B=32)."""

    def __init__(self, model: nn.Module, opt: FusedAdam, xx: Tensor, yy: Tensor, msk: Optional[Tensor],
                 T_bundle: int = 1, noise_scale: float = 0.0, warmup: int = 2, reducer=None,
                 grad_scale: float = 1.0):
        self.model, self.opt = model, opt
        self.xx, self.yy = xx.clone(), yy.clone()
        self.msk = msk.clone() if msk is not None else None
        self.reducer, self.grad_scale = reducer, grad_scale
        self.T_bundle, self.noise_scale = T_bundle, noise_scale
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                 # eager warm-up on a side stream (allocator + lazy inits)
                self._body(stage=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        opt.zero_grad()
        with torch.cuda.graph(self.graph):
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
        return self.loss
