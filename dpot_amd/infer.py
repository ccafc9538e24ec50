"""Checkpoint loading and auto-regressive inference for the MI355X DPOTNet.

* ``load_model_from_checkpoint`` / ``load_components_from_pretrained`` - the contracts of the reference's
  utils/utilities.py:99-166 (``module.`` prefix of DDP checkpoints stripped; fine-tuning may take only some
  components from a pretrained state_dict), written against the same sub-module names, so a pretrained ``.pth``
  (``torch.load(path)['model']``, README.md:30) loads unchanged.
* ``rollout_eval`` / ``GraphedRollout`` - the no-grad rollout of evaluate.py:183-222: the model's own prediction is
  appended to the input window step after step; returns the prediction, the per-step loss sum and the loss of the
  whole trajectory (both SimpleLpLoss(size_average=False)).  With fixed shapes the single forward step is one
  hipGraph that is replayed T_ar / T_bundle times (the window slide writes into the graph's static input).
"""
from __future__ import annotations

import contextlib
import warnings
from collections import OrderedDict
from typing import Dict, Iterable, Mapping, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import ops
from .functional import rel_l2_loss

Tensor = torch.Tensor
COMPONENTS = ("patch_embed", "pos", "blocks", "time_agg", "cls_head", "scale_feats", "out")


def _plain_state_dict(sd: Union[str, Mapping]) -> "OrderedDict[str, Tensor]":
    """accepts a path, a checkpoint dict with a 'model' entry, or a state_dict; strips DDP's 'module.' prefix"""
    if isinstance(sd, (str, bytes)):
        sd = torch.load(sd, map_location="cpu", weights_only=False)
    if "model" in sd and isinstance(sd["model"], Mapping) and not isinstance(sd["model"], Tensor):
        sd = sd["model"]
    if len(sd) and next(iter(sd.keys())).startswith("module."):
        sd = OrderedDict((k.replace("module.", ""), v) for k, v in sd.items())
    return OrderedDict(sd)


def load_model_from_checkpoint(model: nn.Module, model_state_dict: Union[str, Mapping]) -> None:
    """utils/utilities.py:99-109"""
    model.load_state_dict(_plain_state_dict(model_state_dict))


def _sub(sd: Mapping[str, Tensor], prefix: str) -> "OrderedDict[str, Tensor]":
    return OrderedDict((k[len(prefix):], v) for k, v in sd.items() if k.startswith(prefix))


def load_components_from_pretrained(model: nn.Module, state_dict: Union[str, Mapping],
                                    components: Union[str, Iterable[str]] = "all") -> None:
    """utils/utilities.py:112-166: components = 'all' or a subset of COMPONENTS.  Must run BEFORE the optimiser's
    FlatParams is built (the reference replaces the pos_embed Parameter object; here its data is copied in place so
    that even an existing flat binding stays valid)."""
    sd = _plain_state_dict(state_dict)
    if components == "all" or "all" in components:
        model.load_state_dict(sd)
        return
    for name in components:
        if name == "patch_embed" and hasattr(model, "patch_embed"):
            model.patch_embed.load_state_dict(_sub(sd, "patch_embed."))
        elif name == "pos" and hasattr(model, "pos_embed"):
            with torch.no_grad():
                src = sd["pos_embed"]
                if tuple(src.shape) != tuple(model.pos_embed.shape):
                    raise RuntimeError(f"pos_embed shape {tuple(src.shape)} != {tuple(model.pos_embed.shape)}")
                model.pos_embed.copy_(src)
        elif name == "blocks" and hasattr(model, "blocks"):
            for i, block in enumerate(model.blocks):
                block.load_state_dict(_sub(sd, f"blocks.{i}."))
        elif name == "scale_feats" and hasattr(model, "scale_feats_mu"):
            model.scale_feats_mu.load_state_dict(_sub(sd, "scale_feats_mu."))
            model.scale_feats_sigma.load_state_dict(_sub(sd, "scale_feats_sigma."))
        elif name == "cls_head" and hasattr(model, "cls_head"):
            model.cls_head.load_state_dict(_sub(sd, "cls_head."))
        elif name == "time_agg" and hasattr(model, "time_agg_layer"):
            model.time_agg_layer.load_state_dict(_sub(sd, "time_agg_layer."))
        elif name == "out" and hasattr(model, "out_layer"):
            model.out_layer.load_state_dict(_sub(sd, "out_layer."))
        elif name in COMPONENTS:
            # utils/utilities.py:163 prints 'Submodule does not exists' and carries on: fine-tune configs list e.g.
            # 'scale_feats' for models built with normalize=False
            warnings.warn(f"load_components_from_pretrained: this model has no {name!r} component - skipped")
        else:
            raise KeyError(f"unknown component {name!r} (known: {COMPONENTS})")


# ------------------------------------------------------------------------------------------------------
@torch.no_grad()
def rollout_eval(model: nn.Module, xx: Tensor, yy: Tensor, msk: Optional[Tensor], T_bundle: int = 1,
                 step=None) -> Tuple[Tensor, Tensor, Tensor]:
    """evaluate.py:193-213.  Returns (pred [B,X,Y,T_ar,C], sum of the per-step losses, loss of the whole rollout).
    `step(xx) -> im` defaults to the model's forward (a GraphedRollout passes its graph replay)."""
    # weight-only products (packed AFNO weights, folded embed matrices, ...) once per rollout, not once per AR step
    scope = model.weights_scope() if (step is None and hasattr(model, "weights_scope")) else contextlib.nullcontext()
    step = step or (lambda x: model(x)[0])
    T_ar = yy.shape[-2]
    loss_steps = None
    preds = []
    xx = xx.contiguous()
    with scope:
        for t in range(0, T_ar, T_bundle):
            y = yy[..., t:t + T_bundle, :]
            im = step(xx)
            l = rel_l2_loss(im, y.contiguous(), msk)
            loss_steps = l if loss_steps is None else loss_steps + l
            preds.append(im)
            if t + T_bundle < T_ar:
                xx = ops.window_slide(xx, im.contiguous())               # xx[..., T_bundle:, :] ++ im, one kernel
    pred = preds[0] if len(preds) == 1 else torch.cat(preds, dim=-2)
    loss_full = rel_l2_loss(pred.contiguous(), yy.contiguous(), msk)
    return pred, loss_steps, loss_full


class GraphedRollout:
    """One hipGraph of the forward step for a fixed input shape, replayed for every AR step."""

    def __init__(self, model: nn.Module, example_xx: Tensor):
        self.model = model
        self.x = example_xx.detach().clone().contiguous()
        was_training = model.training
        model.eval()
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    model(self.x)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.im, self.cls = model(self.x)
        model.train(was_training)

    def step(self, xx: Tensor) -> Tensor:
        self.x.copy_(xx)
        self.graph.replay()
        return self.im.clone()          # the graph's output buffer is overwritten by the next replay

    def __call__(self, xx: Tensor, yy: Tensor, msk: Optional[Tensor], T_bundle: int = 1):
        if tuple(xx.shape) != tuple(self.x.shape):
            raise ValueError(f"GraphedRollout captured for input {tuple(self.x.shape)}, got {tuple(xx.shape)}")
        return rollout_eval(self.model, xx, yy, msk, T_bundle, step=self.step)
