"""CPU oracle for the DPOT auto-regressive forward/backward step.  TEST INFRASTRUCTURE ONLY.

This is a plain-PyTorch (CPU, fp32 / complex64) *restatement* of the reference algorithm, written
functionally over a ``state_dict`` so that it shares no structure with the reference modules.  It is the
checker for the HIP path (``tests/``), the smoke check (``__graft_entry__.smoke``) and the reported CPU
baseline (``bench.py: cpu_baseline``).  Nothing under ``dpot_amd/`` may import it.

Parity pinning: the reference has no tests or golden vectors of its own, so this
oracle is pinned against the *imported reference itself* in this container
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).

Reference citations (relative to /root/reference):
  * model forward ............ models/dpot.py:364-403
  * AFNO2D spectral mixer .... models/dpot.py:51-110   (soft-shrink is commented out there, :97-98)
  * Block .................... models/dpot.py:165-180  (GroupNorm(8, width), double_skip=False at :294)
  * PatchEmbed ............... models/dpot.py:198-209
  * TimeAggregator ........... models/dpot.py:226-234
  * grid ..................... models/dpot.py:350-360
  * SimpleLpLoss ............. utils/criterion.py:38-59 (live branch :59)
  * Adam ..................... utils/optimizer.py:9-52
  * rollout loop ............. train_temporal.py:189-230
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------------
# configuration (mirrors the keyword arguments of the reference constructor, models/dpot.py:246-247)
# --------------------------------------------------------------------------------------------------
@dataclass
class DPOTConfig:
    img_size: int = 224
    patch_size: int = 16
    mixing_type: str = "afno"
    in_channels: int = 1
    out_channels: int = 4
    in_timesteps: int = 1
    out_timesteps: int = 1
    n_blocks: int = 4
    embed_dim: int = 768
    out_layer_dim: int = 32
    depth: int = 12
    modes: int = 32
    mlp_ratio: float = 1.0
    n_cls: int = 12
    normalize: bool = False
    act: str = "gelu"
    time_agg: str = "exp_mlp"

    @property
    def latent(self) -> int:
        return self.img_size // self.patch_size

    @property
    def patch_hidden(self) -> int:
        # models/dpot.py:278  embed_dim of the first patch conv = out_channels * patch_size + 3
        return self.out_channels * self.patch_size + 3

    @property
    def mlp_hidden(self) -> int:
        return int(self.embed_dim * self.mlp_ratio)


TINY = dict(img_size=128, patch_size=8, in_channels=4, out_channels=4, in_timesteps=10, out_timesteps=1,
            n_blocks=4, embed_dim=512, out_layer_dim=32, depth=4, modes=32, mlp_ratio=1, n_cls=12)
SMALL = dict(TINY, embed_dim=1024, depth=6, n_blocks=8)
MEDIUM = dict(TINY, embed_dim=1024, depth=12, n_blocks=8, mlp_ratio=4)
LARGE = dict(TINY, img_size=256, embed_dim=1536, depth=24, n_blocks=16, mlp_ratio=4, out_layer_dim=128,
             modes=64)
MINI = dict(img_size=32, patch_size=8, in_channels=3, out_channels=3, in_timesteps=4, out_timesteps=1,
            n_blocks=4, embed_dim=64, out_layer_dim=16, depth=2, modes=32, mlp_ratio=1, n_cls=5)


# constructor variants pinned by tests/golden/g10_*.npz (oracle/make_golden.py::g10_variant)
GOLDEN_VARIANTS = {
    "g10_bundle": dict(MINI, out_timesteps=2, out_channels=2, in_channels=3),
    "g10_mlpagg": dict(MINI, time_agg="mlp", n_blocks=2, mlp_ratio=2),
    "g10_leaky_modes1": dict(MINI, act="leaky_relu", modes=1, out_layer_dim=16, n_cls=5),
}


def _act(name: str):
    table = {
        "gelu": lambda v: F.gelu(v),                    # exact erf GELU (nn.GELU() default)
        "tanh": torch.tanh,
        "sigmoid": torch.sigmoid,
        "relu": torch.relu,
        "leaky_relu": lambda v: F.leaky_relu(v, 0.1),
        "softplus": F.softplus,
        "ELU": F.elu,
        "silu": F.silu,
    }
    return table[name]


# --------------------------------------------------------------------------------------------------
# parameter shapes / deterministic recipe weights
# --------------------------------------------------------------------------------------------------
def param_shapes(cfg: DPOTConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict keys and shapes in the reference's registration order."""
    E, P, h = cfg.embed_dim, cfg.patch_size, cfg.latent
    nb, bs = cfg.n_blocks, cfg.embed_dim // cfg.n_blocks
    hid, mh, old = cfg.patch_hidden, cfg.mlp_hidden, cfg.out_layer_dim
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["pos_embed"] = (1, E, h, h)
    s["patch_embed.proj.0.weight"] = (hid, cfg.in_channels + 3, P, P)
    s["patch_embed.proj.0.bias"] = (hid,)
    s["patch_embed.proj.2.weight"] = (E, hid, 1, 1)
    s["patch_embed.proj.2.bias"] = (E,)
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (E,)
        s[p + "norm1.bias"] = (E,)
        s[p + "filter.w1"] = (2, nb, bs, bs)
        s[p + "filter.b1"] = (2, nb, bs)
        s[p + "filter.w2"] = (2, nb, bs, bs)
        s[p + "filter.b2"] = (2, nb, bs)
        s[p + "norm2.weight"] = (E,)
        s[p + "norm2.bias"] = (E,)
        s[p + "mlp.0.weight"] = (mh, E, 1, 1)
        s[p + "mlp.0.bias"] = (mh,)
        s[p + "mlp.2.weight"] = (E, mh, 1, 1)
        s[p + "mlp.2.bias"] = (E,)
    if cfg.normalize:
        s["scale_feats_mu.weight"] = (E, 2 * cfg.in_channels)
        s["scale_feats_mu.bias"] = (E,)
        s["scale_feats_sigma.weight"] = (E, 2 * cfg.in_channels)
        s["scale_feats_sigma.bias"] = (E,)
    s["cls_head.0.weight"] = (E, E)
    s["cls_head.0.bias"] = (E,)
    s["cls_head.2.weight"] = (E, E)
    s["cls_head.2.bias"] = (E,)
    s["cls_head.4.weight"] = (cfg.n_cls, E)
    s["cls_head.4.bias"] = (cfg.n_cls,)
    s["time_agg_layer.w"] = (cfg.in_timesteps, E, E)
    if cfg.time_agg == "exp_mlp":
        s["time_agg_layer.gamma"] = (1, E)
    s["out_layer.0.weight"] = (E, old, P, P)
    s["out_layer.0.bias"] = (old,)
    s["out_layer.2.weight"] = (old, old, 1, 1)
    s["out_layer.2.bias"] = (old,)
    s["out_layer.4.weight"] = (cfg.out_channels * cfg.out_timesteps, old, 1, 1)
    s["out_layer.4.bias"] = (cfg.out_channels * cfg.out_timesteps,)
    return s


def _fan_in(name: str, shape: Tuple[int, ...]) -> int:
    if name.startswith("out_layer.0.weight"):          # ConvTranspose2d: [in, out, kh, kw]
        return shape[0]
    if len(shape) >= 2:
        return int(np.prod(shape[1:]))
    return 1


def recipe_tensor(name: str, shape: Tuple[int, ...], salt: int = 0) -> Tensor:
    """Closed-form pseudo-random tensor (a sine hash in float64): identical wherever it is evaluated,
    so the GPU box regenerates the very same weights/inputs without the reference being present."""
    n = int(np.prod(shape))
    key = (sum((i + 1) * ord(ch) for i, ch in enumerate(name)) % 9973) + 17 * salt
    idx = np.arange(n, dtype=np.float64)
    u = np.sin(idx * 12.9898 + key * 78.233 + 0.5) * 43758.5453
    u = u - np.floor(u)                                    # uniform-ish in [0, 1)
    return torch.from_numpy(u.reshape(shape))


def recipe_state_dict(cfg: DPOTConfig, salt: int = 0) -> "OrderedDict[str, Tensor]":
    """Deterministic weights with reference-like magnitudes (not the reference's RNG init)."""
    sd: "OrderedDict[str, Tensor]" = OrderedDict()
    bs = cfg.embed_dim // cfg.n_blocks
    for name, shape in param_shapes(cfg).items():
        u = recipe_tensor(name, shape, salt)
        if name == "pos_embed":
            v = (u - 0.5) * 0.08
        elif ".filter." in name:
            # reference: scale * rand with scale = 1/bs^2 (models/dpot.py:41-48); that makes the spectral
            # branch numerically invisible, so the recipe uses a larger, sign-symmetric scale
            v = (u - 0.5) * (2.0 / math.sqrt(bs))
        elif name.endswith("norm1.weight") or name.endswith("norm2.weight"):
            v = 0.75 + 0.5 * u
        elif name.endswith("norm1.bias") or name.endswith("norm2.bias"):
            v = (u - 0.5) * 0.2
        elif name == "time_agg_layer.gamma":
            E = cfg.embed_dim
            v = (2.0 ** torch.linspace(-10, 10, E, dtype=torch.float64)).unsqueeze(0) * (0.9 + 0.2 * u)
        elif name == "time_agg_layer.w":
            v = (u - 0.5) * 2.0 * math.sqrt(3.0) / (cfg.in_timesteps * math.sqrt(cfg.embed_dim)) * 3.0
        elif name.endswith(".bias"):
            v = (u - 0.5) * 0.1
        else:
            v = (u - 0.5) * 2.0 * math.sqrt(3.0 / _fan_in(name, shape))
        sd[name] = v.to(torch.float32).contiguous()
    return sd


def recipe_input(shape: Tuple[int, ...], salt: int = 101) -> Tensor:
    return ((recipe_tensor("input", shape, salt) - 0.5) * 3.0).to(torch.float32)


# --------------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------------
def unit_grid(n: int) -> Tensor:
    # models/dpot.py:352  torch.tensor(np.linspace(0, 1, n), dtype=torch.float)
    return torch.tensor(np.linspace(0, 1, n), dtype=torch.float32)


def append_grid(x: Tensor) -> Tensor:
    """[B,X,Y,T,C] -> [B,X,Y,T,C+3] with (x, y, t) coordinates in [0,1].  models/dpot.py:350-360,374."""
    B, X, Y, T, _ = x.shape
    gx = unit_grid(X).view(1, X, 1, 1, 1).expand(B, X, Y, T, 1)
    gy = unit_grid(Y).view(1, 1, Y, 1, 1).expand(B, X, Y, T, 1)
    gt = unit_grid(T).view(1, 1, 1, T, 1).expand(B, X, Y, T, 1)
    return torch.cat([x, gx.to(x), gy.to(x), gt.to(x)], dim=-1)


def patchify(x: Tensor, P: int) -> Tensor:
    """[B,X,Y,T,C] -> [B,h,w,T, C*P*P] with the last axis ordered (c, i, j) like a conv weight."""
    B, X, Y, T, C = x.shape
    h, w = X // P, Y // P
    v = x.view(B, h, P, w, P, T, C).permute(0, 1, 3, 5, 6, 2, 4)       # b h w t c i j
    return v.reshape(B, h, w, T, C * P * P)


def patch_embed(sd: Dict[str, Tensor], x: Tensor, cfg: DPOTConfig) -> Tensor:
    """Strided 8x8 conv == per-patch matmul; -> act -> 1x1 conv; + pos_embed.  Returns [B,h,w,T,E].
    models/dpot.py:198-202,375-380."""
    act = _act(cfg.act)
    a = patchify(append_grid(x), cfg.patch_size)                            # [B,h,w,T,K]
    w0 = sd["patch_embed.proj.0.weight"].reshape(cfg.patch_hidden, -1)      # [hid, K]
    hmid = act(a @ w0.t() + sd["patch_embed.proj.0.bias"])
    w2 = sd["patch_embed.proj.2.weight"].reshape(cfg.embed_dim, cfg.patch_hidden)
    z = hmid @ w2.t() + sd["patch_embed.proj.2.bias"]                        # [B,h,w,T,E]
    pos = sd["pos_embed"][0].permute(1, 2, 0)                                # [h,w,E]
    return z + pos[None, :, :, None, :]


def time_aggregate(sd: Dict[str, Tensor], z: Tensor, cfg: DPOTConfig) -> Tensor:
    """[B,h,w,T,E] -> [B,h,w,E].  models/dpot.py:226-234."""
    w = sd["time_agg_layer.w"]
    if cfg.time_agg == "mlp":
        return torch.einsum("tij,...ti->...j", w, z)
    T = z.shape[-2]
    t = torch.linspace(0, 1, T).unsqueeze(-1)                               # [T,1]
    gamma = sd["time_agg_layer.gamma"]
    if gamma.dtype == torch.float32:
        t_embed = torch.cos(t @ gamma)                                       # [T,E]
    else:
        # float64 diagnostic runs (scripts/diag_grad64.py): same fp32 linspace, fp64 phase
        t_embed = torch.cos(t.to(gamma.dtype) @ gamma)
    return torch.einsum("tij,...ti->...j", w, z * t_embed)


def group_norm_cl(x: Tensor, weight: Tensor, bias: Tensor, groups: int = 8, eps: float = 1e-5) -> Tensor:
    """GroupNorm over a channels-last tensor [B,h,w,E] (statistics over h,w and E/groups channels)."""
    B, h, w, E = x.shape
    v = x.reshape(B, h * w, groups, E // groups)
    mu = v.mean(dim=(1, 3), keepdim=True)
    var = v.var(dim=(1, 3), unbiased=False, keepdim=True)
    v = (v - mu) * torch.rsqrt(var + eps)
    return v.reshape(B, h, w, E) * weight + bias


def afno_mix(sd: Dict[str, Tensor], prefix: str, x: Tensor, cfg: DPOTConfig) -> Tensor:
    """x:[B,h,w,E] (already normalised) -> irfft2(MLP(rfft2(x))) + x.  models/dpot.py:51-110.

    The block-diagonal 2-layer complex MLP is shared by every Fourier mode; GELU acts separately on the
    real and imaginary parts; only modes [:m,:m] of the half spectrum are processed, the rest is zero."""
    act = _act(cfg.act)
    B, h, w, E = x.shape
    nb = cfg.n_blocks
    bs = E // nb
    w1 = torch.complex(sd[prefix + "w1"][0], sd[prefix + "w1"][1])           # [nb,bs,bs]
    b1 = torch.complex(sd[prefix + "b1"][0], sd[prefix + "b1"][1])           # [nb,bs]
    w2 = torch.complex(sd[prefix + "w2"][0], sd[prefix + "w2"][1])
    b2 = torch.complex(sd[prefix + "b2"][0], sd[prefix + "b2"][1])
    spec = torch.fft.rfft2(x, dim=(1, 2), norm="ortho")                      # [B,h,w/2+1,E]
    wf = spec.shape[2]
    mx, my = min(cfg.modes, h), min(cfg.modes, wf)
    s = spec[:, :mx, :my].reshape(B, mx, my, nb, bs)
    o1 = torch.einsum("...ki,kio->...ko", s, w1) + b1
    o1 = torch.complex(act(o1.real), act(o1.imag))
    o2 = torch.einsum("...ki,kio->...ko", o1, w2) + b2
    full = torch.zeros(B, h, wf, nb, bs, dtype=spec.dtype)
    full[:, :mx, :my] = o2
    y = torch.fft.irfft2(full.reshape(B, h, wf, E), s=(h, w), dim=(1, 2), norm="ortho")
    return y + x


def block_forward(sd: Dict[str, Tensor], i: int, x: Tensor, cfg: DPOTConfig) -> Tensor:
    """x:[B,h,w,E] channels-last.  models/dpot.py:165-180 with double_skip=False."""
    act = _act(cfg.act)
    p = f"blocks.{i}."
    E, mh = cfg.embed_dim, cfg.mlp_hidden
    v = group_norm_cl(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    v = afno_mix(sd, p + "filter.", v, cfg)
    v = group_norm_cl(v, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    v = act(v @ sd[p + "mlp.0.weight"].reshape(mh, E).t() + sd[p + "mlp.0.bias"])
    v = v @ sd[p + "mlp.2.weight"].reshape(E, mh).t() + sd[p + "mlp.2.bias"]
    return v + x


def cls_head(sd: Dict[str, Tensor], x: Tensor, cfg: DPOTConfig) -> Tensor:
    """x:[B,h,w,E] -> [B,n_cls].  models/dpot.py:303-309,394-395."""
    act = _act(cfg.act)
    v = x.mean(dim=(1, 2))
    v = act(v @ sd["cls_head.0.weight"].t() + sd["cls_head.0.bias"])
    v = act(v @ sd["cls_head.2.weight"].t() + sd["cls_head.2.bias"])
    return v @ sd["cls_head.4.weight"].t() + sd["cls_head.4.bias"]


def out_layer(sd: Dict[str, Tensor], x: Tensor, cfg: DPOTConfig) -> Tensor:
    """x:[B,h,w,E] -> [B,X,Y,T_out,C_out].  ConvTranspose(k=s=P) == per-token matmul + pixel shuffle;
    then two per-pixel 1x1 convs.  models/dpot.py:315-321,397-398."""
    act = _act(cfg.act)
    B, h, w, E = x.shape
    P, old = cfg.patch_size, cfg.out_layer_dim
    wt = sd["out_layer.0.weight"].reshape(E, old * P * P)                     # [E, (o,i,j)]
    u = (x @ wt).view(B, h, w, old, P, P) + sd["out_layer.0.bias"].view(1, 1, 1, old, 1, 1)
    u = u.permute(0, 1, 4, 2, 5, 3).reshape(B, h * P, w * P, old)            # b (h i) (w j) o
    u = act(u)
    u = act(u @ sd["out_layer.2.weight"].reshape(old, old).t() + sd["out_layer.2.bias"])
    co = cfg.out_channels * cfg.out_timesteps
    u = u @ sd["out_layer.4.weight"].reshape(co, old).t() + sd["out_layer.4.bias"]
    return u.reshape(B, h * P, w * P, cfg.out_timesteps, cfg.out_channels)


def dpot_forward(sd: Dict[str, Tensor], x: Tensor, cfg: DPOTConfig) -> Tuple[Tensor, Tensor]:
    """x:[B,X,Y,T_in,C_in] -> (pred [B,X,Y,T_out,C_out], cls_pred [B,n_cls]).  models/dpot.py:364-403."""
    assert x.shape[1] == cfg.img_size and x.shape[2] == cfg.img_size, \
        f"Input image size ({x.shape[1]}*{x.shape[2]}) doesn't match model ({cfg.img_size}*{cfg.img_size})."
    if cfg.normalize:
        mu = x.mean(dim=(1, 2, 3), keepdim=True)
        sigma = x.std(dim=(1, 2, 3), keepdim=True) + 1e-6
        x = (x - mu) / sigma
        stat = torch.cat([mu, sigma], dim=-1)[:, 0, 0, 0, :]                  # [B,2C]
        s_mu = stat @ sd["scale_feats_mu.weight"].t() + sd["scale_feats_mu.bias"]
        s_sg = stat @ sd["scale_feats_sigma.weight"].t() + sd["scale_feats_sigma.bias"]
    z = patch_embed(sd, x, cfg)
    v = time_aggregate(sd, z, cfg)                                            # [B,h,w,E]
    if cfg.normalize:
        v = s_sg[:, None, None, :] * v + s_mu[:, None, None, :]
    for i in range(cfg.depth):
        v = block_forward(sd, i, v, cfg)
    cls_pred = cls_head(sd, v, cfg)
    y = out_layer(sd, v, cfg)
    if cfg.normalize:
        y = y * sigma + mu
    return y, cls_pred


# --------------------------------------------------------------------------------------------------
# loss / optimiser / rollout step
# --------------------------------------------------------------------------------------------------
def rel_l2_loss(x: Tensor, y: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    """Masked relative L2, summed over the batch (SimpleLpLoss(size_average=False)).
    utils/criterion.py:38-59."""
    B, C = x.shape[0], x.shape[-1]
    if mask is not None:
        x = x * mask
        y = y * mask
        n_ch = mask.sum(dim=list(range(1, mask.ndim - 1))).count_nonzero(dim=-1)   # [B]
    else:
        n_ch = C
    d = torch.linalg.vector_norm(x.reshape(B, -1, C) - y.reshape(B, -1, C), ord=2, dim=1)
    yn = torch.linalg.vector_norm(y.reshape(B, -1, C), ord=2, dim=1) + 1e-8
    return ((d / yn).sum(dim=-1) / n_ch).sum()


def grad_global_norm(grads: List[Tensor]) -> Tensor:
    return torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()


def clip_coef(total_norm: Tensor, max_norm: float) -> Tensor:
    # torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    return torch.clamp(max_norm / (total_norm + 1e-6), max=1.0)


def adam_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float, beta2: float,
                eps: float, weight_decay: float) -> None:
    """One in-place Adam step with L2 weight decay folded into the gradient.  utils/optimizer.py:26-52."""
    if weight_decay != 0:
        g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


@dataclass
class TrainState:
    params: "OrderedDict[str, Tensor]"
    exp_avg: Dict[str, Tensor] = field(default_factory=dict)
    exp_avg_sq: Dict[str, Tensor] = field(default_factory=dict)
    step: int = 0


def rollout_loss(sd: Dict[str, Tensor], xx: Tensor, yy: Tensor, msk: Tensor, cfg: DPOTConfig,
                 T_bundle: int = 1, noise_scale: float = 0.0,
                 noise: Optional[List[Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """Auto-regressive rollout (train_temporal.py:201-219).  Returns (summed loss, pred [B,X,Y,T_ar,C])."""
    loss = 0.0
    preds = []
    for k, t in enumerate(range(0, yy.shape[-2], T_bundle)):
        y = yy[..., t:t + T_bundle, :]
        if noise_scale != 0.0:
            eps = noise[k] if noise is not None else torch.randn_like(xx)
            xx = xx + noise_scale * torch.sum(xx ** 2, dim=(1, 2, 3), keepdim=True) ** 0.5 * eps
        im, _ = dpot_forward(sd, xx, cfg)
        loss = loss + rel_l2_loss(im, y, msk)
        preds.append(im)
        xx = torch.cat((xx[..., T_bundle:, :], im), dim=-2)
    return loss, torch.cat(preds, dim=-2)


def train_step(state: TrainState, xx: Tensor, yy: Tensor, msk: Tensor, cfg: DPOTConfig, lr: float,
               betas=(0.9, 0.9), eps: float = 1e-8, weight_decay: float = 1e-6, grad_clip: float = 10000.0,
               T_bundle: int = 1, noise_scale: float = 0.0, grad_scale: float = 1.0) -> Dict[str, Tensor]:
    """forward rollout -> loss -> backward -> clip -> Adam.  train_temporal.py:201-230.

    ``grad_scale`` models the data-parallel average (grads divided by world size after the sum)."""
    params = state.params
    for p in params.values():
        p.requires_grad_(True)
        p.grad = None
    loss, pred = rollout_loss(params, xx, yy, msk, cfg, T_bundle, noise_scale)
    loss.backward()
    names = [n for n, p in params.items() if p.grad is not None]
    grads = {n: params[n].grad.detach() * grad_scale for n in names}
    gnorm = grad_global_norm(list(grads.values()))
    coef = clip_coef(gnorm, grad_clip)
    state.step += 1
    with torch.no_grad():
        for n in names:
            p = params[n]
            if n not in state.exp_avg:
                state.exp_avg[n] = torch.zeros_like(p)
                state.exp_avg_sq[n] = torch.zeros_like(p)
            adam_update(p, grads[n] * coef, state.exp_avg[n], state.exp_avg_sq[n], state.step, lr,
                        betas[0], betas[1], eps, weight_decay)
    return {"loss": loss.detach(), "grad_norm": gnorm, "pred": pred.detach(), "grads": grads}
