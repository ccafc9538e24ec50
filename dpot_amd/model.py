"""DPOTNet for MI355X.

Drop-in for the reference model class (models/dpot.py:245-403): identical constructor signature, identical
``state_dict`` keys / shapes / ordering (so pretrained ``.pth`` files load unchanged, also through
utils/utilities.py:99-166 style per-component loading) and the same call contract

    pred, cls_pred = model(x)          # x: [B, X, Y, T_in, C_in]  ->  [B, X, Y, T_out, C_out], [B, n_cls]

but the compute is the hand-written HIP path in libdpot_hip.so (functional.py / ops.py).  The torch modules
below (Conv2d, GroupNorm, Linear, ...) are used purely as *parameter containers* - they own the weights with the
reference's names and default initialisation; their own forward() is never called.

The model only runs on a CUDA (ROCm) device; calling it with CPU tensors raises - there is no fallback path.
"""
from __future__ import annotations

import numpy as np
import os

import torch
import torch.nn as nn

from . import _lib, ops
import contextlib

from .functional import (AdaINFn, BlockFn, EmbedFn, HeadFn, embed_derived, embed_grid_matrix, embed_layout_jobs,
                         head_derived, head_layout_jobs,
                         mlp_pack_kind)

ACTIVATIONS = ("gelu", "tanh", "sigmoid", "relu", "leaky_relu", "softplus", "ELU", "silu")


class _AFNOParams(nn.Module):
    """weights of the block-diagonal complex MLP shared by all Fourier modes (models/dpot.py:41-48)"""

    def __init__(self, width: int, num_blocks: int):
        super().__init__()
        assert width % num_blocks == 0, f"hidden_size {width} should be divisble by num_blocks {num_blocks}"
        bs = width // num_blocks
        scale = 1.0 / (bs * bs)
        self.w1 = nn.Parameter(scale * torch.rand(2, num_blocks, bs, bs))
        self.b1 = nn.Parameter(scale * torch.rand(2, num_blocks, bs))
        self.w2 = nn.Parameter(scale * torch.rand(2, num_blocks, bs, bs))
        self.b2 = nn.Parameter(scale * torch.rand(2, num_blocks, bs))


class _BlockParams(nn.Module):
    def __init__(self, width: int, n_blocks: int, mlp_ratio: float):
        super().__init__()
        self.norm1 = nn.GroupNorm(8, width)
        self.filter = _AFNOParams(width, n_blocks)
        self.norm2 = nn.GroupNorm(8, width)
        hidden = int(width * mlp_ratio)
        self.mlp = nn.Sequential(nn.Conv2d(width, hidden, 1), nn.Identity(), nn.Conv2d(hidden, width, 1))


class _PatchEmbedParams(nn.Module):
    def __init__(self, img_size: int, patch_size: int, in_chans: int, hidden: int, out_dim: int):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.out_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.out_size[0] * self.out_size[1]
        self.proj = nn.Sequential(nn.Conv2d(in_chans, hidden, patch_size, patch_size), nn.Identity(),
                                  nn.Conv2d(hidden, out_dim, 1))


class _TimeAggParams(nn.Module):
    def __init__(self, n_timesteps: int, width: int, kind: str):
        super().__init__()
        self.type = kind
        self.w = nn.Parameter(1.0 / (n_timesteps * width ** 0.5) * torch.randn(n_timesteps, width, width))
        if kind == "exp_mlp":
            self.gamma = nn.Parameter(2 ** torch.linspace(-10, 10, width).unsqueeze(0))
        elif kind != "mlp":
            raise ValueError(f"unknown time_agg {kind!r}")


class DPOTNet(nn.Module):
    def __init__(self, img_size=224, patch_size=16, mixing_type='afno', in_channels=1, out_channels=4, in_timesteps=1,
                 out_timesteps=1, n_blocks=4, embed_dim=768, out_layer_dim=32, depth=12, modes=32, mlp_ratio=1.,
                 n_cls=12, normalize=False, act='gelu', time_agg='exp_mlp'):
        super().__init__()
        if act not in ACTIVATIONS:
            raise KeyError(act)
        if mixing_type != 'afno':
            raise ValueError("only mixing_type='afno' exists in the reference")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.in_timesteps, self.out_timesteps = in_timesteps, out_timesteps
        self.n_blocks, self.modes = n_blocks, modes
        self.num_features = self.embed_dim = embed_dim
        self.mlp_ratio = mlp_ratio
        self.normalize, self.time_agg, self.n_cls = normalize, time_agg, n_cls
        self.mixing_type = mixing_type
        self.img_size, self.patch_size = img_size, patch_size
        self.act_name, self._act = act, ops.ACT_IDS[act]

        self.patch_embed = _PatchEmbedParams(img_size, patch_size, in_channels + 3, out_channels * patch_size + 3,
                                             embed_dim)
        self.latent_size = self.patch_embed.out_size
        h = self.latent_size[0]
        self.pos_embed = nn.Parameter(torch.zeros(1, embed_dim, h, h))
        self.blocks = nn.ModuleList([_BlockParams(embed_dim, n_blocks, mlp_ratio) for _ in range(depth)])
        if normalize:
            self.scale_feats_mu = nn.Linear(2 * in_channels, embed_dim)
            self.scale_feats_sigma = nn.Linear(2 * in_channels, embed_dim)
        self.cls_head = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.Identity(), nn.Linear(embed_dim, embed_dim),
                                      nn.Identity(), nn.Linear(embed_dim, n_cls))
        self.time_agg_layer = _TimeAggParams(in_timesteps, embed_dim, time_agg)
        self.out_layer = nn.Sequential(
            nn.ConvTranspose2d(embed_dim, out_layer_dim, patch_size, patch_size), nn.Identity(),
            nn.Conv2d(out_layer_dim, out_layer_dim, 1), nn.Identity(),
            nn.Conv2d(out_layer_dim, out_channels * out_timesteps, 1))
        torch.nn.init.trunc_normal_(self.pos_embed, std=.02)

        # coordinate tables of models/dpot.py:350-360 (np.linspace in float64, then cast) - not part of state_dict
        for name, n in (("_gx", img_size), ("_gy", img_size), ("_gt", in_timesteps)):
            self.register_buffer(name, torch.tensor(np.linspace(0, 1, n), dtype=torch.float32), persistent=False)
        self.register_buffer("_tt", torch.linspace(0, 1, in_timesteps), persistent=False)

        # activation recomputation inside every Block (functional.BlockFn): keep only block inputs between forward
        # and backward - for long auto-regressive rollouts at 256^2 (BASELINE configs[4])
        self.recompute_blocks = False
        # selective recomputation in an auto-regressive rollout (train.rollout tells the model which AR step a call is): the LAST
        # `recompute_keep_last` AR steps keep their activations like an ordinary forward - their backward runs first and frees them
        # before the first recomputation, so the peak stays at the end of the forward - and skip the second forward pass
        # (bit-identical either way; spends free HBM on time: DPOT-L, 20 steps, batch 16: 112 GB with 0 kept)
        self.recompute_keep_last = 0
        self._ar_pos = None                          # (index, count) of the current AR step, set by train.rollout
        # precision of the channel-MLP GEMMs of THIS model: None = the process default (ops.set_mlp_precision /
        # DPOT_MLP_PRECISION), or 'f32' | 'bf16x6' | 'auto' | 'bf16' (BASELINE configs[2]: "bf16 channel-MLP on MFMA")
        self.mlp_precision = None
        # precision of every OTHER fp32 GEMM of this model: None = the process default (ops.set_gemm_precision /
        # DPOT_GEMM_PRECISION), or 'f32' (native fp32 MFMA) | 'bf16x6' | 'auto' (the fp32-accurate operand split where faster)
        self.gemm_precision = None
        self._scope_depth = 0
        self._scope_cache = None
        # optional callable(b, lat) -> lat, called with the latent ENTERING stage b (1..depth = block b-1,
        # depth+1 = the head); train.SegmentedTrainStep cuts the autograd graph there
        self._boundary_hook = None

    # ------------------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def weights_scope(self):
        """Inside this scope the weight-only products of the forward (block-diagonal complex weights packed as real
        GEMM matrices, padded / transposed conv weights, the folded embed matrices) are computed by the FIRST forward
        call and re-used by the following ones: an auto-regressive rollout calls the model T_ar times per optimiser
        step on unchanged weights (train_temporal.py:201-219, evaluate.py:193-213).  The caller promises not to
        modify parameters inside the scope; outside a scope every forward derives them afresh."""
        self._scope_depth += 1
        dev = self.pos_embed.device
        if self._scope_depth == 1:
            self._scope_cache = None
            if dev.type == "cuda":
                with ops.precision_scope(self.gemm_precision, self.mlp_precision):
                    self._derived_weights()
        try:
            yield self
        finally:
            self._scope_depth -= 1
            if self._scope_depth == 0:
                self._scope_cache = None

    def _derived_weights(self):
        if self._scope_depth > 0 and self._scope_cache is not None:
            return self._scope_cache
        pe, ta, ol = self.patch_embed.proj, self.time_agg_layer, self.out_layer
        grid = None
        if ops.embed_supported(self.in_channels, self.patch_size, self.in_timesteps, pe[0].weight.shape[0],
                               self.img_size // self.patch_size):
            grid = getattr(self, "_embed_grid", None)
            if grid is None or grid.device != self._gx.device:
                grid = self._embed_grid = embed_grid_matrix(self._gx, self._gy, self._gt, self.img_size, self.img_size,
                                                            self.in_timesteps, self.in_channels, self.patch_size)
        # every small layout piece of the model (padded conv weights, pos_embed^T + bias, de-embed bias per pixel, padded
        # tail weights) in ONE launch from a device-resident job table with persistent outputs (ops.LayoutJobs)
        lay_e = lay_h = None
        if ops.tune("fused_small") != 0:
            je = embed_layout_jobs(self.pos_embed, pe[0].weight, pe[0].bias, pe[2].weight, pe[2].bias)
            jh = head_layout_jobs(ol[0].bias, ol[4].weight, ol[4].bias, self.patch_size, ol[0].weight.shape[1])
            lj = getattr(self, "_layout_jobs", None)
            if lj is None or lj.key != ops.LayoutJobs.key_of(je + jh):
                lj = self._layout_jobs = ops.LayoutJobs(je + jh)
            lay = lj.refresh()
            lay_e, lay_h = lay[:len(je)], lay[len(je):]
        emb = embed_derived(self.pos_embed, pe[0].weight, pe[0].bias, pe[2].weight, pe[2].bias, ta.w,
                            ta.gamma if self.time_agg == "exp_mlp" else None, self._tt, self.in_timesteps, grid=grid,
                            layouts=lay_e)
        # Wbig = [[Wr, Wi], [-Wi, Wr]] of every AFNO layer + its fragment-block-major forms, ONE launch for all layers
        pk = []
        if len(self.blocks):
            pairs = [p for blk in self.blocks for p in ((blk.filter.w1, blk.filter.b1), (blk.filter.w2, blk.filter.b2))]
            ap = getattr(self, "_afno_packs", None)
            if ap is None or ap.key != tuple(t.data_ptr() for p in pairs for t in p):
                ap = self._afno_packs = ops.AfnoPacks(pairs)
            pk = ap.refresh()
        PP_old = self.patch_size ** 2 * ol[0].weight.shape[1]
        wt_buf = getattr(self, "_wt_buf", None)
        if wt_buf is None or wt_buf.device != ol[0].weight.device or wt_buf.numel() != self.embed_dim * PP_old:
            wt_buf = self._wt_buf = torch.empty(self.embed_dim, PP_old, dtype=torch.float32, device=ol[0].weight.device)
        head = head_derived(ol[0].weight, ol[0].bias, ol[4].weight, ol[4].bias, self.patch_size, wt_out=wt_buf,
                            layouts=lay_h)
        mlp_pk, head_pk = self._panel_packs_refresh(wt_buf)
        d = (emb, pk, head + (head_pk,), mlp_pk)
        if self._scope_depth > 0:
            self._scope_cache = d
        return d

    def _panel_packs_refresh(self, wt):
        """fragment-block-major copies of the static weights of the panel GEMMs, refreshed by ONE launch each per
        optimiser step: per block W1, W1^T, W2, W2^T (channel-MLP forward x W^T and data gradient dy W) - fp32 for
        csrc/gemm_panel.hip, or bf16 for csrc/gemm_bf16p.hip when the channel-MLP precision is 'bf16' - and the de-embed
        matrix wt [E, P*P*old] both ways (fp32).  Returns (per-block tuples | None, (wt fwd, wt bwd) | None)."""
        E = self.embed_dim
        n_out = wt.shape[1]
        mlp_pk = head_pk = None
        nb = len(self.blocks)

        def mlp_jobs():
            jobs = []
            for b in self.blocks:
                w1, w2 = b.mlp[0].weight, b.mlp[2].weight                   # [mh, E, 1, 1], [E, mh, 1, 1]
                jobs += [(w1, mh, E, E, False), (w1, E, mh, E, True), (w2, E, mh, mh, False), (w2, mh, E, mh, True)]
            return jobs

        def cached(attr, key, make_jobs, bf16, planes=1):
            pp = getattr(self, attr, None)
            if pp is None or pp.key_all != key:
                pp = ops.PanelPacks(make_jobs(), bf16=bf16, planes=planes)
                pp.key_all = key
                setattr(self, attr, pp)
            pp.refresh()
            return pp

        wkey = tuple(b.mlp[i].weight.data_ptr() for b in self.blocks for i in (0, 2))
        kind = None
        if nb:
            mh = self.blocks[0].mlp[0].weight.shape[0]
            kind = mlp_pack_kind(E, mh)
            if kind in ("bf16", "bf16x6"):
                planes = 3 if kind == "bf16x6" else 1
                pp = cached("_panel_packs_" + kind, wkey, mlp_jobs, True, planes)
                mlp_pk = [ops.MlpPacks(pp.bufs[4 * i:4 * i + 4], kind) for i in range(nb)]
        f32_mlp = kind == "f32"
        use_head = (ops.panel_enabled() and ops.gemm_panel_supported(1, n_out, E)
                    and ops.gemm_panel_supported(1, E, n_out))
        if f32_mlp or use_head:
            def jobs32():
                jobs = mlp_jobs() if f32_mlp else []
                if use_head:
                    jobs += [(wt, n_out, E, n_out, True), (wt, E, n_out, n_out, False)]
                return jobs
            pp = cached("_panel_packs", wkey + (wt.data_ptr(), f32_mlp, use_head), jobs32, False)
            n0 = 4 * nb if f32_mlp else 0
            if f32_mlp:
                mlp_pk = [ops.MlpPacks(pp.bufs[4 * i:4 * i + 4], "f32") for i in range(nb)]
            if use_head:
                head_pk = tuple(pp.bufs[n0:n0 + 2])
        return mlp_pk, head_pk

    def forward(self, x):
        with ops.precision_scope(self.gemm_precision, self.mlp_precision):
            return self._forward(x)

    def _forward(self, x):
        if not x.is_cuda:
            raise _lib.DpotHipError("DPOTNet (dpot_amd) runs on MI355X only: move the model and the input to 'cuda'. "
                                    "There is no CPU fallback.")
        B, X, Y, T, Cin = x.shape
        assert X == self.img_size and Y == self.img_size, \
            f"Input image size ({X}*{Y}) doesn't match model ({self.img_size}*{self.img_size})."
        assert T == self.in_timesteps and Cin == self.in_channels, "input timesteps / channels mismatch"
        x = x.float()
        if self.normalize:
            # models/dpot.py:366-370 - per-sample statistics + two tiny Linear(2C -> E): O(B*C) host-side glue
            mu = x.mean(dim=(1, 2, 3), keepdim=True)
            sigma = x.std(dim=(1, 2, 3), keepdim=True) + 1e-6
            x = (x - mu) / sigma
            stat = torch.cat([mu, sigma], dim=-1)[:, 0, 0, 0, :]
            s_mu = self.scale_feats_mu(stat)
            s_sigma = self.scale_feats_sigma(stat)

        pe, ta = self.patch_embed.proj, self.time_agg_layer
        P = self.patch_size
        h = self.latent_size[0]
        d_emb, pk, d_head, mlp_pk = self._derived_weights()
        lat = EmbedFn.apply(x, self.pos_embed, pe[0].weight, pe[0].bias, pe[2].weight, pe[2].bias, ta.w,
                            ta.gamma if self.time_agg == "exp_mlp" else None, self._gx, self._gy, self._gt, self._tt,
                            P, self._act, d_emb)
        if self.normalize:
            lat = AdaINFn.apply(lat, s_sigma, s_mu)                     # AdaIN (models/dpot.py:386-387)
        recompute = self.recompute_blocks and torch.is_grad_enabled()
        if recompute and self.recompute_keep_last > 0 and self._ar_pos is not None \
                and self._ar_pos[0] >= self._ar_pos[1] - self.recompute_keep_last:
            recompute = False
        hook = self._boundary_hook
        for i, blk in enumerate(self.blocks):
            cut = False
            if hook is not None:
                lat0 = lat
                lat = hook(i + 1, lat)
                cut = lat is not lat0          # the autograd graph is cut here (train.SegmentedTrainStep): this Block's input
            f = blk.filter                     # gradient goes to a leaf, not to the previous Block - no gradient packs
            lat = BlockFn.apply(lat, blk.norm1.weight, blk.norm1.bias, f.w1, f.b1, f.w2, f.b2, blk.norm2.weight,
                                blk.norm2.bias, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight,
                                blk.mlp[2].bias, h, h, self.n_blocks, self.modes, self._act,
                                (pk[2 * i], pk[2 * i + 1]), recompute, mlp_pk[i] if mlp_pk is not None else None,
                                torch.is_grad_enabled(), i > 0 and not cut)
        if hook is not None:
            lat = hook(len(self.blocks) + 1, lat)
        ol, ch = self.out_layer, self.cls_head
        pred, cls_pred = HeadFn.apply(lat, ol[0].weight, ol[0].bias, ol[2].weight, ol[2].bias, ol[4].weight,
                                      ol[4].bias, ch[0].weight, ch[0].bias, ch[2].weight, ch[2].bias, ch[4].weight,
                                      ch[4].bias, h, h, P, self._act, d_head)
        pred = pred.view(B, X, Y, self.out_timesteps, self.out_channels)
        if self.normalize:
            pred = pred * sigma + mu
        return pred, cls_pred

    def extra_repr(self) -> str:
        return (f"img_size={self.img_size}, patch_size={self.patch_size}, embed_dim={self.embed_dim}, "
                f"depth={len(self.blocks)}, n_blocks={self.n_blocks}, modes={self.modes}, act={self.act_name}, "
                f"backend=libdpot_hip(gfx950)")
