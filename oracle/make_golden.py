#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REFERENCE implementation.  TEST INFRASTRUCTURE ONLY.

Runs only in the build container, where /root/reference exists (it never travels to the GPU box).
It imports the reference modules as they are (models/dpot.py, utils/criterion.py, utils/optimizer.py),
loads *recipe* weights (oracle.dpot_ref.recipe_state_dict - closed form, so the GPU box can rebuild the
identical weights and inputs without the reference), runs them on recipe inputs and stores the outputs
and gradients.  While doing so it also cross-checks the oracle restatement against the reference and
refuses to write fixtures if they disagree.

    python oracle/make_golden.py            # writes tests/golden/*.npz

Fixture index:
  g1_afno_trunc      AFNO2D fwd + all grads, modes=5 (truncation active)      models/dpot.py:51-110
  g1_afno_tiny       AFNO2D Tiny layer (E512, nb4) subsampled + checksums
  g2_block           GroupNorm + Block fwd/bwd                                 models/dpot.py:137-180
  g3_embed           PatchEmbed(+grid,+pos) -> TimeAggregator fwd/bwd          models/dpot.py:183-234
  g3_out             out_layer + cls_head fwd/bwd                              models/dpot.py:303-321
  g4_mini            full mini DPOTNet: outputs + every gradient               models/dpot.py:364-403
  g4_mini_norm       same with normalize=True
  g5_tiny            DPOT-Tiny whole-model forward, B=2 (subsample + checksums)
  g6_rollout         3-step AR rollout train step (loss, grad norms, Adam)     train_temporal.py:189-230
  g7_loss            SimpleLpLoss with a partial mask                          utils/criterion.py:38-59
  g8_dp              2-rank data-parallel equivalence numbers                  train_temporal_parallel.py:243-244
  g10_*              constructor variants (temporal bundling out_timesteps=2, time_agg='mlp', other activation /
                     mlp_ratio / block count / 1 kept mode): outputs + subsampled gradients   models/dpot.py:246-326
"""
from __future__ import annotations

import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("DPOT_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import dpot_ref as R  # noqa: E402

from models.dpot import AFNO2D, Block, DPOTNet  # noqa: E402  (reference)
from utils.criterion import SimpleLpLoss  # noqa: E402        (reference)
from utils.optimizer import Adam  # noqa: E402                 (reference)

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)
torch.manual_seed(0)


def npy(t):
    return t.detach().cpu().numpy()


def sub(t, stride):
    """deterministic subsample + checksums of a big tensor"""
    f = t.detach().reshape(-1)
    return {"sub": npy(f[::stride]), "sum": np.float64(f.double().sum().item()),
            "abssum": np.float64(f.double().abs().sum().item()), "stride": np.int64(stride)}


def check(name, a, b, rtol=2e-5, atol=None):
    a, b = a.detach().double(), b.detach().double()
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    tol = (atol if atol is not None else rtol * scale)
    status = "ok" if err <= tol else "MISMATCH"
    print(f"  oracle-vs-reference {name:34s} max|d|={err:.3e}  scale={scale:.3e}  {status}")
    if err > tol:
        raise SystemExit(f"oracle disagrees with the reference on {name}")


def save(name, **arrays):
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}.{kk}"] = vv
        else:
            flat[k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def ref_model(cfg: R.DPOTConfig, sd):
    m = DPOTNet(img_size=cfg.img_size, patch_size=cfg.patch_size, mixing_type=cfg.mixing_type,
                in_channels=cfg.in_channels, out_channels=cfg.out_channels, in_timesteps=cfg.in_timesteps,
                out_timesteps=cfg.out_timesteps, n_blocks=cfg.n_blocks, embed_dim=cfg.embed_dim,
                out_layer_dim=cfg.out_layer_dim, depth=cfg.depth, modes=cfg.modes, mlp_ratio=cfg.mlp_ratio,
                n_cls=cfg.n_cls, normalize=cfg.normalize, act=cfg.act, time_agg=cfg.time_agg)
    ref_keys = list(m.state_dict().keys())
    assert ref_keys == list(sd.keys()), "oracle param_shapes() key order differs from the reference"
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
    m.load_state_dict(sd)
    return m


def leaf_sd(sd):
    return OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())


# --------------------------------------------------------------------------------------------------
def g1_afno(name, B, h, E, nb, modes, stride):
    cfg = R.DPOTConfig(img_size=h * 8, patch_size=8, embed_dim=E, n_blocks=nb, modes=modes, depth=1)
    sd_full = R.recipe_state_dict(cfg, salt=3)
    pre = "blocks.0.filter."
    x = R.recipe_input((B, h, h, E), salt=11)
    up = R.recipe_input((B, h, h, E), salt=12) * 0.3
    ref = AFNO2D(width=E, num_blocks=nb, channel_first=True, modes=modes)
    ref.load_state_dict({k[len(pre):]: v for k, v in sd_full.items() if k.startswith(pre)})
    xr = x.permute(0, 3, 1, 2).contiguous().requires_grad_(True)              # reference is NCHW here
    yr = ref(xr)
    (yr * up.permute(0, 3, 1, 2)).sum().backward()
    # oracle
    sd = leaf_sd({k: v for k, v in sd_full.items() if k.startswith(pre)})
    xo = x.clone().requires_grad_(True)
    yo = R.afno_mix(sd, pre, xo, cfg)
    (yo * up).sum().backward()
    check(name + ".y", yo, yr.permute(0, 2, 3, 1))
    check(name + ".dx", xo.grad, xr.grad.permute(0, 2, 3, 1))
    out = {}
    for k in ("w1", "b1", "w2", "b2"):
        check(name + ".d" + k, sd[pre + k].grad, getattr(ref, k).grad)
        out["d" + k] = npy(getattr(ref, k).grad) if stride == 1 else sub(getattr(ref, k).grad, stride)
    y_cl = yr.permute(0, 2, 3, 1).contiguous()
    dx_cl = xr.grad.permute(0, 2, 3, 1).contiguous()
    save(name, B=np.int64(B), h=np.int64(h), E=np.int64(E), nb=np.int64(nb), modes=np.int64(modes),
         y=npy(y_cl) if stride == 1 else sub(y_cl, stride),
         dx=npy(dx_cl) if stride == 1 else sub(dx_cl, stride), **out)


def g2_block():
    B, h, E, nb = 2, 8, 64, 4
    cfg = R.DPOTConfig(img_size=h * 8, patch_size=8, embed_dim=E, n_blocks=nb, modes=32, depth=1, mlp_ratio=2)
    sd_full = R.recipe_state_dict(cfg, salt=5)
    pre = "blocks.0."
    x = R.recipe_input((B, h, h, E), salt=21)
    up = R.recipe_input((B, h, h, E), salt=22) * 0.3
    ref = Block(mixing_type="afno", double_skip=False, width=E, n_blocks=nb, mlp_ratio=2, channel_first=True,
                modes=32, act="gelu")
    ref.load_state_dict({k[len(pre):]: v for k, v in sd_full.items() if k.startswith(pre)})
    xr = x.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    gn = ref.norm1(xr)                                                         # GroupNorm alone
    yr = ref(xr)
    (yr * up.permute(0, 3, 1, 2)).sum().backward()
    sd = leaf_sd({k: v for k, v in sd_full.items() if k.startswith(pre)})
    xo = x.clone().requires_grad_(True)
    gno = R.group_norm_cl(xo, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    yo = R.block_forward(sd, 0, xo, cfg)
    (yo * up).sum().backward()
    check("g2.gn", gno, gn.permute(0, 2, 3, 1))
    check("g2.y", yo, yr.permute(0, 2, 3, 1))
    check("g2.dx", xo.grad, xr.grad.permute(0, 2, 3, 1))
    grads = {}
    for k, p in ref.named_parameters():
        check("g2.d" + k, sd[pre + k].grad, p.grad)
        grads["d." + k] = npy(p.grad)
    save("g2_block", B=np.int64(B), h=np.int64(h), E=np.int64(E), nb=np.int64(nb),
         gn=npy(gn.permute(0, 2, 3, 1)), y=npy(yr.permute(0, 2, 3, 1)), dx=npy(xr.grad.permute(0, 2, 3, 1)),
         **grads)


def g3_embed_out():
    cfg = R.DPOTConfig(**dict(R.MINI, depth=1))
    sd0 = R.recipe_state_dict(cfg, salt=7)
    m = ref_model(cfg, sd0)
    B = 2
    x = R.recipe_input((B, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=31)
    # --- embed: literal re-statement of the first lines of the reference forward on reference modules
    from einops import rearrange
    xr = x.clone().requires_grad_(True)
    v = torch.cat((xr, m.get_grid_3d(xr)), dim=-1).contiguous()
    v = rearrange(v, "b x y t c -> (b t) c x y")
    v = m.patch_embed(v) + m.pos_embed
    z_ref = rearrange(v, "(b t) c x y -> b x y t c", b=B, t=cfg.in_timesteps)
    a_ref = m.time_agg_layer(z_ref)                                           # [B,h,w,E]
    up = R.recipe_input(tuple(a_ref.shape), salt=32) * 0.3
    m.zero_grad()
    (a_ref * up).sum().backward()
    sd = leaf_sd(sd0)
    xo = x.clone().requires_grad_(True)
    z_o = R.patch_embed(sd, xo, cfg)
    a_o = R.time_aggregate(sd, z_o, cfg)
    (a_o * up).sum().backward()
    check("g3.embed.z", z_o, z_ref)
    check("g3.embed.agg", a_o, a_ref)
    check("g3.embed.dx", xo.grad, xr.grad)
    grads = {}
    for k in ("pos_embed", "patch_embed.proj.0.weight", "patch_embed.proj.0.bias", "patch_embed.proj.2.weight",
              "patch_embed.proj.2.bias", "time_agg_layer.w", "time_agg_layer.gamma"):
        gref = dict(m.named_parameters())[k].grad
        check("g3.embed.d" + k, sd[k].grad, gref, rtol=5e-5)
        grads["d." + k] = npy(gref)
    save("g3_embed", z=npy(z_ref), agg=npy(a_ref), dx=npy(xr.grad), **grads)

    # --- out layer + cls head
    h = cfg.latent
    lat = R.recipe_input((B, h, h, cfg.embed_dim), salt=33)
    lr_ = lat.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    cls_ref = m.cls_head(lr_.mean(dim=(2, 3)))
    o = m.out_layer(lr_).permute(0, 2, 3, 1)
    o_ref = o.reshape(*o.shape[:3], cfg.out_timesteps, cfg.out_channels).contiguous()
    up_o = R.recipe_input(tuple(o_ref.shape), salt=34) * 0.3
    up_c = R.recipe_input(tuple(cls_ref.shape), salt=35) * 0.3
    m.zero_grad()
    ((o_ref * up_o).sum() + (cls_ref * up_c).sum()).backward()
    sd = leaf_sd(sd0)
    lo = lat.clone().requires_grad_(True)
    o_o = R.out_layer(sd, lo, cfg)
    c_o = R.cls_head(sd, lo, cfg)
    ((o_o * up_o).sum() + (c_o * up_c).sum()).backward()
    check("g3.out.y", o_o, o_ref)
    check("g3.out.cls", c_o, cls_ref)
    check("g3.out.dlat", lo.grad, lr_.grad.permute(0, 2, 3, 1))
    grads = {}
    for k, p in m.named_parameters():
        if k.startswith("out_layer.") or k.startswith("cls_head."):
            check("g3.out.d" + k, sd[k].grad, p.grad, rtol=5e-5)
            grads["d." + k] = npy(p.grad)
    save("g3_out", y=npy(o_ref), cls=npy(cls_ref), dlat=npy(lr_.grad.permute(0, 2, 3, 1)), **grads)


def g4_mini(name, normalize):
    cfg = R.DPOTConfig(**dict(R.MINI, normalize=normalize))
    sd0 = R.recipe_state_dict(cfg, salt=9)
    m = ref_model(cfg, sd0)
    B = 2
    x = R.recipe_input((B, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=41)
    if normalize:
        x = x * 2.0 + 0.7
    xr = x.clone().requires_grad_(True)
    y_ref, c_ref = m(xr)
    up_y = R.recipe_input(tuple(y_ref.shape), salt=42) * 0.3
    up_c = R.recipe_input(tuple(c_ref.shape), salt=43) * 0.3
    ((y_ref * up_y).sum() + (c_ref * up_c).sum()).backward()
    sd = leaf_sd(sd0)
    xo = x.clone().requires_grad_(True)
    y_o, c_o = R.dpot_forward(sd, xo, cfg)
    ((y_o * up_y).sum() + (c_o * up_c).sum()).backward()
    check(name + ".pred", y_o, y_ref)
    check(name + ".cls", c_o, c_ref)
    check(name + ".dx", xo.grad, xr.grad, rtol=5e-5)
    grads = {}
    for k, p in m.named_parameters():
        check(name + ".d" + k, sd[k].grad, p.grad, rtol=1e-4)
        grads["d." + k] = npy(p.grad)
    save(name, pred=npy(y_ref), cls=npy(c_ref), dx=npy(xr.grad), **grads)


def g5_tiny():
    cfg = R.DPOTConfig(**R.TINY)
    sd0 = R.recipe_state_dict(cfg, salt=1)
    m = ref_model(cfg, sd0)
    n_params = sum(p.numel() for p in m.parameters())
    B = 2
    x = R.recipe_input((B, 128, 128, 10, 4), salt=51)
    with torch.no_grad():
        y_ref, c_ref = m(x)
        y_o, c_o = R.dpot_forward(sd0, x, cfg)
    check("g5.pred", y_o, y_ref)
    check("g5.cls", c_o, c_ref)
    save("g5_tiny", n_params=np.int64(n_params), pred=sub(y_ref, 37), cls=npy(c_ref))


def g6_rollout():
    cfg = R.DPOTConfig(**R.MINI)
    sd0 = R.recipe_state_dict(cfg, salt=13)
    m = ref_model(cfg, sd0)
    B, T_ar, T_bundle = 3, 3, 1
    lr, betas, wd, clip = 1e-3, (0.9, 0.9), 1e-6, 10000.0
    xx = R.recipe_input((B, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=61)
    yy = R.recipe_input((B, cfg.img_size, cfg.img_size, T_ar, cfg.out_channels), salt=62)
    msk = torch.ones(B, cfg.img_size, cfg.img_size, 1, cfg.out_channels)
    # reference objects, driven by the same sequence of calls as train_temporal.py:135,176,201-229
    opt = Adam(m.parameters(), lr=lr, betas=betas, weight_decay=wd)
    crit = SimpleLpLoss(size_average=False)
    m.train()
    cur, loss = xx, 0.0
    chunks = []
    for t in range(0, T_ar, T_bundle):
        im, _ = m(cur)
        loss = loss + crit(im, yy[..., t:t + T_bundle, :], mask=msk)
        chunks.append(im)
        cur = torch.cat((cur[..., T_bundle:, :], im), dim=-2)
    pred_ref = torch.cat(chunks, dim=-2)
    opt.zero_grad()
    loss.backward()
    gnorm_ref = torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
    gnorms = {k: p.grad.norm().item() for k, p in m.named_parameters() if p.grad is not None}
    opt.step()
    # oracle
    st = R.TrainState(params=leaf_sd(sd0))
    res = R.train_step(st, xx, yy, msk, cfg, lr=lr, betas=betas, weight_decay=wd, grad_clip=clip,
                       T_bundle=T_bundle)
    check("g6.loss", res["loss"], loss)
    check("g6.gnorm", res["grad_norm"], gnorm_ref)
    check("g6.pred", res["pred"], pred_ref)
    post = {}
    for k, p in m.named_parameters():
        if k in gnorms:
            # first-step Adam moves every weight by ~ +-lr * g/(|g|+eps): elements whose gradient is
            # ~1e-7 are rounding-sensitive, so the post-step comparison is absolute, in units of lr
            check("g6.post." + k, st.params[k], p, atol=0.05 * lr)
        post["p." + k] = sub(p, 5)
    print(f"  g6: loss={loss.item():.6f} grad_norm={gnorm_ref.item():.6f}")
    save("g6_rollout", loss=np.float64(loss.item()), grad_norm=np.float64(gnorm_ref.item()),
         pred=sub(pred_ref, 3), names=np.array(list(gnorms.keys())),
         grad_norms=np.array(list(gnorms.values()), dtype=np.float64), lr=np.float64(lr), B=np.int64(B),
         T_ar=np.int64(T_ar), **post)


def g7_loss():
    B, X, T, C = 3, 16, 2, 4
    x = R.recipe_input((B, X, X, T, C), salt=71)
    y = R.recipe_input((B, X, X, T, C), salt=72)
    msk = torch.ones(B, X, X, 1, C)
    msk[0, :, :, :, 2:] = 0.0          # sample 0: channels 2,3 void  (get_target_mask pattern)
    msk[1, :, :, :, 3] = 0.0
    xr = x.clone().requires_grad_(True)
    crit = SimpleLpLoss(size_average=False)
    l_ref = crit(xr, y, mask=msk)
    l_ref.backward()
    xo = x.clone().requires_grad_(True)
    l_o = R.rel_l2_loss(xo, y, msk)
    l_o.backward()
    check("g7.loss", l_o, l_ref)
    check("g7.dx", xo.grad, xr.grad)
    l_nomask = crit(x, y)
    check("g7.nomask", R.rel_l2_loss(x, y, None), l_nomask)
    save("g7_loss", loss=np.float64(l_ref.item()), dx=npy(xr.grad), loss_nomask=np.float64(l_nomask.item()),
         mask=npy(msk))


def g8_dp():
    """DDP averages per-rank gradients of a batch-SUM loss: grad = (g_rank0 + g_rank1) / 2, then clip + Adam on
    every rank (train_temporal_parallel.py:243-244 via accelerate/DDP)."""
    cfg = R.DPOTConfig(**R.MINI)
    sd0 = R.recipe_state_dict(cfg, salt=17)
    B = 4
    xx = R.recipe_input((B, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=81)
    yy = R.recipe_input((B, cfg.img_size, cfg.img_size, 1, cfg.out_channels), salt=82)
    msk = torch.ones(B, cfg.img_size, cfg.img_size, 1, cfg.out_channels)
    crit = SimpleLpLoss(size_average=False)
    per_rank = []
    for r in range(2):
        m = ref_model(cfg, sd0)
        sl = slice(2 * r, 2 * r + 2)
        im, cls_pred = m(xx[sl])
        total = crit(im, yy[sl], mask=msk[sl]) + 0.0 * cls_pred.sum()       # cls head gets zero-valued grads
        total.backward()
        per_rank.append({k: p.grad.clone() for k, p in m.named_parameters()})
    avg = {k: (per_rank[0][k] + per_rank[1][k]) / 2 for k in per_rank[0]}
    gnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in avg.values()))
    m = ref_model(cfg, sd0)
    for k, p in m.named_parameters():
        p.grad = avg[k].clone()
    opt = Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.9), weight_decay=1e-6)
    torch.nn.utils.clip_grad_norm_(m.parameters(), 10000.0)
    opt.step()
    save("g8_dp", grad_norm=np.float64(gnorm.item()), names=np.array(list(avg.keys())),
         grad_norms=np.array([avg[k].norm().item() for k in avg], dtype=np.float64),
         **{"p." + k: sub(p, 5) for k, p in m.named_parameters()})
    print(f"  g8: averaged grad_norm={gnorm.item():.6f}")


def smoke_reference_main():
    """the reference's own __main__ smoke configuration (models/dpot.py:462-468): shape only"""
    cfg = R.DPOTConfig(img_size=20, patch_size=5, in_channels=3, out_channels=3, in_timesteps=6, out_timesteps=1,
                       embed_dim=32, normalize=True)
    sd0 = R.recipe_state_dict(cfg, salt=19)
    m = ref_model(cfg, sd0)
    x = R.recipe_input((4, 20, 20, 6, 3), salt=91)
    with torch.no_grad():
        y_ref, c_ref = m(x)
        y_o, c_o = R.dpot_forward(sd0, x, cfg)
    check("main.pred", y_o, y_ref)
    check("main.cls", c_o, c_ref)
    save("g9_refmain", pred=npy(y_ref), cls=npy(c_ref))


VARIANTS = R.GOLDEN_VARIANTS


def g10_variant(name):
    cfg = R.DPOTConfig(**VARIANTS[name])
    sd0 = R.recipe_state_dict(cfg, salt=13)
    m = ref_model(cfg, sd0)
    x = R.recipe_input((2, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=61)
    xr = x.clone().requires_grad_(True)
    y_ref, c_ref = m(xr)
    up_y = R.recipe_input(tuple(y_ref.shape), salt=62) * 0.3
    up_c = R.recipe_input(tuple(c_ref.shape), salt=63) * 0.3
    ((y_ref * up_y).sum() + (c_ref * up_c).sum()).backward()
    sd = leaf_sd(sd0)
    xo = x.clone().requires_grad_(True)
    y_o, c_o = R.dpot_forward(sd, xo, cfg)
    ((y_o * up_y).sum() + (c_o * up_c).sum()).backward()
    check(name + ".pred", y_o, y_ref)
    check(name + ".cls", c_o, c_ref)
    check(name + ".dx", xo.grad, xr.grad, rtol=5e-5)
    arrays = {}
    for k, p in m.named_parameters():
        check(name + ".d" + k, sd[k].grad, p.grad, rtol=1e-4)
        arrays["d." + k] = sub(p.grad, 5)
    save(name, pred=npy(y_ref), cls=npy(c_ref), dx=sub(xr.grad, 3), **arrays)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    g1_afno("g1_afno_trunc", B=2, h=16, E=64, nb=4, modes=5, stride=1)
    g1_afno("g1_afno_tiny", B=2, h=16, E=512, nb=4, modes=32, stride=11)
    g2_block()
    g3_embed_out()
    g4_mini("g4_mini", normalize=False)
    g4_mini("g4_mini_norm", normalize=True)
    g5_tiny()
    g6_rollout()
    g7_loss()
    g8_dp()
    smoke_reference_main()
    for v in VARIANTS:
        g10_variant(v)
    print("all fixtures written; oracle == reference on every case")
