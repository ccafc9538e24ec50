"""GPU parity tests, stage and model level: the HIP path against (a) the committed golden vectors generated from
the reference (oracle/make_golden.py) and (b) the CPU oracle on the same seeded inputs; plus size-independent
properties at the full BASELINE size (DPOT-Tiny, B=32)."""
from collections import OrderedDict

import numpy as np
import os

import pytest
import torch

from helpers import assert_close, assert_sub, load, set_tune, tune_value
from oracle import dpot_ref as R

pytestmark = pytest.mark.gpu


def build(kw, salt):
    from dpot_amd import DPOTNet
    cfg = R.DPOTConfig(**kw)
    m = DPOTNet(**kw)
    m.load_state_dict(R.recipe_state_dict(cfg, salt=salt))
    return m.cuda(), cfg


def grads_of(m):
    return {k: p.grad for k, p in m.named_parameters()}


# ------------------------------------------------------------------------------------------------------
def _block_fn(m, x, h):
    from dpot_amd.functional import BlockFn
    blk = m.blocks[0]
    f = blk.filter
    return BlockFn.apply(x, blk.norm1.weight, blk.norm1.bias, f.w1, f.b1, f.w2, f.b2, blk.norm2.weight,
                         blk.norm2.bias, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias, h, h,
                         m.n_blocks, m.modes, m._act)


def test_afno_mixer_golden_one_launch_layer(monkeypatch):
    """the same reference golden (g1_afno_tiny: the DPOT-Tiny layer, E = 512, 4 blocks of 128) through the ONE-launch form
    of the mixer's forward (csrc/afno_fused.hip, SURVEY 8 f4: rfft2 -> both MLP layers -> irfft2 + x in one kernel), which
    `auto` only selects from 205 (sample, block) workgroups on; the backward consumes the S / pre-activation it saves"""
    from dpot_amd import ops
    set_tune(monkeypatch, afno_layer=1)
    if not (ops.afno_mlp2_supported(4, 128) and ops.afno_mlp3_supported(4, 128) and ops.afno_fused_supported(16, 16, 512, 4, 16, 9, G=0)):
        pytest.skip("one-launch AFNO layer switched off (DPOT_TUNE mixer != 3)")
    calls = []
    real = ops.afno_fused_fwd
    monkeypatch.setattr(ops, "afno_fused_fwd", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    test_afno_mixer_golden("g1_afno_tiny")
    assert calls, "the one-launch kernel did not run"


def test_afno_mixer_golden_bf16x6_kernel(monkeypatch):
    """round 6: the same reference golden (g1_afno_tiny, 128 channels per block) through the bf16x6 mixer kernel
    (csrc/afno_mlp6.hip) that gemm_precision 'auto' selects at 96 channels per block and DPOT_TUNE mixer6=2 everywhere:
    forward through the three-launch form, backward through its data-gradient form - same tolerance (fp32-accurate)"""
    from dpot_amd import ops
    set_tune(monkeypatch, mixer6=2, afno_layer=0)
    if not ops.afno_mlp6_supported(4, 128):
        pytest.skip("bf16x6 mixer switched off (DPOT_TUNE mixer = 0)")
    calls = []
    real = ops.afno_mlp2
    monkeypatch.setattr(ops, "afno_mlp2", lambda *a, **k: (calls.append(k.get("layout")), real(*a, **k))[1])
    with ops.precision_scope("auto", None):
        test_afno_mixer_golden("g1_afno_tiny")
    assert calls and all(l == 2 for l in calls), calls


@pytest.mark.parametrize("name", ["g1_afno_trunc", "g1_afno_tiny"])
def test_afno_mixer_golden(name):
    """AFNO2D alone (golden g1, written by the imported reference's AFNO2D module): the PRODUCT mixer -
    functional.AFNO2DFn = the helpers BlockFn runs: rfft2 -> afno_mlp3_kernel (both layers, three-product form; the two
    generic GEMMs at block sizes the fused kernel does not cover) -> irfft2 + residual, and backward through
    afno_mlp2(mode 1) + dpot_afno_wgrad2 - forward, dx and the four parameter gradients"""
    from dpot_amd import ops
    from dpot_amd.functional import AFNO2DFn
    fx = load(name)
    B, h, E, nb, modes = (int(fx[k]) for k in ("B", "h", "E", "nb", "modes"))
    cfg = R.DPOTConfig(img_size=h * 8, patch_size=8, embed_dim=E, n_blocks=nb, modes=modes, depth=1)
    pre = "blocks.0.filter."
    sd = {k[len(pre):]: v.cuda().requires_grad_(True) for k, v in R.recipe_state_dict(cfg, salt=3).items()
          if k.startswith(pre)}
    x = R.recipe_input((B, h, h, E), salt=11)
    up = (R.recipe_input((B, h, h, E), salt=12) * 0.3)
    bs = E // nb
    opted_out = tune_value("mixer", 3) != 3 or tune_value("panel", 1) == 0
    if name == "g1_afno_tiny" and not opted_out:      # the DPOT-Tiny layer must run on the fused three-product kernel
        assert ops.afno_mlp2_supported(nb, bs) and ops.afno_mlp3_supported(nb, bs)
        assert ops.afno_wgrad2_splitk(B * min(modes, h) * min(modes, h // 2 + 1), nb, bs) > 0
    xg = x.cuda().view(B, h * h, E).requires_grad_(True)
    y = AFNO2DFn.apply(xg, sd["w1"], sd["b1"], sd["w2"], sd["b2"], h, h, nb, modes, 1)
    (y * up.cuda().view(B, h * h, E)).sum().backward()
    full = name == "g1_afno_trunc"
    cmp = (lambda t, k: assert_close(t.reshape(fx[k].shape), fx[k], k)) if full else \
        (lambda t, k: assert_sub(t, fx, k, k))
    cmp(y.view(B, h, h, E), "y")
    cmp(xg.grad.view(B, h, h, E), "dx")
    for k in ("w1", "b1", "w2", "b2"):
        cmp(sd[k].grad, "d" + k)


def test_block_golden():
    fx = load("g2_block")
    B, h, E, nb = (int(fx[k]) for k in ("B", "h", "E", "nb"))
    kw = dict(img_size=h * 8, patch_size=8, embed_dim=E, n_blocks=nb, modes=32, depth=1, mlp_ratio=2)
    m, cfg = build(kw, salt=5)
    from dpot_amd import ops
    x = R.recipe_input((B, h, h, E), salt=21).cuda().view(B, h * h, E).requires_grad_(True)
    up = (R.recipe_input((B, h, h, E), salt=22) * 0.3).cuda().view(B, h * h, E)
    gn, _, _ = ops.groupnorm_fwd(x.detach(), m.blocks[0].norm1.weight.detach(), m.blocks[0].norm1.bias.detach())
    assert_close(gn.view(B, h, h, E), fx["gn"], "groupnorm")
    y = _block_fn(m, x, h)
    (y * up).sum().backward()
    assert_close(y.view(B, h, h, E), fx["y"], "block.y")
    assert_close(x.grad.view(B, h, h, E), fx["dx"], "block.dx")
    g = grads_of(m)
    for k in fx.files:
        if k.startswith("d."):
            assert_close(g["blocks.0." + k[2:]], fx[k], k)


def test_embed_and_head_golden():
    from dpot_amd.functional import EmbedFn, HeadFn
    kw = dict(R.MINI, depth=1)
    m, cfg = build(kw, salt=7)
    B = 2
    fx = load("g3_embed")
    x = R.recipe_input((B, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=31).cuda()
    x.requires_grad_(True)
    pe, ta = m.patch_embed.proj, m.time_agg_layer
    lat = EmbedFn.apply(x, m.pos_embed, pe[0].weight, pe[0].bias, pe[2].weight, pe[2].bias, ta.w, ta.gamma, m._gx,
                        m._gy, m._gt, m._tt, m.patch_size, m._act)
    h = cfg.latent
    up = (R.recipe_input((B, h, h, cfg.embed_dim), salt=32) * 0.3).cuda().view(B, h * h, -1)
    (lat * up).sum().backward()
    assert_close(lat.view(B, h, h, -1), fx["agg"], "embed.agg")
    assert_close(x.grad, fx["dx"], "embed.dx")
    g = grads_of(m)
    for k in fx.files:
        if k.startswith("d."):
            assert_close(g[k[2:]], fx[k], "embed." + k)
    # out layer + cls head
    fx = load("g3_out")
    m.zero_grad()
    latin = R.recipe_input((B, h, h, cfg.embed_dim), salt=33).cuda().view(B, h * h, -1).requires_grad_(True)
    ol, ch = m.out_layer, m.cls_head
    pred, cls = HeadFn.apply(latin, ol[0].weight, ol[0].bias, ol[2].weight, ol[2].bias, ol[4].weight, ol[4].bias,
                             ch[0].weight, ch[0].bias, ch[2].weight, ch[2].bias, ch[4].weight, ch[4].bias, h, h,
                             m.patch_size, m._act)
    pred = pred.view(B, cfg.img_size, cfg.img_size, cfg.out_timesteps, cfg.out_channels)
    up_o = (R.recipe_input(tuple(pred.shape), salt=34) * 0.3).cuda()
    up_c = (R.recipe_input(tuple(cls.shape), salt=35) * 0.3).cuda()
    ((pred * up_o).sum() + (cls * up_c).sum()).backward()
    assert_close(pred, fx["y"], "out.y")
    assert_close(cls, fx["cls"], "out.cls")
    assert_close(latin.grad.view(B, h, h, -1), fx["dlat"], "out.dlat")
    g = grads_of(m)
    for k in fx.files:
        if k.startswith("d."):
            assert_close(g[k[2:]], fx[k], "out." + k)


@pytest.mark.parametrize("name,normalize", [("g4_mini", False), ("g4_mini_norm", True)])
def test_full_mini_model_golden(name, normalize):
    fx = load(name)
    m, cfg = build(dict(R.MINI, normalize=normalize), salt=9)
    x = R.recipe_input((2, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=41)
    if normalize:
        x = x * 2.0 + 0.7
    x = x.cuda().requires_grad_(True)
    y, c = m(x)
    assert y.is_contiguous() and tuple(y.shape) == (2, cfg.img_size, cfg.img_size, 1, cfg.out_channels)
    up_y = (R.recipe_input(tuple(y.shape), salt=42) * 0.3).cuda()
    up_c = (R.recipe_input(tuple(c.shape), salt=43) * 0.3).cuda()
    ((y * up_y).sum() + (c * up_c).sum()).backward()
    assert_close(y, fx["pred"], "pred")
    assert_close(c, fx["cls"], "cls")
    assert_close(x.grad, fx["dx"], "dx")
    g = grads_of(m)
    for k in fx.files:
        if k.startswith("d."):
            assert_close(g[k[2:]], fx[k], k)


@pytest.mark.parametrize("name", sorted(R.GOLDEN_VARIANTS))
def test_constructor_variants_golden(name):
    """temporal bundling (out_timesteps=2), time_agg='mlp', mlp_ratio / n_blocks / activation / 1 kept mode"""
    fx = load(name)
    m, cfg = build(R.GOLDEN_VARIANTS[name], salt=13)
    x = R.recipe_input((2, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=61)
    x = x.cuda().requires_grad_(True)
    y, c = m(x)
    assert tuple(y.shape) == (2, cfg.img_size, cfg.img_size, cfg.out_timesteps, cfg.out_channels)
    up_y = (R.recipe_input(tuple(y.shape), salt=62) * 0.3).cuda()
    up_c = (R.recipe_input(tuple(c.shape), salt=63) * 0.3).cuda()
    ((y * up_y).sum() + (c * up_c).sum()).backward()
    assert_close(y, fx["pred"], "pred")
    assert_close(c, fx["cls"], "cls")
    assert_sub(x.grad, fx, "dx", "dx")
    for k, g in grads_of(m).items():
        assert_sub(g, fx, "d." + k, "d." + k)


def test_reference_main_config_golden():
    fx = load("g9_refmain")
    kw = dict(img_size=20, patch_size=5, in_channels=3, out_channels=3, in_timesteps=6, out_timesteps=1, embed_dim=32,
              normalize=True)
    m, cfg = build(kw, salt=19)
    with torch.no_grad():
        y, c = m(R.recipe_input((4, 20, 20, 6, 3), salt=91).cuda())
    assert tuple(y.shape) == (4, 20, 20, 1, 3)
    assert_close(y, fx["pred"], "pred")
    assert_close(c, fx["cls"], "cls")


def test_tiny_forward_golden():
    fx = load("g5_tiny")
    m, cfg = build(R.TINY, salt=1)
    with torch.no_grad():
        y, c = m(R.recipe_input((2, 128, 128, 10, 4), salt=51).cuda())
    assert_sub(y, fx, "pred", "tiny.pred")
    assert_close(c, fx["cls"], "tiny.cls")


def test_rollout_train_step_golden():
    """3-step AR rollout + backward + clip + Adam (golden g6 from the reference's own Adam/SimpleLpLoss)"""
    from dpot_amd.train import FlatParams, FusedAdam, train_step
    fx = load("g6_rollout")
    m, cfg = build(R.MINI, salt=13)
    B, T_ar, lr = int(fx["B"]), int(fx["T_ar"]), float(fx["lr"])
    xx = R.recipe_input((B, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=61).cuda()
    yy = R.recipe_input((B, cfg.img_size, cfg.img_size, T_ar, cfg.out_channels), salt=62).cuda()
    msk = torch.ones(B, cfg.img_size, cfg.img_size, 1, cfg.out_channels, device="cuda")
    fp = FlatParams(m)
    opt = FusedAdam(fp, lr=lr, betas=(0.9, 0.9), weight_decay=1e-6, max_norm=10000.0)
    cls_before = m.cls_head[0].weight.detach().clone()
    loss, pred = train_step(m, opt, xx, yy, msk)
    assert abs(loss.item() - float(fx["loss"])) <= 1e-4 * float(fx["loss"])
    assert abs(opt.grad_norm().item() - float(fx["grad_norm"])) <= 1e-4 * float(fx["grad_norm"])
    assert_sub(pred, fx, "pred", "rollout.pred")
    g = grads_of(m)
    for n, gn in zip(fx["names"], fx["grad_norms"]):
        assert abs(g[str(n)].norm().item() - gn) <= 1e-4 * gn + 1e-7, n
    sd = m.state_dict()
    for n in fx["names"]:
        n = str(n)
        stride = int(fx[f"p.{n}.stride"])
        got = sd[n].detach().cpu().reshape(-1)[::stride]
        assert (got - torch.from_numpy(fx[f"p.{n}.sub"])).abs().max().item() <= 0.05 * lr, n
    assert torch.equal(m.cls_head[0].weight.detach(), cls_before)      # no gradient -> skipped, as the reference


def test_train_step_vs_oracle_two_steps_and_graph():
    """eager step == oracle step, and the hipGraph-captured step reproduces the eager one bit-for-bit"""
    from dpot_amd.train import FlatParams, FusedAdam, GraphedTrainStep, train_step
    kw = R.MINI
    B = 4
    xx = R.recipe_input((B, 32, 32, 4, 3), salt=1).cuda()
    yy = R.recipe_input((B, 32, 32, 1, 3), salt=2).cuda()
    msk = torch.ones(B, 32, 32, 1, 3, device="cuda")
    lr = 1e-3
    # oracle
    cfg = R.DPOTConfig(**kw)
    st = R.TrainState(params=OrderedDict((k, v.clone()) for k, v in R.recipe_state_dict(cfg, salt=23).items()))
    ref_losses = [R.train_step(st, xx.cpu(), yy.cpu(), msk.cpu(), cfg, lr=lr)["loss"].item() for _ in range(2)]
    # eager
    m, _ = build(kw, salt=23)
    opt = FusedAdam(FlatParams(m), lr=lr, betas=(0.9, 0.9), weight_decay=1e-6, max_norm=10000.0)
    losses = [train_step(m, opt, xx, yy, msk)[0].item() for _ in range(2)]
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-4 * abs(b), (losses, ref_losses)
    eager_params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    # graphed (the eager warm-up inside GraphedTrainStep is rolled back: it is not a training step)
    m2, _ = build(kw, salt=23)
    opt2 = FusedAdam(FlatParams(m2), lr=lr, betas=(0.9, 0.9), weight_decay=1e-6, max_norm=10000.0)
    gs = GraphedTrainStep(m2, opt2, xx, yy, msk, warmup=1)
    l1 = gs.replay(lr).item()
    l2 = gs.replay(lr).item()
    assert abs(l1 - losses[0]) <= 1e-6 * abs(losses[0])
    assert abs(l2 - losses[1]) <= 1e-6 * abs(losses[1])
    for k, v in m2.state_dict().items():
        assert (v - eager_params[k]).abs().max().item() <= 1e-6, k


@pytest.mark.parametrize("out_channels,act", [(4, "gelu"), (3, "gelu"), (4, "silu"), (9, "tanh")])
def test_head_fused_tail_vs_oracle(out_channels, act):
    """out_layer_dim == 32 takes the fused per-pixel tail kernels (csrc/tail.hip): forward + every gradient vs oracle"""
    from dpot_amd.functional import HeadFn
    kw = dict(R.MINI, depth=1, out_layer_dim=32, out_channels=out_channels, act=act)
    m, cfg = build(kw, salt=29)
    B, h = 3, cfg.latent
    lat = R.recipe_input((B, h, h, cfg.embed_dim), salt=33)
    sd = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in R.recipe_state_dict(cfg, salt=29).items())
    lo = lat.clone().requires_grad_(True)
    o_ref, c_ref = R.out_layer(sd, lo, cfg), R.cls_head(sd, lo, cfg)
    up_o = R.recipe_input(tuple(o_ref.shape), salt=34) * 0.3
    up_c = R.recipe_input(tuple(c_ref.shape), salt=35) * 0.3
    ((o_ref * up_o).sum() + (c_ref * up_c).sum()).backward()
    latin = lat.cuda().view(B, h * h, -1).requires_grad_(True)
    ol, ch = m.out_layer, m.cls_head
    pred, cls = HeadFn.apply(latin, ol[0].weight, ol[0].bias, ol[2].weight, ol[2].bias, ol[4].weight, ol[4].bias,
                             ch[0].weight, ch[0].bias, ch[2].weight, ch[2].bias, ch[4].weight, ch[4].bias, h, h,
                             m.patch_size, m._act)
    pred = pred.view(B, cfg.img_size, cfg.img_size, cfg.out_timesteps, cfg.out_channels)
    ((pred * up_o.cuda()).sum() + (cls * up_c.cuda()).sum()).backward()
    assert_close(pred, o_ref, "fused tail fwd")
    assert_close(latin.grad.view(B, h, h, -1), lo.grad, "fused tail dlat")
    for k, p in m.named_parameters():
        if k.startswith("out_layer.") or k.startswith("cls_head."):
            assert_close(p.grad, sd[k].grad, "fused tail d" + k)


# ------------------------------------------------------------------------------------------------------
# full-size properties (DPOT-Tiny, B=32: the BASELINE configs[1] workload)
# ------------------------------------------------------------------------------------------------------
def test_tiny_b32_properties():
    from dpot_amd import ops
    m, cfg = build(R.TINY, salt=1)
    B = 32
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 128, 128, 10, 4, generator=g).cuda()
    with torch.no_grad():
        y, c = m(x)
        y2, _ = m(x)
        assert torch.isfinite(y).all() and torch.isfinite(c).all()
        assert torch.equal(y, y2), "forward must be deterministic"
        # batch independence: sample 5 alone == sample 5 inside the batch
        y5, _ = m(x[5:6].contiguous())
        assert_close(y5, y[5:6], "batch independence")
        # rfft2 -> irfft2 round trip at full size (all modes kept) is the identity
        lat = torch.randn(B, 256, 512, generator=g).cuda()
        S = ops.rfft2(lat, 16, 16, 4, 16, 9, 0)
        back = ops.irfft2(S, B, 16, 16, 512, 4, 16, 9, 1)
        assert_close(back, lat, "rfft2/irfft2 round trip")
        # Parseval with the Hermitian column weights: ||x||^2 == sum_k w_k |X_k|^2
        Sw = S.view(B, 16, 9, 4, 2, 128)
        wts = torch.tensor([1.0] + [2.0] * 7 + [1.0], device="cuda").view(1, 1, 9, 1, 1, 1)
        assert abs((Sw ** 2 * wts).sum().item() / (lat ** 2).sum().item() - 1.0) < 1e-5
    # one AR train step runs at full size and produces finite gradients for every trained tensor
    from dpot_amd.train import FlatParams, FusedAdam, train_step
    opt = FusedAdam(FlatParams(m), lr=1e-3, betas=(0.9, 0.9), weight_decay=1e-6, max_norm=10000.0)
    yy = torch.randn(B, 128, 128, 1, 4, generator=g).cuda()
    msk = torch.ones(B, 128, 128, 1, 4, device="cuda")
    loss, _ = train_step(m, opt, x, yy, msk)
    assert torch.isfinite(loss) and torch.isfinite(opt.fp.grad).all() and torch.isfinite(opt.fp.flat).all()
    # oracle comparison of the loss on a 2-sample slice of the same weights (CPU oracle runs in ~1 s)
    m2, _ = build(R.TINY, salt=1)
    with torch.no_grad():
        from dpot_amd.functional import rel_l2_loss
        p2, _ = m2(x[:2].contiguous())
        l_gpu = rel_l2_loss(p2, yy[:2].contiguous(), msk[:2].contiguous()).item()
        po, _ = R.dpot_forward(R.recipe_state_dict(cfg, salt=1), x[:2].cpu(), cfg)
        l_ref = R.rel_l2_loss(po, yy[:2].cpu(), msk[:2].cpu()).item()
    assert_close(p2, po, "tiny pred vs oracle")
    assert abs(l_gpu - l_ref) <= 1e-4 * abs(l_ref)


@pytest.mark.parametrize("name", ["SMALL", "MEDIUM"])
def test_baseline_configs_forward_vs_oracle(name):
    """BASELINE.json configs[2..3]: DPOT-S / -M (mlp_ratio 4, 8 blocks) forward against the CPU oracle, B=1.  (DPOT-L's
    forward AND gradients: tests/test_gpu_sizes.py - oracle at batch 4, reference golden numbers at batch 1 x 20 steps and
    at batch 16.)"""
    kw = getattr(R, name)
    m, cfg = build(kw, salt=0)
    x = R.recipe_input((1, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels))
    with torch.no_grad():
        ref, ref_cls = R.dpot_forward(R.recipe_state_dict(cfg, salt=0), x, cfg)
        out, cls = m(x.cuda())
    assert_close(out, ref, f"{name} pred")
    assert_close(cls, ref_cls, f"{name} cls")


def test_inference_rollout_vs_oracle_and_graph_replay():
    """evaluate.py:193-213: no-grad rollout feeding the model its own prediction; eager and hipGraph-replayed"""
    from dpot_amd.infer import GraphedRollout, rollout_eval
    m, cfg = build(R.MINI, salt=2)
    m.eval()
    B, T_ar = 3, 4
    S = cfg.img_size
    xx = R.recipe_input((B, S, S, cfg.in_timesteps, cfg.in_channels), salt=5)
    yy = R.recipe_input((B, S, S, T_ar, cfg.out_channels), salt=6)
    msk = torch.ones(B, S, S, 1, cfg.out_channels)
    msk[1, :, :, :, 2] = 0.0                                            # one masked-out channel
    sd = R.recipe_state_dict(cfg, salt=2)
    with torch.no_grad():
        loss_ref, pred_ref = R.rollout_loss(sd, xx, yy, msk, cfg)
        full_ref = R.rel_l2_loss(pred_ref, yy, msk)
    pred, l_steps, l_full = rollout_eval(m, xx.cuda(), yy.cuda(), msk.cuda())
    assert_close(pred, pred_ref, "rollout pred (4 AR steps)")
    assert_close(l_steps, loss_ref, "sum of step losses")
    assert_close(l_full, full_ref, "full-trajectory loss")
    g = GraphedRollout(m, xx.cuda())
    pred_g, l_steps_g, l_full_g = g(xx.cuda(), yy.cuda(), msk.cuda())
    assert torch.equal(pred_g, pred) and torch.equal(l_steps_g, l_steps) and torch.equal(l_full_g, l_full)
    with pytest.raises(ValueError):
        g(xx.cuda()[:2], yy.cuda()[:2], msk.cuda()[:2])


def test_graphed_step_draws_fresh_noise_every_replay():
    """train_temporal.py:205 inside the captured step: the in-kernel generator's device-side offset advances on each
    replay, so two replays of the same batch from the same weights see different noise (and no noise = same loss)"""
    from dpot_amd import ops
    from dpot_amd.train import FlatParams, FusedAdam, GraphedTrainStep
    m, cfg = build(R.MINI, salt=1)
    S = cfg.img_size
    xx = R.recipe_input((2, S, S, cfg.in_timesteps, cfg.in_channels), salt=3).cuda()
    yy = R.recipe_input((2, S, S, 1, cfg.out_channels), salt=4).cuda()
    msk = torch.ones(2, S, S, 1, cfg.out_channels, device="cuda")
    opt = FusedAdam(FlatParams(m), lr=0.0, betas=(0.9, 0.9), weight_decay=0.0, max_norm=1e4)   # lr 0: weights frozen
    st = ops.rng_state(xx.device)
    g = GraphedTrainStep(m, opt, xx, yy, msk, noise_scale=0.05, warmup=1)
    off0 = int(st[1].item())
    l1 = g.replay(0.0).item()
    l2 = g.replay(0.0).item()
    assert int(st[1].item()) == off0 + 2
    assert l1 != l2 and abs(l1 - l2) < 0.2 * abs(l1)
    g0 = GraphedTrainStep(m, opt, xx, yy, msk, noise_scale=0.0, warmup=1)
    assert g0.replay(0.0).item() == g0.replay(0.0).item()
