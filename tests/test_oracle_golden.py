"""CPU: the oracle restatement vs the golden vectors generated from the reference
(oracle/make_golden.py).  This is what pins the oracle."""
from collections import OrderedDict

import numpy as np
import torch

from oracle import dpot_ref as R
from helpers import assert_close, assert_sub, load


def leaf(sd):
    return OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())


def test_param_shapes_tiny_count():
    cfg = R.DPOTConfig(**R.TINY)
    n = sum(int(np.prod(s)) for s in R.param_shapes(cfg).values())
    fx = load("g5_tiny")
    assert n == int(fx["n_params"]) == 7534643
    assert len(R.param_shapes(cfg)) == 67


def test_afno_truncated_modes():
    fx = load("g1_afno_trunc")
    B, h, E, nb, modes = (int(fx[k]) for k in ("B", "h", "E", "nb", "modes"))
    cfg = R.DPOTConfig(img_size=h * 8, patch_size=8, embed_dim=E, n_blocks=nb, modes=modes, depth=1)
    pre = "blocks.0.filter."
    sd = leaf({k: v for k, v in R.recipe_state_dict(cfg, salt=3).items() if k.startswith(pre)})
    x = R.recipe_input((B, h, h, E), salt=11).requires_grad_(True)
    up = R.recipe_input((B, h, h, E), salt=12) * 0.3
    y = R.afno_mix(sd, pre, x, cfg)
    (y * up).sum().backward()
    assert_close(y, fx["y"], "y")
    assert_close(x.grad, fx["dx"], "dx")
    for k in ("w1", "b1", "w2", "b2"):
        assert_close(sd[pre + k].grad, fx["d" + k], "d" + k)


def test_afno_tiny_layer():
    fx = load("g1_afno_tiny")
    B, h, E, nb, modes = (int(fx[k]) for k in ("B", "h", "E", "nb", "modes"))
    cfg = R.DPOTConfig(img_size=h * 8, patch_size=8, embed_dim=E, n_blocks=nb, modes=modes, depth=1)
    pre = "blocks.0.filter."
    sd = leaf({k: v for k, v in R.recipe_state_dict(cfg, salt=3).items() if k.startswith(pre)})
    x = R.recipe_input((B, h, h, E), salt=11).requires_grad_(True)
    up = R.recipe_input((B, h, h, E), salt=12) * 0.3
    y = R.afno_mix(sd, pre, x, cfg)
    (y * up).sum().backward()
    assert_sub(y, fx, "y", "y")
    assert_sub(x.grad, fx, "dx", "dx")
    for k in ("w1", "b1", "w2", "b2"):
        assert_sub(sd[pre + k].grad, fx, "d" + k, "d" + k)


def test_block():
    fx = load("g2_block")
    B, h, E, nb = (int(fx[k]) for k in ("B", "h", "E", "nb"))
    cfg = R.DPOTConfig(img_size=h * 8, patch_size=8, embed_dim=E, n_blocks=nb, modes=32, depth=1, mlp_ratio=2)
    pre = "blocks.0."
    sd = leaf({k: v for k, v in R.recipe_state_dict(cfg, salt=5).items() if k.startswith(pre)})
    x = R.recipe_input((B, h, h, E), salt=21).requires_grad_(True)
    up = R.recipe_input((B, h, h, E), salt=22) * 0.3
    assert_close(R.group_norm_cl(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"]), fx["gn"], "gn")
    y = R.block_forward(sd, 0, x, cfg)
    (y * up).sum().backward()
    assert_close(y, fx["y"], "y")
    assert_close(x.grad, fx["dx"], "dx")
    for k in fx.files:
        if k.startswith("d."):
            assert_close(sd[pre + k[2:]].grad, fx[k], k)


def test_full_mini_model_and_grads():
    for name, normalize in (("g4_mini", False), ("g4_mini_norm", True)):
        fx = load(name)
        cfg = R.DPOTConfig(**dict(R.MINI, normalize=normalize))
        sd = leaf(R.recipe_state_dict(cfg, salt=9))
        x = R.recipe_input((2, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=41)
        if normalize:
            x = x * 2.0 + 0.7
        x.requires_grad_(True)
        y, c = R.dpot_forward(sd, x, cfg)
        up_y = R.recipe_input(tuple(y.shape), salt=42) * 0.3
        up_c = R.recipe_input(tuple(c.shape), salt=43) * 0.3
        ((y * up_y).sum() + (c * up_c).sum()).backward()
        assert_close(y, fx["pred"], name + ".pred")
        assert_close(c, fx["cls"], name + ".cls")
        assert_close(x.grad, fx["dx"], name + ".dx")
        for k in fx.files:
            if k.startswith("d."):
                assert_close(sd[k[2:]].grad, fx[k], name + "." + k)


def test_constructor_variants():
    """temporal bundling (out_timesteps=2), time_agg='mlp', mlp_ratio / n_blocks / activation / 1 kept mode"""
    for name, kw in R.GOLDEN_VARIANTS.items():
        fx = load(name)
        cfg = R.DPOTConfig(**kw)
        sd = leaf(R.recipe_state_dict(cfg, salt=13))
        x = R.recipe_input((2, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=61)
        x.requires_grad_(True)
        y, c = R.dpot_forward(sd, x, cfg)
        up_y = R.recipe_input(tuple(y.shape), salt=62) * 0.3
        up_c = R.recipe_input(tuple(c.shape), salt=63) * 0.3
        ((y * up_y).sum() + (c * up_c).sum()).backward()
        assert_close(y, fx["pred"], name + ".pred")
        assert_close(c, fx["cls"], name + ".cls")
        assert_sub(x.grad, fx, "dx", name + ".dx")
        for k in sd:
            assert_sub(sd[k].grad, fx, "d." + k, name + ".d." + k)


def test_tiny_forward():
    fx = load("g5_tiny")
    cfg = R.DPOTConfig(**R.TINY)
    sd = R.recipe_state_dict(cfg, salt=1)
    x = R.recipe_input((2, 128, 128, 10, 4), salt=51)
    with torch.no_grad():
        y, c = R.dpot_forward(sd, x, cfg)
    assert_sub(y, fx, "pred", "pred")
    assert_close(c, fx["cls"], "cls")


def test_rollout_train_step():
    fx = load("g6_rollout")
    cfg = R.DPOTConfig(**R.MINI)
    B, T_ar, lr = int(fx["B"]), int(fx["T_ar"]), float(fx["lr"])
    st = R.TrainState(params=leaf(R.recipe_state_dict(cfg, salt=13)))
    xx = R.recipe_input((B, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=61)
    yy = R.recipe_input((B, cfg.img_size, cfg.img_size, T_ar, cfg.out_channels), salt=62)
    msk = torch.ones(B, cfg.img_size, cfg.img_size, 1, cfg.out_channels)
    res = R.train_step(st, xx, yy, msk, cfg, lr=lr)
    assert abs(res["loss"].item() - float(fx["loss"])) <= 1e-4 * float(fx["loss"])
    assert abs(res["grad_norm"].item() - float(fx["grad_norm"])) <= 1e-4 * float(fx["grad_norm"])
    assert_sub(res["pred"], fx, "pred", "pred")
    for n, gn in zip(fx["names"], fx["grad_norms"]):
        assert abs(res["grads"][str(n)].norm().item() - gn) <= 1e-4 * gn + 1e-7, n
    for n in fx["names"]:
        n = str(n)
        stride = int(fx[f"p.{n}.stride"])
        got = st.params[n].detach().reshape(-1)[::stride]
        assert (got - torch.from_numpy(fx[f"p.{n}.sub"])).abs().max().item() <= 0.05 * lr, n


def test_loss_partial_mask():
    fx = load("g7_loss")
    B, X, T, C = 3, 16, 2, 4
    x = R.recipe_input((B, X, X, T, C), salt=71).requires_grad_(True)
    y = R.recipe_input((B, X, X, T, C), salt=72)
    msk = torch.from_numpy(fx["mask"])
    l = R.rel_l2_loss(x, y, msk)
    l.backward()
    assert abs(l.item() - float(fx["loss"])) <= 1e-5 * float(fx["loss"])
    assert_close(x.grad, fx["dx"], "dx")
    assert abs(R.rel_l2_loss(x.detach(), y, None).item() - float(fx["loss_nomask"])) <= 1e-5 * float(fx["loss_nomask"])


def test_reference_main_smoke_config():
    fx = load("g9_refmain")
    cfg = R.DPOTConfig(img_size=20, patch_size=5, in_channels=3, out_channels=3, in_timesteps=6, out_timesteps=1,
                       embed_dim=32, normalize=True)
    sd = R.recipe_state_dict(cfg, salt=19)
    with torch.no_grad():
        y, c = R.dpot_forward(sd, R.recipe_input((4, 20, 20, 6, 3), salt=91), cfg)
    assert tuple(y.shape) == (4, 20, 20, 1, 3)
    assert_close(y, fx["pred"], "pred")
    assert_close(c, fx["cls"], "cls")
