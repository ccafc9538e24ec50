"""GPU parity at the BASELINE.json sizes: full-model GRADIENTS of DPOT-Tiny / -Small / -Medium / -Large(256^2) against
the CPU oracle, and the configs[4] workload - a 20-step auto-regressive DPOT-Large rollout train step - with and
without activation recomputation.  (The toy-size golden tests live in test_gpu_model.py.)"""
from collections import OrderedDict

import pytest
import torch

from helpers import RTOL, assert_close
from oracle import dpot_ref as R

pytestmark = pytest.mark.gpu


def build(kw, salt):
    from dpot_amd import DPOTNet
    cfg = R.DPOTConfig(**kw)
    m = DPOTNet(**kw)
    m.load_state_dict(R.recipe_state_dict(cfg, salt=salt))
    return m.cuda(), cfg


def _rel(a, b):
    return abs(a - b) / (abs(b) + 1e-30)


@pytest.mark.parametrize("name,B", [("TINY", 2), ("SMALL", 1), ("MEDIUM", 1), ("LARGE", 1)])
def test_full_model_gradients_vs_oracle(name, B):
    """every parameter gradient + dx of the whole model at the BASELINE sizes (Tiny: nb=4/bs=128, S/M: nb=8, mlp_ratio 4,
    L: 256^2, nb=16/bs=96 edge tiles, out_layer_dim=128 un-fused tail, 32x32 DFT) vs torch autograd on the CPU oracle"""
    kw = getattr(R, name)
    m, cfg = build(kw, salt=4)
    S = cfg.img_size
    x = R.recipe_input((B, S, S, cfg.in_timesteps, cfg.in_channels), salt=71)
    up_y = R.recipe_input((B, S, S, cfg.out_timesteps, cfg.out_channels), salt=72) * 0.3
    up_c = R.recipe_input((B, cfg.n_cls), salt=73) * 0.3
    # oracle
    sd = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in R.recipe_state_dict(cfg, salt=4).items())
    xo = x.clone().requires_grad_(True)
    yo, co = R.dpot_forward(sd, xo, cfg)
    ((yo * up_y).sum() + (co * up_c).sum()).backward()
    # HIP path
    xg = x.cuda().requires_grad_(True)
    y, c = m(xg)
    ((y * up_y.cuda()).sum() + (c * up_c.cuda()).sum()).backward()
    assert_close(y, yo.detach(), f"{name} pred")
    assert_close(c, co.detach(), f"{name} cls")
    assert_close(xg.grad, xo.grad, f"{name} dx")
    worst = 0.0
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        worst = max(worst, assert_close(p.grad, sd[k].grad, f"{name} d{k}"))
        n_ref = sd[k].grad.double().norm().item()            # float64 accumulation: torch's fp32 norm of a 10M-element
        assert _rel(p.grad.double().norm().item(), n_ref) <= RTOL, f"{name} |d{k}|"   # tensor is itself off by ~5e-4
    print(f"[{name}] worst normalised gradient error {worst:.2e}")


def _oracle_rollout_ckpt(sd, xx, yy, msk, cfg):
    """R.rollout_loss with every AR step under torch.utils.checkpoint: the same arithmetic, one step of activations
    alive at a time (the 20-step DPOT-L rollout would otherwise keep ~100 GB of CPU activations)"""
    from torch.utils.checkpoint import checkpoint
    names = list(sd.keys())

    def fwd(x, *ps):
        return R.dpot_forward(OrderedDict(zip(names, ps)), x, cfg)[0]

    loss, preds = 0.0, []
    for t in range(yy.shape[-2]):
        im = checkpoint(fwd, xx, *sd.values(), use_reentrant=False)
        loss = loss + R.rel_l2_loss(im, yy[..., t:t + 1, :], msk)
        preds.append(im.detach())
        xx = torch.cat((xx[..., 1:, :], im), dim=-2)
    return loss, torch.cat(preds, dim=-2)


def _gpu_rollout_grads(kw, salt, xx, yy, msk, recompute):
    from dpot_amd.train import FlatParams, FusedAdam, rollout
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    m, _ = build(kw, salt=salt)
    m.recompute_blocks = recompute
    fp = FlatParams(m)
    opt = FusedAdam(fp, lr=1e-3, betas=(0.9, 0.9), weight_decay=1e-6, max_norm=10000.0)
    opt.zero_grad()
    loss, pred = rollout(m, xx.cuda(), yy.cuda(), msk.cuda())
    loss.backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    norms = {k: p.grad.double().norm().item() for k, p in m.named_parameters()}
    return loss.item(), opt.grad_norm().item(), norms, pred.detach(), fp.grad.clone(), peak


def test_large_20_step_rollout_vs_reference_golden():
    """BASELINE configs[4]: DPOT-Large 256^2, modes 64, 20-step auto-regressive rollout (configs/pretrain_large.yaml,
    train_temporal.py:201-230), B=1: loss, global grad norm, per-tensor grad norms and the 20-step prediction against
    golden numbers produced by the imported reference (oracle/make_golden_large.py - the CPU run takes ~10 min, so it
    is a committed fixture); with activation recomputation (BlockFn re-runs its forward in backward) and without -
    the two must agree bit for bit"""
    from helpers import assert_sub, load
    fx = load("g11_large_rollout")
    T_ar, B = int(fx["T_ar"]), int(fx["B"])
    cfg = R.DPOTConfig(**R.LARGE)
    S = cfg.img_size
    xx = R.recipe_input((B, S, S, cfg.in_timesteps, cfg.in_channels), salt=81)
    yy = R.recipe_input((B, S, S, T_ar, cfg.out_channels), salt=82)
    msk = torch.ones(B, S, S, 1, cfg.out_channels)
    res = {}
    for recompute in (True, False):
        loss, gn, norms, pred, flat, peak = _gpu_rollout_grads(R.LARGE, 6, xx, yy, msk, recompute)
        print(f"[LARGE T_ar={T_ar} B={B}] recompute={recompute}: loss {loss:.6f} (reference {float(fx['loss']):.6f}), "
              f"|g| {gn:.6e} (reference {float(fx['grad_norm']):.6e}), peak memory {peak:.2f} GiB")
        assert _rel(loss, float(fx["loss"])) <= RTOL
        assert _rel(gn, float(fx["grad_norm"])) <= RTOL
        assert_sub(pred, fx, "pred", "20-step prediction")
        for n, want in zip(fx["names"], fx["grad_norms"]):
            assert _rel(norms[str(n)], float(want)) <= RTOL + 1e-7 / (float(want) + 1e-30), str(n)
        res[recompute] = (loss, flat, peak)
        del pred
    assert res[True][0] == res[False][0]
    assert torch.equal(res[True][1], res[False][1]), "recomputation must not change a single bit"
    assert res[True][2] < 0.6 * res[False][2], "recomputation should cut the activation memory"


def test_tiny_5_step_rollout_vs_oracle():
    """T_ar=5 DPOT-Tiny rollout (B=2) against the CPU oracle run live: loss, grad norms, prediction; +- recomputation"""
    kw, T_ar, B = R.TINY, 5, 2
    cfg = R.DPOTConfig(**kw)
    S = cfg.img_size
    xx = R.recipe_input((B, S, S, cfg.in_timesteps, cfg.in_channels), salt=81)
    yy = R.recipe_input((B, S, S, T_ar, cfg.out_channels), salt=82)
    msk = torch.ones(B, S, S, 1, cfg.out_channels)
    sd = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in R.recipe_state_dict(cfg, salt=6).items())
    loss_ref, pred_ref = _oracle_rollout_ckpt(sd, xx, yy, msk, cfg)
    loss_ref.backward()
    ref_norms = {k: v.grad.double().norm().item() for k, v in sd.items() if v.grad is not None}
    ref_total = sum(v ** 2 for v in ref_norms.values()) ** 0.5
    res = {}
    for recompute in (True, False):
        loss, gn, norms, pred, flat, peak = _gpu_rollout_grads(kw, 6, xx, yy, msk, recompute)
        print(f"[TINY T_ar={T_ar} B={B}] recompute={recompute}: loss {loss:.6f} (oracle {loss_ref.item():.6f}), "
              f"|g| {gn:.6e} (oracle {ref_total:.6e}), peak memory {peak:.2f} GiB")
        assert _rel(loss, loss_ref.item()) <= RTOL
        assert _rel(gn, ref_total) <= RTOL
        assert_close(pred, pred_ref, "rollout pred")
        for k, want in ref_norms.items():
            assert _rel(norms[k], want) <= RTOL + 1e-7 / (want + 1e-30), k
        res[recompute] = (loss, flat, peak)
    assert res[True][0] == res[False][0] and torch.equal(res[True][1], res[False][1])
    assert res[True][2] < res[False][2]


def test_recompute_with_noise_and_graph_capture():
    """recomputation inside a captured T_ar=3 step with in-kernel noise: replays are reproducible and the step trains"""
    from dpot_amd.train import FlatParams, FusedAdam, GraphedTrainStep
    m, cfg = build(R.MINI, salt=2)
    m.recompute_blocks = True
    S = cfg.img_size
    xx = R.recipe_input((2, S, S, cfg.in_timesteps, cfg.in_channels), salt=3).cuda()
    yy = R.recipe_input((2, S, S, 3, cfg.out_channels), salt=4).cuda()
    msk = torch.ones(2, S, S, 1, cfg.out_channels, device="cuda")
    opt = FusedAdam(FlatParams(m), lr=1e-3, betas=(0.9, 0.9), weight_decay=1e-6, max_norm=1e4)
    g = GraphedTrainStep(m, opt, xx, yy, msk, noise_scale=0.01, warmup=1)
    losses = [g.replay(1e-3).item() for _ in range(6)]
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0]
