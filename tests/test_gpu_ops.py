"""GPU parity tests, kernel level: every C-ABI entry point against the CPU oracle / fp64 torch on the same
seeded inputs.  Tolerance: rtol 1e-4 (north_star, fp32) with an absolute term of 1e-4 * max|ref|."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import assert_close, set_tune
from oracle import dpot_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from dpot_amd import ops as _ops
    from dpot_amd import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


ACTS = {"gelu": torch.nn.functional.gelu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "relu": torch.relu,
        "leaky_relu": lambda v: torch.nn.functional.leaky_relu(v, 0.1), "softplus": torch.nn.functional.softplus,
        "ELU": torch.nn.functional.elu, "silu": torch.nn.functional.silu}


# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("transA,transB", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K,tile", [(256, 256, 128, 0), (100, 35, 77, 0), (130, 260, 36, 128), (8192, 512, 512, 0),
                                        (33, 12, 512, 0)])
def test_gemm_layouts(ops, transA, transB, M, N, K, tile):
    A = rnd(K, M, seed=1) if transA else rnd(M, K, seed=1)
    B = rnd(N, K, seed=2) if transB else rnd(K, N, seed=2)
    ref = (A.t() if transA else A).double() @ (B.t() if transB else B).double()
    Ad, Bd = A.cuda(), B.cuda()
    C = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm(Ad, Bd, C, M, N, K, transA=transA, transB=transB, lda=A.shape[1], ldb=B.shape[1], ldc=N, tile=tile)
    assert_close(C, ref, f"gemm {M}x{N}x{K} tA={transA} tB={transB}")


@pytest.mark.parametrize("act", list(ACTS))
def test_gemm_epilogue_activations(ops, act):
    M, N, K = 200, 96, 64
    A, W, b = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=0.3), rnd(N, seed=5)
    res = rnd(M, N, seed=6)
    a_id = ops.ACT_IDS[act]
    y, pre = ops.linear_fwd(A.cuda(), W.cuda(), b.cuda(), act=a_id, save_pre=True, res=res.cuda())
    pre_ref = A.double() @ W.double().t() + b.double()
    assert_close(pre, pre_ref, "preact")
    assert_close(y, ACTS[act](pre_ref) + res.double(), f"act {act}")
    # DACT epilogue: dx = (dy @ W2) * act'(aux)
    dy = rnd(M, N, seed=7)
    W2 = rnd(N, K, seed=8, scale=0.3)
    aux = rnd(M, K, seed=9)
    dx = ops.linear_bwd_data(dy.cuda(), W2.cuda(), act=a_id, aux=aux.cuda())
    a = aux.double().requires_grad_(True)
    ACTS[act](a).sum().backward()
    assert_close(dx, (dy.double() @ W2.double()) * a.grad, f"dact {act}")


def test_gemm_residual_row_map_and_accumulate(ops):
    M, N, K, T, tok = 240, 64, 36, 4, 12                    # rows are (b, tok, t): res row = (m / T) % tok
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    pos = rnd(tok, N, seed=4)
    y, _ = ops.linear_fwd(A.cuda(), W.cuda(), b.cuda(), res=pos.cuda(), res_div=T, res_mod=tok)
    idx = (torch.arange(M) // T) % tok
    assert_close(y, A.double() @ W.double().t() + b.double() + pos.double()[idx], "row-mapped residual")
    C = rnd(M, N, seed=5).cuda()
    C0 = C.clone()
    ops.gemm(A.cuda(), W.cuda(), C, M, N, K, transB=True, lda=K, ldb=K, ldc=N, accumulate=True)
    assert_close(C, C0.cpu().double() + A.double() @ W.double().t(), "accumulate")


def test_gemm_batched_strided_and_splitk(ops):
    nb, bs, Mm = 4, 24, 300                                  # the AFNO mixer call pattern, bs not a multiple of 32
    E2 = 2 * bs * nb
    S, Wb, bb = rnd(Mm, E2, seed=1), rnd(nb, 2 * bs, 2 * bs, seed=2, scale=0.2), rnd(nb, 2 * bs, seed=3)
    O = torch.full((Mm, E2), float("nan"), device="cuda")
    Opre = torch.empty_like(O)
    kw = dict(lda=E2, ldb=2 * bs, ldc=E2, batch=nb, strideA=2 * bs, strideB=4 * bs * bs, strideC=2 * bs)
    ops.gemm(S.cuda(), Wb.cuda(), O, Mm, 2 * bs, 2 * bs, bias=bb.cuda(), strideBias=2 * bs, act=1, mode=ops.EPI_ACT,
             preact=Opre, ldpre=E2, stridePre=2 * bs, **kw)
    ref = torch.einsum("mki,kio->mko", S.double().view(Mm, nb, 2 * bs), Wb.double()) + bb.double()
    assert_close(Opre, ref.reshape(Mm, E2), "batched preact")
    assert_close(O, torch.nn.functional.gelu(ref).reshape(Mm, E2), "batched gelu")
    # wgrad pattern: dW_k = S_k^T dO_k with split-K, deterministic
    dO = rnd(Mm, E2, seed=4)
    outs = []
    for _ in range(2):
        dW = torch.empty(nb, 2 * bs, 2 * bs, device="cuda")
        ops.gemm(S.cuda(), dO.cuda(), dW, 2 * bs, 2 * bs, Mm, transA=True, lda=E2, ldb=E2, ldc=2 * bs, batch=nb,
                 strideA=2 * bs, strideB=2 * bs, strideC=4 * bs * bs, splitk=5)
        outs.append(dW)
    refw = torch.einsum("mki,mko->kio", S.double().view(Mm, nb, 2 * bs), dO.double().view(Mm, nb, 2 * bs))
    assert_close(outs[0], refw, "split-K wgrad")
    assert torch.equal(outs[0], outs[1]), "split-K reduction must be deterministic"


@pytest.mark.parametrize("splitk", [1, 5])
def test_gemm_fused_column_sums(ops, splitk):
    """bias gradients ride on the wgrad GEMM: colsum_of=1 sums A's columns (A stored [K,M]), =2 B's (B stored [K,N])"""
    nb, bs, Mm = 4, 24, 333                                  # batched mixer wgrad: db = colsum(dO) per block
    E2 = 2 * bs * nb
    S, dO = rnd(Mm, E2, seed=1), rnd(Mm, E2, seed=4)
    dW = torch.empty(nb, 2 * bs, 2 * bs, device="cuda")
    db = torch.full((E2,), float("nan"), device="cuda")
    ops.gemm(S.cuda(), dO.cuda(), dW, 2 * bs, 2 * bs, Mm, transA=True, lda=E2, ldb=E2, ldc=2 * bs, batch=nb,
             strideA=2 * bs, strideB=2 * bs, strideC=4 * bs * bs, splitk=splitk, colsum_out=db, colsum_of=2,
             strideColsum=2 * bs)
    assert_close(dW, torch.einsum("mki,mko->kio", S.double().view(Mm, nb, 2 * bs), dO.double().view(Mm, nb, 2 * bs)),
                 "wgrad next to the column sums")
    assert_close(db, dO.double().sum(0), "colsum of B")
    for M, N, K in [(70000, 200, 130), (1000, 512, 512), (32, 12, 512)]:   # nn.Linear wgrad + bias grad, several tiles
        dy, x = rnd(M, N, seed=1), rnd(M, K, seed=2)
        dWl, dbl = ops.linear_bwd_wb(dy.cuda(), x.cuda())
        assert_close(dWl, dy.double().t() @ x.double(), f"wgrad {M}x{N}x{K}")
        assert_close(dbl, dy.double().sum(0), f"bias grad {M}x{N}x{K}")
        dWl2, dbl2 = ops.linear_bwd_wb(dy.cuda(), x.cuda())
        assert torch.equal(dbl, dbl2) and torch.equal(dWl, dWl2)


def test_linear_bwd_weight_auto_splitk_large_k(ops):
    M, N, K = 65536, 32, 32                                  # out-layer tail wgrad: tiny output, huge contraction
    dy, x = rnd(M, N, seed=1), rnd(M, K, seed=2)
    dW = ops.linear_bwd_weight(dy.cuda(), x.cuda())
    assert_close(dW, dy.double().t() @ x.double(), "wgrad")


# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,h,w,E,nb,modes", [(2, 16, 16, 64, 4, 32), (3, 16, 16, 512, 4, 32), (2, 16, 16, 64, 4, 5),
                                              (2, 32, 32, 96, 2, 64), (2, 4, 4, 32, 4, 32), (1, 14, 14, 24, 3, 6),
                                              (2, 5, 7, 8, 1, 32), (4, 32, 32, 1536, 16, 64), (8, 32, 32, 512, 4, 64),
                                              # round 3: register FFTs for the 8x8 / 64x64 latent grids (64^2 and 512^2
                                              # fields at patch 8, utils/griddataset.py:35), full and truncated mode sets;
                                              # 128x128 (1024^2) stays on the generic direct-sum kernel
                                              (16, 8, 8, 512, 4, 32), (2, 8, 8, 64, 4, 3), (2, 64, 64, 64, 4, 32),
                                              (1, 64, 64, 512, 4, 64),
                                              # round 4: 128 x 128 (1024^2 fields at patch 8) as two 64-point register FFTs
                                              # per line - 4-channel chunks (<= 36 kept columns) and 2-channel chunks
                                              # (all 65 columns incl. the Nyquist one), truncated and full mode sets
                                              (1, 128, 128, 8, 1, 32), (2, 128, 128, 8, 2, 20), (1, 128, 128, 12, 2, 128),
                                              (1, 128, 128, 6, 3, 64),
                                              # round 5: 3 * 2^k grids (192^2 / 384^2 / 768^2 fields at patch 8) on register
                                              # FFTs with a radix-3 front stage - full and truncated mode sets, both chunk widths
                                              (2, 24, 24, 64, 4, 32), (2, 24, 24, 8, 2, 7), (1, 48, 48, 32, 2, 20),
                                              (1, 48, 48, 12, 3, 48), (1, 96, 96, 8, 1, 64), (1, 96, 96, 4, 2, 30)])
def test_rfft2_irfft2_vs_torch(ops, B, h, w, E, nb, modes):
    bs = E // nb
    mx, my = min(modes, h), min(modes, w // 2 + 1)
    x = rnd(B, h, w, E, seed=1)
    spec = ops.rfft2(x.cuda().view(B, h * w, E), h, w, nb, mx, my, 0)            # [B*mx*my, 2E]
    ref = torch.fft.rfft2(x.double(), dim=(1, 2), norm="ortho")[:, :mx, :my]     # [B,mx,my,E]
    got = spec.cpu().view(B, mx, my, nb, 2, bs)
    assert_close(got[..., 0, :].reshape(B, mx, my, E), ref.real, "rfft2.re")
    assert_close(got[..., 1, :].reshape(B, mx, my, E), ref.imag, "rfft2.im")
    # inverse on an arbitrary (non-Hermitian) spectrum, with residual
    sre, sim = rnd(B, mx, my, E, seed=2), rnd(B, mx, my, E, seed=3)
    res = rnd(B, h, w, E, seed=4)
    full = torch.zeros(B, h, w // 2 + 1, E, dtype=torch.complex128)
    full[:, :mx, :my] = torch.complex(sre.double(), sim.double())
    yref = torch.fft.irfft2(full, s=(h, w), dim=(1, 2), norm="ortho") + res.double()
    planar = torch.stack([sre.view(B, mx, my, nb, bs), sim.view(B, mx, my, nb, bs)], dim=-2).reshape(B * mx * my, 2 * E)
    y = ops.irfft2(planar.cuda(), B, h, w, E, nb, mx, my, 1, res=res.cuda().view(B, h * w, E))
    assert_close(y.view(B, h, w, E), yref, "irfft2")


@pytest.mark.parametrize("h,mx,my", [(16, 7, 6), (24, 24, 13), (48, 9, 25)])
def test_dft_adjoints_match_autograd(ops, h, mx, my):
    """col_weights variants are the exact adjoints torch.autograd uses for irfft2 / rfft2 (16-point lines; round 5: the
    3 * 2^k lines too, with every mode kept incl. the Nyquist column)"""
    B, w, E, nb = 2, h, 32, 4
    bs = E // nb
    x = rnd(B, h, w, E, seed=1).double().requires_grad_(True)
    G = torch.complex(rnd(B, mx, my, E, seed=2).double(), rnd(B, mx, my, E, seed=3).double())
    s = torch.fft.rfft2(x, dim=(1, 2), norm="ortho")[:, :mx, :my]
    (s.real * G.real + s.imag * G.imag).sum().backward()
    planar = torch.stack([G.real.view(B, mx, my, nb, bs), G.imag.view(B, mx, my, nb, bs)], dim=-2)
    gx = ops.irfft2(planar.reshape(B * mx * my, 2 * E).float().cuda(), B, h, w, E, nb, mx, my, 0)
    assert_close(gx.view(B, h, w, E), x.grad, "adjoint of rfft2")
    Sr = rnd(B, mx, my, E, seed=4).double().requires_grad_(True)
    Si = rnd(B, mx, my, E, seed=5).double().requires_grad_(True)
    full = torch.zeros(B, h, w // 2 + 1, E, dtype=torch.complex128)
    full[:, :mx, :my] = torch.complex(Sr, Si)
    g = rnd(B, h, w, E, seed=6)
    (torch.fft.irfft2(full, s=(h, w), dim=(1, 2), norm="ortho") * g.double()).sum().backward()
    gs = ops.rfft2(g.cuda().view(B, h * w, E), h, w, nb, mx, my, 1).cpu().view(B, mx, my, nb, 2, bs)
    assert_close(gs[..., 0, :].reshape(B, mx, my, E), Sr.grad, "adjoint of irfft2 (re)")
    assert_close(gs[..., 1, :].reshape(B, mx, my, E), Si.grad, "adjoint of irfft2 (im)")


def test_afno_pack_unpack(ops):
    nb, bs = 3, 8
    w, b = rnd(2, nb, bs, bs, seed=1), rnd(2, nb, bs, seed=2)
    wb, bb = ops.afno_pack(w.cuda(), b.cuda())
    ref = torch.cat([torch.cat([w[0], w[1]], dim=2), torch.cat([-w[1], w[0]], dim=2)], dim=1)
    assert torch.equal(wb.cpu(), ref)
    assert torch.equal(bb.cpu(), b.permute(1, 0, 2).reshape(nb, 2 * bs))
    dwb, dbb = rnd(nb, 2 * bs, 2 * bs, seed=3), rnd(nb, 2 * bs, seed=4)
    dw, db = ops.afno_unpack_grad(dwb.cuda(), dbb.cuda(), nb, bs)
    assert_close(dw[0], dwb[:, :bs, :bs] + dwb[:, bs:, bs:], "dWr")
    assert_close(dw[1], dwb[:, :bs, bs:] - dwb[:, bs:, :bs], "dWi")
    assert torch.equal(db.cpu(), dbb.view(nb, 2, bs).permute(1, 0, 2).contiguous())


# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T,E", [(2, 64, 64), (3, 256, 512), (2, 16, 96), (2, 1024, 1536), (1, 9, 8)])
def test_groupnorm(ops, B, T, E):
    x = (rnd(B, T, E, seed=1) * 2.0 + 0.5)
    gamma, beta = rnd(E, seed=2) * 0.3 + 1.0, rnd(E, seed=3) * 0.2
    dy, add = rnd(B, T, E, seed=4), rnd(B, T, E, seed=5)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yref = torch.nn.functional.group_norm(xd.permute(0, 2, 1), 8, gd, bd, 1e-5).permute(0, 2, 1)
    (yref * dy.double()).sum().backward()
    y, mean, rstd = ops.groupnorm_fwd(x.cuda(), gamma.cuda(), beta.cuda())
    assert_close(y, yref, "gn fwd")
    dx, dg, db = ops.groupnorm_bwd(dy.cuda(), x.cuda(), mean, rstd, gamma.cuda(), add=add.cuda())
    assert_close(dx, xd.grad + add.double(), "gn dx")
    assert_close(dg, gd.grad, "gn dgamma")
    assert_close(db, bd.grad, "gn dbeta")


# ------------------------------------------------------------------------------------------------------
def test_patchify_unpatchify(ops):
    B, X, T, Cc, P = 2, 24, 3, 2, 4
    x = rnd(B, X, X, T, Cc, seed=1)
    gx, gt = R.unit_grid(X), R.unit_grid(T)
    A = ops.patchify(x.cuda(), gx.cuda(), gx.cuda(), gt.cuda(), P)
    ref = R.patchify(R.append_grid(x), P).reshape(-1, (Cc + 3) * P * P)
    assert torch.equal(A.cpu(), ref)
    dA = rnd(*A.shape, seed=2)
    dx = ops.unpatchify(dA.cuda(), B, X, X, T, Cc, P)
    xr = x.clone().requires_grad_(True)
    (R.patchify(R.append_grid(xr), P).reshape(-1, (Cc + 3) * P * P) * dA).sum().backward()
    assert torch.equal(dx.cpu(), xr.grad)


def test_small_data_movement_ops(ops):
    B, h, w, P, Cc = 2, 3, 5, 4, 3
    z = rnd(B * h * w * P * P, Cc, seed=1)
    out = ops.pixel_shuffle(z.cuda(), B, h, w, P, Cc)
    ref = z.view(B, h, w, P, P, Cc).permute(0, 1, 3, 2, 4, 5).reshape(B, h * P, w * P, Cc)
    assert torch.equal(out.cpu(), ref)
    back = ops.pixel_shuffle(out, B, h, w, P, Cc, inverse=True)
    assert torch.equal(back.cpu(), z)
    src = rnd(35, 448, seed=2)
    assert torch.equal(ops.copy2d_pad(src.cuda(), 35, 448, 36, 448).cpu(), torch.cat([src, torch.zeros(1, 448)]))
    assert torch.equal(ops.copy2d_pad(src.cuda(), 35, 448, 30, 440).cpu(), src[:30, :440])
    t = rnd(5, 70, 33, seed=3)
    assert torch.equal(ops.transpose2d(t.cuda(), 5, 70, 33).cpu(), t.transpose(1, 2).contiguous())
    Xm = rnd(5000, 77, seed=4)
    assert_close(ops.colsum(Xm.cuda(), 5000, 77), Xm.double().sum(0), "colsum")
    assert_close(ops.colsum(Xm.cuda(), 5000, 40, ld=77), Xm[:, :40].double().sum(0), "colsum ld")
    Bq, Rr, T, N = 3, 6, 4, 40
    Xg = rnd(Bq * Rr * T, N, seed=5)
    assert_close(ops.group_rowsum(Xg.cuda(), Bq, Rr, T, N), Xg.double().view(Bq, Rr, T, N).sum((0, 2)), "group_rowsum")
    xt = rnd(3, 50, 64, seed=6)
    assert_close(ops.token_mean(xt.cuda()), xt.double().mean(1), "token_mean")
    dy, addt = rnd(3, 64, seed=7), rnd(3, 50, 64, seed=8)
    assert_close(ops.token_mean_bwd(dy.cuda(), 50, add=addt.cuda()), dy.double()[:, None, :] / 50 + addt.double(),
                 "token_mean_bwd")
    sc, sh = rnd(3, 64, seed=9), rnd(3, 64, seed=10)
    assert_close(ops.scale_shift(xt.cuda(), sc.cuda(), sh.cuda()), xt.double() * sc.double()[:, None] + sh.double()[:, None],
                 "scale_shift")
    dxs, dsc, dsh = ops.scale_shift_bwd(addt.cuda(), xt.cuda(), sc.cuda())
    assert_close(dxs, addt.double() * sc.double()[:, None], "scale_shift_bwd dx")
    assert_close(dsc, (addt.double() * xt.double()).sum(1), "scale_shift_bwd dscale")
    assert_close(dsh, addt.double().sum(1), "scale_shift_bwd dshift")


def test_timeagg_scale(ops):
    T, E = 10, 64
    w = rnd(T, E, E, seed=1)
    gamma = (2 ** torch.linspace(-10, 10, E)).unsqueeze(0) * (0.9 + 0.2 * torch.rand(1, E, generator=torch.Generator().manual_seed(2)))
    tt = torch.linspace(0, 1, T)
    ws = ops.timeagg_scale_w(w.cuda(), gamma.cuda(), tt.cuda())
    temb = torch.cos(tt.unsqueeze(-1) @ gamma)                                   # fp32 arguments, as the reference
    assert_close(ws, w.double() * temb.double()[:, :, None], "timeagg scale")
    dws = rnd(T, E, E, seed=3)
    wd, gd = w.double().requires_grad_(True), gamma.double().requires_grad_(True)
    ((wd * torch.cos(tt.double().unsqueeze(-1) @ gd)[:, :, None]) * dws.double()).sum().backward()
    dw, dg = ops.timeagg_scale_w_bwd(dws.cuda(), w.cuda(), gamma.cuda(), tt.cuda())
    assert_close(dw, wd.grad, "timeagg dw", rtol=2e-4, atol_scale=2e-4)       # cos of fp32-rounded t*gamma (~1e3 rad)
    assert_close(dg, gd.grad, "timeagg dgamma", rtol=2e-4, atol_scale=2e-4)


# ------------------------------------------------------------------------------------------------------
def test_rel_l2_loss_and_grad(ops):
    from dpot_amd.functional import rel_l2_loss
    B, X, T, Cc = 3, 16, 2, 4
    x, y = rnd(B, X, X, T, Cc, seed=1), rnd(B, X, X, T, Cc, seed=2)
    msk = torch.ones(B, X, X, 1, Cc)
    msk[0, :, :, :, 2:] = 0.0
    msk[1, :, :, :, 3] = 0.0
    for m in (msk, None):
        xr = x.clone().requires_grad_(True)
        lref = R.rel_l2_loss(xr, y, m)
        lref.backward()
        xg = x.cuda().requires_grad_(True)
        l = rel_l2_loss(xg, y.cuda(), m.cuda() if m is not None else None)
        (l * 1.7).backward()
        assert abs(l.item() - lref.item()) <= 1e-5 * abs(lref.item())
        assert_close(xg.grad, 1.7 * xr.grad, "loss grad")


def test_sumsq_adam_noise(ops):
    n = 1_000_003
    g = rnd(n + 1, seed=1)[:n].contiguous()
    gd = torch.zeros(n + 5, device="cuda")[:n]
    gd.copy_(g)
    out, part = torch.zeros(1, device="cuda"), torch.zeros(1024, device="cuda")
    ops.sumsq(gd, out, part)
    assert abs(out.item() - (g.double() ** 2).sum().item()) <= 1e-6 * (g.double() ** 2).sum().item()
    # Adam vs the oracle's update rule, 3 steps, with clipping active (max_norm below the gradient norm)
    p0, m0, v0 = rnd(n, seed=2), torch.zeros(n), torch.zeros(n)
    p, m, v = p0.clone().cuda(), m0.clone().cuda(), v0.clone().cuda()
    pr, mr, vr = p0.clone(), m0.clone(), v0.clone()
    lr, b1, b2, eps, wd, max_norm, gscale = 1e-3, 0.9, 0.9, 1e-8, 1e-6, 50.0, 0.5
    hyper = torch.zeros(8, device="cuda")
    for step in range(1, 4):
        gk = rnd(n, seed=10 + step)
        ops.sumsq(gk.cuda(), out, part)
        hyper.copy_(torch.tensor([lr, b1, b2, eps, wd, 1 - b1 ** step, 1 - b2 ** step, max_norm]))
        ops.adam_step(p, gk.cuda(), m, v, hyper, out, gscale)
        gs = gk * gscale
        coef = R.clip_coef(R.grad_global_norm([gs]), max_norm)
        assert coef.item() < 1.0
        R.adam_update(pr, gs * coef, mr, vr, step, lr, b1, b2, eps, wd)
    assert (p.cpu() - pr).abs().max().item() <= 2e-3 * lr * 3
    assert_close(m, mr, "exp_avg")
    assert_close(v, vr, "exp_avg_sq")
    xx, epsn = rnd(2, 8, 8, 3, 4, seed=5), rnd(2, 8, 8, 3, 4, seed=6)
    got = ops.noise_inject(xx.cuda(), epsn.cuda(), 0.05)
    ref = xx + 0.05 * torch.sum(xx ** 2, dim=(1, 2, 3), keepdim=True) ** 0.5 * epsn
    assert_close(got, ref, "noise inject")


# ------------------------------------------------------------------------------------------------------
# bf16x6: the same fp32 GEMM computed on the bf16 matrix cores by 3-way operand splitting (csrc/gemm_split.h)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("transA,transB", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K,tile", [(256, 256, 128, 0), (100, 35, 77, 0), (130, 260, 36, 128), (1000, 384, 520, 128),
                                        (257, 129, 64, 64)])
def test_gemm_bf16x6_layouts(ops, transA, transB, M, N, K, tile):
    A, B = rnd(K, M, seed=1) if transA else rnd(M, K, seed=1), rnd(N, K, seed=2) if transB else rnd(K, N, seed=2)
    C = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm(A.cuda(), B.cuda(), C, M, N, K, transA=transA, transB=transB, lda=A.shape[1], ldb=B.shape[1], ldc=N,
             tile=tile, precision=ops.GEMM_BF16X6)
    ref = (A.double().t() if transA else A.double()) @ (B.double().t() if transB else B.double())
    assert_close(C, ref, f"bf16x6 {M}x{N}x{K} tA={transA} tB={transB}")


def test_gemm_bf16x6_is_fp32_accurate(ops):
    """the split GEMM must be as close to the exact (fp64) product as the native fp32 MFMA GEMM is - that is what
    makes it an fp32 GEMM and not a reduced-precision one.  Wide dynamic range on purpose (1e-3 .. 1e3 magnitudes)."""
    g = torch.Generator().manual_seed(5)
    for (M, N, K) in [(512, 512, 512), (1024, 256, 4096), (384, 640, 96)]:
        A = torch.randn(M, K, generator=g) * torch.logspace(-3, 3, K)[None, :]
        B = torch.randn(N, K, generator=g) * torch.logspace(2, -2, K)[None, :]
        ref = A.double() @ B.double().t()
        errs = {}
        for name, prec in (("f32", ops.GEMM_F32), ("bf16x6", ops.GEMM_BF16X6)):
            C = torch.empty(M, N, device="cuda")
            ops.gemm(A.cuda(), B.cuda(), C, M, N, K, transB=True, lda=K, ldb=K, ldc=N, precision=prec)
            errs[name] = ((C.cpu().double() - ref).abs().max() / ref.abs().max()).item()
        assert errs["bf16x6"] <= 1.5 * errs["f32"] + 1e-7, errs
        assert errs["bf16x6"] < 5e-6, errs


def test_gemm_bf16x6_epilogue_splitk_colsum(ops):
    M, N, K = 300, 200, 4100                                  # wgrad pattern with a partial last K-slab
    dy, x = rnd(K, M, seed=1), rnd(K, N, seed=2)
    for splitk in (1, 7):
        dW = torch.full((M, N), float("nan"), device="cuda")
        db = torch.full((M,), float("nan"), device="cuda")
        ops.gemm(dy.cuda(), x.cuda(), dW, M, N, K, transA=True, lda=M, ldb=N, ldc=N, splitk=splitk, colsum_out=db,
                 colsum_of=1, precision=ops.GEMM_BF16X6, tile=128)
        assert_close(dW, dy.double().t() @ x.double(), f"bf16x6 wgrad sk{splitk}")
        assert_close(db, dy.double().sum(0), f"bf16x6 fused bias grad sk{splitk}")
    # fused bias + GELU + saved pre-activation epilogue, batched / strided like the AFNO mixer
    nb, bs, Mm = 4, 24, 300
    E2 = 2 * bs * nb
    S, Wb, bb = rnd(Mm, E2, seed=1), rnd(nb, 2 * bs, 2 * bs, seed=2, scale=0.2), rnd(nb, 2 * bs, seed=3)
    O = torch.full((Mm, E2), float("nan"), device="cuda")
    Opre = torch.empty_like(O)
    ops.gemm(S.cuda(), Wb.cuda(), O, Mm, 2 * bs, 2 * bs, bias=bb.cuda(), strideBias=2 * bs, act=1, mode=ops.EPI_ACT,
             preact=Opre, ldpre=E2, stridePre=2 * bs, lda=E2, ldb=2 * bs, ldc=E2, batch=nb, strideA=2 * bs,
             strideB=4 * bs * bs, strideC=2 * bs, precision=ops.GEMM_BF16X6)
    ref = torch.einsum("mki,kio->mko", S.double().view(Mm, nb, 2 * bs), Wb.double()) + bb.double()
    assert_close(Opre, ref.reshape(Mm, E2), "bf16x6 batched preact")
    assert_close(O, torch.nn.functional.gelu(ref).reshape(Mm, E2), "bf16x6 batched gelu")


def test_model_step_gemm_auto_matches_f32(ops):
    """whole DPOT-Tiny fwd+bwd with precision 'auto' (large GEMMs on bf16x6) against the native-fp32 run"""
    from dpot_amd import DPOTNet
    from oracle import dpot_ref as R
    cfg = R.DPOTConfig(**R.TINY)
    sd = R.recipe_state_dict(cfg)
    x = R.recipe_input((4, 128, 128, 10, 4)).cuda()
    outs = {}
    try:
        for prec in ("f32", "auto"):
            ops.set_gemm_precision(prec)
            m = DPOTNet(**R.TINY).cuda()
            m.load_state_dict(sd)
            pred, _ = m(x)
            pred.square().sum().backward()
            outs[prec] = (pred.detach(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    finally:
        ops.set_gemm_precision("f32")
    assert_close(outs["auto"][0], outs["f32"][0].cpu(), "pred auto vs f32")
    for k, g in outs["f32"][1].items():
        assert_close(outs["auto"][1][k], g.cpu(), f"grad {k} auto vs f32")


def test_block_mlp_precision_override(ops):
    """ops.set_mlp_precision('bf16x6') moves only the channel-MLP GEMMs to the split kernel: same result to fp32
    accuracy, and the AFNO branch is bit-identical (checked with the MLP weights zeroed)"""
    from dpot_amd import DPOTNet
    cfg = R.DPOTConfig(**R.MINI)
    m = DPOTNet(**R.MINI).cuda()
    m.load_state_dict(R.recipe_state_dict(cfg, salt=4))
    x = R.recipe_input((2, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=9).cuda()
    try:
        with torch.no_grad():
            y32, _ = m(x)
            ops.set_mlp_precision("bf16x6")
            y6, _ = m(x)
            assert not torch.equal(y6, y32)
            assert_close(y6, y32.cpu(), "mlp on bf16x6 vs native")
            for blk in m.blocks:
                blk.mlp[0].weight.zero_()
                blk.mlp[2].weight.zero_()
            y6z, _ = m(x)
            ops.set_mlp_precision(None)
            y32z, _ = m(x)
            assert torch.equal(y6z, y32z)
    finally:
        ops.set_mlp_precision(None)


@pytest.mark.parametrize("precision", [0, 1])
def test_gemm_afno_wgrad_epilogue(ops, precision):
    """split-K reduction that un-packs dWbig -> dw[2,nb,bs,bs], db[2,nb,bs] == wgrad GEMM followed by afno_unpack_grad"""
    nb, bs, Mm = 3, 40, 700
    E2 = 2 * bs * nb
    S, dO = rnd(Mm, E2, seed=1), rnd(Mm, E2, seed=4)
    dw = torch.full((2, nb, bs, bs), float("nan"), device="cuda")
    db = torch.full((2, nb, bs), float("nan"), device="cuda")
    ops.gemm(S.cuda(), dO.cuda(), dw, 2 * bs, 2 * bs, Mm, transA=True, lda=E2, ldb=E2, ldc=2 * bs, batch=nb,
             strideA=2 * bs, strideB=2 * bs, strideC=4 * bs * bs, splitk=5, colsum_out=db, colsum_of=2,
             mode=ops.EPI_AFNO_WGRAD, precision=precision)
    dWbig = torch.einsum("mki,mko->kio", S.double().view(Mm, nb, 2 * bs), dO.double().view(Mm, nb, 2 * bs))
    ref_w = torch.stack([dWbig[:, :bs, :bs] + dWbig[:, bs:, bs:], dWbig[:, :bs, bs:] - dWbig[:, bs:, :bs]])
    ref_b = dO.double().sum(0).view(nb, 2, bs).permute(1, 0, 2)
    assert_close(dw, ref_w, "AFNO wgrad un-packed by the split-K reduction")
    assert_close(db, ref_b, "AFNO bias grad")


def test_noise_inject_in_kernel_generator(ops):
    """eps drawn inside the kernel (Philox4x32-10 + Box-Muller): right scale per (b,c), N(0,1) statistics, a fresh
    draw per call (the device-side offset advances), reproducible from the same {seed, offset}"""
    B, X, T, C = 3, 32, 10, 4
    xx = (rnd(B, X, X, T, C, seed=3) * torch.tensor([1.0, 5.0, 0.2, 2.0])).cuda()
    st = ops.rng_state(xx.device)
    st.copy_(torch.tensor([1234, 0], device=st.device))
    s = 0.05
    o1 = ops.noise_inject(xx, None, s)
    assert int(st[1].item()) == 1
    o2 = ops.noise_inject(xx, None, s)
    assert int(st[1].item()) == 2
    st.copy_(torch.tensor([1234, 0], device=st.device))
    o1b = ops.noise_inject(xx, None, s)
    assert torch.equal(o1, o1b) and not torch.equal(o1, o2)
    n = xx.double().pow(2).sum(dim=(1, 2, 3), keepdim=True).sqrt()
    z = ((o1.double() - xx.double()) / (s * n)).cpu()                  # should be i.i.d. N(0,1)
    assert abs(z.mean().item()) < 0.02 and abs(z.std().item() - 1.0) < 0.02
    assert abs((z ** 3).mean().item()) < 0.05 and abs((z ** 4).mean().item() - 3.0) < 0.15
    for c in range(C):                                                 # per-channel scale
        assert abs(z[..., c].std().item() - 1.0) < 0.03
    assert abs(torch.corrcoef(torch.stack([z.flatten()[:-1], z.flatten()[1:]]))[0, 1].item()) < 0.02


# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nb,bs,M", [(4, 128, 4608), (4, 128, 333), (16, 96, 544), (8, 128, 144), (2, 64, 100),
                                     (3, 32, 17), (4, 128, 1)])
@pytest.mark.parametrize("act", ["gelu", "silu"])
def test_afno_mlp2_fused_two_layers(ops, nb, bs, M, act):
    """the fused 2-layer block-diagonal complex MLP (csrc/afno_mlp.hip): forward (pre, mid, Y) and the backward data
    path (mid = dO1pre, Y = dS) against float64 torch, incl. ragged panels (M not a multiple of the panel height),
    DPOT-Tiny's full size (M = 32*16*9) and DPOT-L's 192-wide blocks"""
    assert ops.afno_mlp2_supported(nb, bs)
    N = 2 * bs
    X = rnd(M, nb * N, seed=1)
    W1 = rnd(nb, N, N, seed=2, scale=1.0 / math.sqrt(N))          # W[k][n]
    W2 = rnd(nb, N, N, seed=3, scale=1.0 / math.sqrt(N))
    b1, b2 = rnd(nb, N, seed=4, scale=0.3), rnd(nb, N, seed=5, scale=0.3)
    f = ACTS[act]
    Xd = X.double().view(M, nb, N)
    pre_ref = torch.einsum("mkn,kno->mko", Xd, W1.double()) + b1.double()
    mid_ref = f(pre_ref)
    Y_ref = torch.einsum("mkn,kno->mko", mid_ref, W2.double()) + b2.double()
    W1T, W1B = ops.afno_block_weights(W1.cuda())                 # blocked W (forward), blocked W^T (backward)
    W2T, W2B = ops.afno_block_weights(W2.cuda())
    Y, pre, mid = ops.afno_mlp2(X.cuda(), W1T, b1.cuda(), W2T, b2.cuda(), nb, bs, ops.ACT_IDS[act], mode=0,
                                want_pre=True, want_mid=True)
    assert_close(pre, pre_ref.reshape(M, -1), "pre")
    assert_close(mid, mid_ref.reshape(M, -1), "mid")
    assert_close(Y, Y_ref.reshape(M, -1), "Y")
    Yi, p_none, m_none = ops.afno_mlp2(X.cuda(), W1T, b1.cuda(), W2T, b2.cuda(), nb, bs, ops.ACT_IDS[act], mode=0)
    assert p_none is None and m_none is None and torch.equal(Yi, Y)          # inference form: same numbers, no stores
    # backward data path: dO1pre = (dO2 W2^T) * act'(pre), dS = dO1pre W1^T
    dO2 = rnd(M, nb * N, seed=6)
    pr = pre_ref.clone().requires_grad_(True)
    (f(pr)).backward(torch.ones_like(pr))
    dact = pr.grad
    dmid_ref = torch.einsum("mko,kno->mkn", dO2.double().view(M, nb, N), W2.double()) * dact
    dS_ref = torch.einsum("mko,kno->mkn", dmid_ref, W1.double())
    dS, _, dmid = ops.afno_mlp2(dO2.cuda(), W2B, None, W1B, None, nb, bs, ops.ACT_IDS[act], mode=1,
                                aux=pre_ref.float().reshape(M, -1).contiguous().cuda(), want_mid=True)
    assert_close(dmid, dmid_ref.reshape(M, -1), "dO1pre")
    assert_close(dS, dS_ref.reshape(M, -1), "dS")
    # round 3: the backward launch can also re-derive the forward's activated layer-1 output act(aux)
    dS2, o1, dmid2 = ops.afno_mlp2(dO2.cuda(), W2B, None, W1B, None, nb, bs, ops.ACT_IDS[act], mode=1,
                                   aux=pre_ref.float().reshape(M, -1).contiguous().cuda(), want_mid=True, want_pre=True)
    assert_close(o1, f(pre_ref.float().double()).reshape(M, -1), "act(aux) re-derived by the backward launch")
    assert_close(dS2, dS_ref.reshape(M, -1), "dS (with act(aux) output)")
    assert_close(dmid2, dmid_ref.reshape(M, -1), "dO1pre (with act(aux) output)")


def test_plain_bf16_mlp_mode_is_reduced_precision_but_sane(ops):
    """BASELINE configs[2] "bf16 channel-MLP on MFMA": precision 'bf16' rounds the GEMM operands to bf16 (one product per
    k-step, fp32 accumulation): error ~ 2^-8 per product, i.e. OUTSIDE the 1e-4 parity tolerance (it is opt-in), but
    within the bf16 bound against the fp32 result; forward and backward of a DPOT-Small sized MLP GEMM + a model step"""
    M, K, N = 4096, 1024, 1024
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1.0 / math.sqrt(K))
    ref = A.double() @ W.double().t()
    y32, _ = ops.linear_fwd(A.cuda(), W.cuda(), None)
    y16, _ = ops.linear_fwd(A.cuda(), W.cuda(), None, precision=ops.GEMM_BF16)
    e32 = (y32.cpu().double() - ref).norm() / ref.norm()
    e16 = (y16.cpu().double() - ref).norm() / ref.norm()
    assert e32 < 1e-6 and 1e-4 < e16 < 8e-3, (e32.item(), e16.item())
    dW = ops.linear_bwd_weight(y16, A.cuda(), precision=ops.GEMM_BF16)                # TN, split-K
    dref = y16.cpu().double().t() @ A.double()
    assert (dW.cpu().double() - dref).norm() / dref.norm() < 8e-3
    from dpot_amd import DPOTNet
    cfg = R.DPOTConfig(**R.MINI)
    m = DPOTNet(**R.MINI).cuda()
    m.load_state_dict(R.recipe_state_dict(cfg, salt=4))
    x = R.recipe_input((2, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=9).cuda()
    try:
        y_ref, _ = m(x)
        ops.set_mlp_precision("bf16")
        y_b, _ = m(x)
        (y_b ** 2).sum().backward()
        rel = ((y_b - y_ref).norm() / y_ref.norm()).item()
        assert 1e-6 < rel < 2e-2, rel
        assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    finally:
        ops.set_mlp_precision(None)


@pytest.mark.parametrize("M,N,K", [(8192, 512, 512), (333, 256, 96), (100, 64, 32), (4096, 1024, 4096), (1000, 192, 64),
                                   (77, 1536, 6144 // 4)])
def test_gemm_panel_static_weight(ops, M, N, K):
    """panel GEMM (csrc/gemm_panel.hip) with a pre-packed weight: forward form x W^T (+bias, GELU, pre-activation),
    data-gradient form dy W (* act'(aux)) through the transposed pack, residual epilogue, ragged panels"""
    assert ops.gemm_panel_supported(M, N, K)
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1.0 / math.sqrt(K))            # W [N, K] = nn.Linear layout
    b, R_ = rnd(N, seed=3, scale=0.3), rnd(M, N, seed=4)
    Wd = W.cuda()
    pk = ops.PanelPacks([(Wd, N, K, K, False), (Wd, K, N, K, True)])                 # W (forward), W^T (dgrad)
    pk.refresh()
    pre_ref = A.double() @ W.double().t() + b.double()
    y, pre = ops.gemm_panel(A.cuda(), pk.bufs[0], N, bias=b.cuda(), act=1, mode=ops.EPI_ACT, save_pre=True)
    assert_close(pre, pre_ref, "pre")
    assert_close(y, torch.nn.functional.gelu(pre_ref), "gelu(pre)")
    y2, _ = ops.gemm_panel(A.cuda(), pk.bufs[0], N, bias=b.cuda(), res=R_.cuda())
    assert_close(y2, pre_ref + R_.double(), "linear + residual")
    if dpot_ok := ops.gemm_panel_supported(M, K, N):
        dY, aux = rnd(M, N, seed=5), rnd(M, K, seed=6)
        pr = aux.double().clone().requires_grad_(True)
        torch.nn.functional.gelu(pr).backward(torch.ones_like(pr))
        dx_ref = (dY.double() @ W.double()) * pr.grad
        dx, _ = ops.gemm_panel(dY.cuda(), pk.bufs[1], K, act=1, mode=ops.EPI_DACT, aux=aux.cuda())
        assert_close(dx, dx_ref, "dgrad * gelu'")


@pytest.mark.parametrize("M,N,K", [(8192, 1024, 1024), (300, 256, 64), (4096, 512, 2048), (129, 768, 96), (1, 256, 32),
                                   (65500, 256, 64), (16400, 1024, 96)])   # >= 512 tiles: the two-workgroup kernel, ragged M
def test_gemm_bf16_panel(ops, M, N, K):
    """bf16 panel GEMM (csrc/gemm_bf16p.hip): packed bf16 operands, fp32 accumulation.  Checked against an fp64 product of
    the bf16-ROUNDED operands (exactly what the kernel multiplies: error there is fp32 accumulation only) and, loosely,
    against the un-rounded product; epilogues as for the fp32 panel kernel; ragged M."""
    assert ops.gemm_bf16p_supported(M, N, K)
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1.0 / math.sqrt(K))
    b, R_ = rnd(N, seed=3, scale=0.3), rnd(M, N, seed=4)
    Wd = W.cuda()
    pk = ops.PanelPacks([(Wd, N, K, K, False), (Wd, K, N, K, True)] if K % 256 == 0 else [(Wd, N, K, K, False)], bf16=True)
    pk.refresh()
    Ab, Wb = A.bfloat16().double(), W.bfloat16().double()
    pre_ref = Ab @ Wb.t() + b.double()
    Ap = ops.bf16_pack_rows(A.cuda())
    y, pre = ops.gemm_bf16p(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), act=1, mode=ops.EPI_ACT, save_pre=True)
    assert_close(pre, pre_ref, "pre (vs product of the rounded operands)", rtol=2e-5, atol_scale=2e-5)
    assert_close(y, torch.nn.functional.gelu(pre_ref), "gelu", rtol=2e-5, atol_scale=2e-5)
    full = A.double() @ W.double().t() + b.double()
    assert ((pre.cpu().double() - full).norm() / full.norm()).item() < 8e-3
    y2, _ = ops.gemm_bf16p(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), res=R_.cuda())
    assert_close(y2, pre_ref + R_.double(), "linear + residual", rtol=2e-5, atol_scale=2e-5)
    if K % 256 == 0:                                                 # data-gradient form through the transposed pack
        dY, aux = rnd(M, N, seed=5), rnd(M, K, seed=6)
        pr = aux.double().clone().requires_grad_(True)
        torch.nn.functional.gelu(pr).backward(torch.ones_like(pr))
        dx_ref = (dY.bfloat16().double() @ Wb) * pr.grad
        dx, _ = ops.gemm_bf16p(ops.bf16_pack_rows(dY.cuda()), pk.bufs[1], M, K, N, act=1, mode=ops.EPI_DACT, aux=aux.cuda())
        assert_close(dx, dx_ref, "dgrad * gelu'", rtol=2e-5, atol_scale=2e-5)


@pytest.mark.parametrize("M,N,K,splitk", [(1024, 512, 2048, None), (256, 256, 4096, 4), (1536, 768, 1000 * 32, None),
                                          (96, 256, 96, 1), (512, 256, 2048 + 32, 3)])
def test_gemm_bf16_panel_wgrad_splitk(ops, M, N, K, splitk):
    """weight-gradient form of the bf16 panel GEMM: dW[n, k] = dy^T x with BOTH operands packed transposed (the GEMM's
    k-dimension = tokens) and split-K over it (fixed-order reduction: deterministic); ragged last split"""
    dy, x = rnd(K, M, seed=1), rnd(K, N, seed=2)                           # [tokens, features]
    ref = dy.bfloat16().double().t() @ x.bfloat16().double()
    dyp, xp = ops.bf16_pack_rows(dy.cuda(), trans=True), ops.bf16_pack_rows(x.cuda(), trans=True)
    # the transposed pack is the row pack of the transposed matrix
    assert torch.equal(dyp, ops.bf16_pack_rows(dy.t().contiguous().cuda()))
    out = torch.full((M, N), float("nan"), device="cuda")
    dw, _ = ops.gemm_bf16p(dyp, xp, M, N, K, out=out, splitk=splitk)
    assert dw.data_ptr() == out.data_ptr()
    assert_close(dw, ref, "dW", rtol=3e-5, atol_scale=3e-5)
    dw2, _ = ops.gemm_bf16p(dyp, xp, M, N, K, splitk=splitk)
    assert torch.equal(dw, dw2), "split-K reduction must be deterministic"


def test_bf16_panel_model_step_within_bf16_bound_of_fp32(ops, monkeypatch):
    """channel-MLP precision 'bf16' routes fc1 / fc2 forward, data and weight gradients through csrc/gemm_bf16p.hip when
    the shapes allow it (E, mlp hidden multiples of 256): operands rounded to bf16, fp32 accumulation -> outputs and
    gradients sit within the bf16 bound of the fp32 model."""
    from dpot_amd import DPOTNet
    kw = dict(R.MINI, embed_dim=256, out_layer_dim=32, depth=2, mlp_ratio=1, n_blocks=4)
    cfg = R.DPOTConfig(**kw)
    x = R.recipe_input((2, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=9).cuda()

    def run(prec):
        m = DPOTNet(**kw).cuda()
        m.load_state_dict(R.recipe_state_dict(cfg, salt=4))
        ops.set_mlp_precision(prec)
        try:
            y, _ = m(x)
            (y ** 2).sum().backward()
            used = getattr(m, "_panel_packs_bf16", None) is not None
        finally:
            ops.set_mlp_precision(None)
        return y.detach(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, used

    y32, g32, used_32 = run(None)
    yp, gp, used_p = run("bf16")
    assert used_p and not used_32
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    assert 1e-6 < rel(yp, y32) < 2e-2
    for n in g32:
        assert rel(gp[n], g32[n]) < 5e-2, (n, rel(gp[n], g32[n]))


@pytest.mark.parametrize("B,X,Y,T,hid,act", [(2, 32, 32, 10, 35, 1), (3, 16, 64, 4, 35, 1), (1, 24, 32, 7, 16, 2),
                                             (2, 32, 32, 9, 48, 1), (5, 8, 32, 1, 3, 0)])
def test_implicit_patch_embed_matches_patch_matrix_path(ops, B, X, Y, T, hid, act):
    """csrc/embed.hip: the patch conv gathered from x (data channels) + the unit-grid channels folded into a bias table
    == patchify + GEMM on the materialised patch matrix (the round-1 path, itself pinned to the oracle); forward
    (Hpre, Hh) and the weight gradient incl. the grid-channel columns.  Ragged row tiles (4T % 16 != 0), T = 1,
    the widest hidden layer, several groups per workgroup."""
    from dpot_amd import functional as F
    Cc, P = 4, 8
    assert ops.embed_supported(Cc, P, T, hid, Y // P)
    K0 = (Cc + 3) * P * P
    hidp = (hid + 3) // 4 * 4
    x = rnd(B, X, Y, T, Cc, seed=1).cuda()
    w0 = rnd(hid, Cc + 3, P, P, seed=2, scale=1.0 / math.sqrt(K0)).cuda()
    b0 = rnd(hid, seed=3, scale=0.2).cuda()
    gx = torch.linspace(0, 1, X).cuda()
    gy = torch.linspace(0, 1, Y).cuda()
    gt = torch.linspace(0, 1, T).cuda()
    A0 = ops.patchify(x, gx, gy, gt, P)
    w0p = ops.copy2d_pad(w0, hid, K0, hidp, K0)
    b0p = ops.copy2d_pad(b0, 1, hid, 1, hidp).view(hidp)
    Hh_ref, Hpre_ref = ops.linear_fwd(A0, w0p, b0p, act=act, save_pre=True)
    grid = F.embed_grid_matrix(gx, gy, gt, X, Y, T, Cc, P)
    assert torch.equal(grid, A0[:grid.shape[0], Cc * P * P:])
    wfrag = ops.embed_pack_w0(w0)
    bt = torch.empty(grid.shape[0], hidp, device="cuda")
    ops.gemm(grid, w0p[:, Cc * P * P:], bt, grid.shape[0], hidp, grid.shape[1], transB=True, lda=grid.shape[1], ldb=K0,
             ldc=hidp, bias=b0p)
    Hh, Hpre = ops.embed_fwd(x, wfrag, bt, hidp, act)
    ref64 = A0.double() @ w0p.double().t() + b0p.double()
    assert_close(Hpre, ref64, "Hpre vs fp64", rtol=2e-5, atol_scale=2e-6)
    assert_close(Hpre, Hpre_ref, "Hpre vs patch-matrix path", rtol=2e-5, atol_scale=2e-6)
    assert_close(Hh[:, :hid], Hh_ref[:, :hid], "Hh", rtol=2e-5, atol_scale=2e-6)
    # weight gradient
    dH = rnd(A0.shape[0], hidp, seed=5).cuda()
    dw_ref = dH.double().t() @ A0.double()                              # [hidp, K0]
    dw0 = torch.full((hid, K0), float("nan"), device="cuda")
    ops.embed_wgrad(x, dH, dw0, hid)
    tokT = grid.shape[0]
    dHs = ops.group_rowsum(dH, B, tokT, 1, hidp)
    kg = grid.shape[1]
    ops.gemm(dHs, grid, dw0[:, K0 - kg:], hid, kg, tokT, transA=True, lda=hidp, ldb=kg, ldc=K0)
    assert_close(dw0, dw_ref[:hid], "dW0 (data + grid columns)", rtol=3e-5, atol_scale=3e-6)
    dw0b = torch.empty_like(dw0)
    ops.embed_wgrad(x, dH, dw0b, hid)
    assert torch.equal(dw0b[:, :Cc * P * P], dw0[:, :Cc * P * P]), "implicit wgrad must be deterministic"


@pytest.mark.parametrize("M,N,K", [(1000, 256, 96), (4096, 1024, 1024), (130, 512, 64)])
def test_gemm_bf16x6_panel_is_fp32_accurate(ops, M, N, K):
    """three-plane form of the bf16 panel GEMM (x = x1 + x2 + x3, six plane products): the error against fp64 is that of
    an fp32 GEMM (<= a few 2^-24 * sqrt(K)), for every epilogue, the data-gradient form and the split-K weight-gradient
    form; ragged M, odd slab counts (the ring / tail paths of the kernel)"""
    assert ops.gemm_bf16p_supported(M, N, K)
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1.0 / math.sqrt(K))
    b, R_ = rnd(N, seed=3, scale=0.3), rnd(M, N, seed=4)
    Wd = W.cuda()
    pk = ops.PanelPacks([(Wd, N, K, K, False)], bf16=True, planes=3)
    pk.refresh()
    Ap = ops.bf16_pack_rows(A.cuda(), planes=3)
    ref = A.double() @ W.double().t() + b.double()
    y32, _ = ops.linear_fwd(A.cuda(), Wd, b.cuda())                    # native fp32 MFMA for comparison
    y, pre = ops.gemm_bf16p(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), act=1, mode=ops.EPI_ACT, save_pre=True, planes=3)
    e6 = ((pre.cpu().double() - ref).norm() / ref.norm()).item()
    e32 = ((y32.cpu().double() - ref).norm() / ref.norm()).item()
    assert e6 < 2.0 * e32 + 1e-7, (e6, e32)
    assert_close(pre, ref, "pre", rtol=2e-5, atol_scale=2e-6)
    assert_close(y, torch.nn.functional.gelu(ref), "gelu", rtol=2e-5, atol_scale=2e-6)
    y2, _ = ops.gemm_bf16p(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), res=R_.cuda(), planes=3)
    assert_close(y2, ref + R_.double(), "linear + residual", rtol=2e-5, atol_scale=2e-6)
    if K % 256 == 0:
        pkT = ops.PanelPacks([(Wd, K, N, K, True)], bf16=True, planes=3)
        pkT.refresh()
        dY = rnd(M, N, seed=5)
        dx, _ = ops.gemm_bf16p(ops.bf16_pack_rows(dY.cuda(), planes=3), pkT.bufs[0], M, K, N, planes=3)
        assert_close(dx, dY.double() @ W.double(), "dgrad", rtol=2e-5, atol_scale=2e-6)
    if M % 32 == 0 or True:
        # weight-gradient form: rows = features (N resp. K), GEMM k-dim = tokens (M, padded to 32 by the pack)
        Mt = (M // 32) * 32
        if Mt >= 32 and K % 256 == 0:
            dY = rnd(Mt, N, seed=6)
            dw, _ = ops.gemm_bf16p(ops.bf16_pack_rows(dY.cuda(), trans=True, planes=3),
                                   ops.bf16_pack_rows(A[:Mt].contiguous().cuda(), trans=True, planes=3), N, K, Mt,
                                   planes=3, splitk=3 if Mt >= 96 else 1)
            assert_close(dw, dY.double().t() @ A[:Mt].double(), "wgrad (split-K)", rtol=2e-5, atol_scale=2e-6)


@pytest.mark.parametrize("M,N,K,batch,cs", [(512, 512, 8192, 1, 1), (256, 256, 4608, 1, 2), (128, 384, 2048, 1, 0),
                                            (512, 2048, 1024, 1, 0), (256, 128, 32 * 37, 1, 1)])
def test_weight_gradient_kernel_vs_fp64(ops, M, N, K, batch, cs):
    """csrc/gemm_tn.hip (A^T B over token-major operands, row-interleaved fragments, split-K through the workspace) against
    fp64: dW, the fused bias-gradient column sums of either operand, operands that are column windows of wider matrices,
    odd slab counts per split"""
    lda, ldb = M * batch, N * batch
    A, B = rnd(K, lda, seed=1), rnd(K, ldb, seed=2)
    Ad, Bd = A.cuda(), B.cuda()

    def run():
        C = torch.full((batch, M, N), float("nan"), device="cuda")
        csum = torch.full((batch, M if cs == 1 else N), float("nan"), device="cuda") if cs else None
        sk = ops.auto_splitk(M, N, K, batch, tn=True)
        ops.gemm(Ad, Bd, C, M, N, K, transA=True, lda=lda, ldb=ldb, ldc=N, batch=batch, strideA=M, strideB=N,
                 strideC=M * N, splitk=sk, colsum_out=csum, colsum_of=cs, strideColsum=(M if cs == 1 else N))
        return C, csum, sk

    C, csum, sk = run()
    assert sk >= 2 and ops._lib.load().dpot_gemm_tn_splitk(M, N, K, batch) == sk
    for z in range(batch):
        a, b = A[:, z * M:(z + 1) * M].double(), B[:, z * N:(z + 1) * N].double()
        assert_close(C[z], a.t() @ b, f"dW[{z}]", rtol=2e-5, atol_scale=2e-6)
        if cs:
            assert_close(csum[z], (a if cs == 1 else b).sum(0), f"colsum[{z}]", rtol=2e-5, atol_scale=2e-6)
    C2, _, _ = run()
    assert torch.equal(C, C2), "deterministic"


@pytest.mark.parametrize("M,K", [(128, 256), (4096, 1024), (192, 768)])
def test_bf16_pack_both_equals_the_two_single_packs(ops, M, K):
    """the fused pass produces bit-identical row-form and transposed packs to dpot_bf16_pack_rows, and the column sums"""
    x = rnd(M, K, seed=3).cuda()
    pr, pt, cs = ops.bf16_pack_both(x, want_colsum=True)
    assert torch.equal(pr, ops.bf16_pack_rows(x))
    assert torch.equal(pt, ops.bf16_pack_rows(x, trans=True))
    assert_close(cs, x.double().sum(0), "colsum", rtol=2e-5, atol_scale=2e-6)
    pr2, pt2, cs2 = ops.bf16_pack_both(x, want_rows=False)
    assert pr2 is None and cs2 is None and torch.equal(pt2, pt)


def test_bf16_mlp_pack_both_path_matches_separate_packs(ops, monkeypatch):
    """the bf16 channel MLP with one fused pack pass per activation (transposed packs saved for the backward instead of
    the fp32 activations) vs the same path with separate pack passes: forward bit for bit; the gradients agree to the
    rounding of the saved activation derivative (round 3: the packed path keeps act'(pre) as bf16 - 2^-9 relative per
    element - where the separate path re-evaluates act' from the fp32 pre-activation): norm-wise <= 8e-3 (measured up to
    4.0e-3 on this 2-block model); recomputation
    must reproduce the packed path bit for bit"""
    from dpot_amd import DPOTNet
    kw = dict(R.MINI, img_size=64, embed_dim=256, out_layer_dim=32, depth=2, mlp_ratio=1, n_blocks=4)
    cfg = R.DPOTConfig(**kw)
    x = R.recipe_input((2, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=9).cuda()

    def run(both, recompute=False):
        set_tune(monkeypatch, pack_both=1 if both else 0)
        m = DPOTNet(**kw).cuda()
        m.load_state_dict(R.recipe_state_dict(cfg, salt=4))
        m.recompute_blocks = recompute
        ops.set_mlp_precision("bf16")
        try:
            y, _ = m(x)
            (y ** 2).sum().backward()
        finally:
            ops.set_mlp_precision(None)
        return y.detach(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    y0, g0 = run(False)
    y1, g1 = run(True, False)
    y2, g2 = run(True, True)
    assert torch.equal(y0, y1) and torch.equal(y1, y2)
    for n in g0:
        assert torch.equal(g1[n], g2[n]), n                              # recomputation: same kernels, same bits
        err = ((g1[n].double() - g0[n].double()).norm() / (g0[n].double().norm() + 1e-300)).item()
        assert err <= 8e-3, (n, err)


@pytest.mark.parametrize("bs", [96, 128])
def test_bf16x6_mixer_in_the_model_with_recomputation(ops, monkeypatch, bs):
    """round 6: a 32 x 32-grid model (the DPOT-L form of the Block: chunked GroupNorm statistics, rfft2 / irfft2 with GroupNorm
    on load) with 96 / 128 channels per block under gemm_precision 'auto' runs its mixer MLP on the bf16x6 kernel
    (csrc/afno_mlp6.hip) forward AND backward: (a) prediction and every gradient agree with the native-fp32 run to fp32
    rounding; (b) activation recomputation (what `bench.py --config L20` runs: the packs travel through the autograd context)
    reproduces the stored-activation run BIT FOR BIT; (c) the kernel really ran"""
    from dpot_amd import DPOTNet
    set_tune(monkeypatch, mixer6=2)
    kw = dict(R.MINI, img_size=256, patch_size=8, embed_dim=2 * bs, out_layer_dim=32, depth=2, mlp_ratio=1, n_blocks=2, modes=32)
    cfg = R.DPOTConfig(**kw)
    x = R.recipe_input((2, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=9).cuda()
    calls = []
    real = ops.afno_mlp2
    monkeypatch.setattr(ops, "afno_mlp2", lambda *a, **k: (calls.append(k.get("layout")), real(*a, **k))[1])

    def run(prec, recompute=False):
        m = DPOTNet(**kw).cuda()
        m.load_state_dict(R.recipe_state_dict(cfg, salt=4))
        m.gemm_precision = prec
        m.recompute_blocks = recompute
        del calls[:]
        y, _ = m(x)
        (y ** 2).sum().backward()
        return y.detach(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, list(calls)

    y0, g0, c0 = run("f32")
    y1, g1, c1 = run("auto")
    y2, g2, c2 = run("auto", True)
    assert c0 and all(l != 2 for l in c0)
    assert c1 and all(l == 2 for l in c1), c1                            # forward and backward-data launches of both Blocks
    assert len(c2) > len(c1) and all(l == 2 for l in c2)                 # + the recomputed forwards
    assert_close(y1, y0.cpu(), "pred auto vs f32")
    for n in g0:
        assert_close(g1[n], g0[n].cpu(), f"grad {n} auto vs f32")
    assert torch.equal(y1, y2)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n                              # recomputation: same kernels, same bits


def _unpack_rows(pk, M, N):
    """row-form pack [M/32][N/16][64 chunks][8 bf16] (chunk l = row l & 31, columns 8 * (l >> 5) .. + 7) -> [M, N] fp32"""
    t = pk.view(M // 32, N // 16, 2, 32, 8).float()                       # [rt, kb, half, row, 8]
    return t.permute(0, 3, 1, 2, 4).reshape(M, N)


def _unpack_frag(pk, M, N):
    """act' pack in FRAGMENT order [M/32][N/32][2 halves][64 lanes][8 bf16] (csrc/gemm_bf16p.hip epi_fragment_direct):
    lane l = (column l & 31, kh = l >> 5), element j of half s is accumulator register r = 8 s + j, i.e. row
    (r & 3) + 8 (r >> 2) + 4 kh = (j & 3) + 4 kh + 8 (j >> 2) + 16 s of the 32x32 fragment  ->  [M, N] fp32"""
    t = pk.view(M // 32, N // 32, 2, 2, 32, 2, 4).float()                 # [mt, nt, s, kh, col, j >> 2, j & 3]
    return t.permute(0, 2, 5, 3, 6, 1, 4).reshape(M, N)                   # row bits [s][j >> 2][kh][j & 3]


def test_gemm_bf16_panel_saved_activation_derivative(ops):
    """round 3: the EPI_ACT launch of the bf16 channel MLP saves act'(pre-activation) as a bf16 pack instead of the fp32
    pre-activation, and the EPI_DACT launch multiplies by that pack: (a) the pack holds bf16(act'(pre)) - checked against
    torch autograd of the activation on the kernel's own fp32 pre-activation, within one bf16 ulp; (b) the data-gradient
    launch fed with the pack == the same launch fed with the fp32 pre-activation, to the rounding of the pack"""
    M, N, K = 256, 512, 256
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1.0 / math.sqrt(K)), rnd(N, seed=3, scale=0.3)
    pk = ops.PanelPacks([(W.cuda(), N, K, K, False)], bf16=True)
    pk.refresh()
    Ap = ops.bf16_pack_rows(A.cuda())
    y, pre = ops.gemm_bf16p(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), act=1, mode=ops.EPI_ACT, save_pre=True)
    y2, D, pr, _, _ = ops.gemm_bf16p_packed(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), act=1, mode=ops.EPI_ACT,
                                            save_dact=True, pack_rows=True)
    assert D.dtype == torch.bfloat16 and torch.equal(y, y2)
    p64 = pre.double().cpu().requires_grad_(True)
    torch.nn.functional.gelu(p64).sum().backward()
    want = p64.grad
    got = _unpack_frag(D, M, N).double().cpu()
    assert ((got - want).abs() <= 2.0 ** -8 * want.abs() + 1e-6).all()      # one bf16 ulp (8-bit significand)
    # (b) an act'-product epilogue on the same shapes: out = (dY W^T) * act'(pre)
    dY = rnd(M, K, seed=5)
    dYp = ops.bf16_pack_rows(dY.cuda())
    ref, _ = ops.gemm_bf16p(dYp, pk.bufs[0], M, N, K, act=1, mode=ops.EPI_DACT, aux=pre)
    out, _, _, _, _ = ops.gemm_bf16p_packed(dYp, pk.bufs[0], M, N, K, act=1, mode=ops.EPI_DACT, dact=D)
    lin, _ = ops.gemm_bf16p(dYp, pk.bufs[0], M, N, K)                      # dY W^T without the derivative
    assert torch.equal(out, lin * _unpack_frag(D, M, N))                   # exactly the product with the stored bf16
    err = ((out.double() - ref.double()).norm() / ref.double().norm()).item()
    assert err <= 3e-3, err


def test_gemm_bf16_panel_packed_epilogue_outputs(ops):
    """the bf16 panel GEMM can emit its output as packed operands (row form, transposed form) + partial column sums:
    bit-identical to packing the fp32 output afterwards; the fp32 store can be skipped"""
    M, N, K = 256, 512, 256
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1.0 / math.sqrt(K)), rnd(N, seed=3, scale=0.3)
    Wd = W.cuda()
    pk = ops.PanelPacks([(Wd, N, K, K, False)], bf16=True)
    pk.refresh()
    Ap = ops.bf16_pack_rows(A.cuda())
    y, pre = ops.gemm_bf16p(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), act=1, mode=ops.EPI_ACT, save_pre=True)
    y2, pre2, pr, pt, cs = ops.gemm_bf16p_packed(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), act=1, mode=ops.EPI_ACT,
                                                 save_pre=True, pack_rows=True, pack_trans=True, colsum=True)
    assert torch.equal(y, y2) and torch.equal(pre, pre2)
    assert torch.equal(pr, ops.bf16_pack_rows(y)) and torch.equal(pt, ops.bf16_pack_rows(y, trans=True))
    assert_close(cs, y.double().sum(0), "colsum", rtol=2e-5, atol_scale=2e-6)
    y3, _, pr3, pt3, cs3 = ops.gemm_bf16p_packed(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), act=1, mode=ops.EPI_ACT,
                                                 pack_trans=True, store=False)
    assert y3 is None and pr3 is None and cs3 is None and torch.equal(pt3, pt)


@pytest.mark.parametrize("nb,bs,M,act", [(4, 128, 4608, "gelu"), (2, 128, 333, "gelu"), (8, 128, 100, "relu"),
                                         (1, 128, 16, "gelu"), (16, 96, 2176, "gelu"), (3, 96, 50, "gelu"),
                                         (2, 64, 200, "gelu")])
def test_afno_mlp3_three_product_form(ops, nb, bs, M, act):
    """the three-product (Gauss) form of the fused complex MLP for bs = 128 (csrc/afno_mlp.hip, afno_mlp3_kernel):
    packs written by AfnoPacks (layout 1), forward (pre, mid, Y) and backward data path against float64 complex
    arithmetic and against the four-product kernel; ragged panels"""
    N = 2 * bs
    assert ops.afno_mlp3_supported(nb, bs)
    w1, w2 = rnd(2, nb, bs, bs, seed=2, scale=1.0 / math.sqrt(N)), rnd(2, nb, bs, bs, seed=3, scale=1.0 / math.sqrt(N))
    b1, b2 = rnd(2, nb, bs, seed=4, scale=0.3), rnd(2, nb, bs, seed=5, scale=0.3)
    X = rnd(M, nb * N, seed=1)
    packs = ops.AfnoPacks([(w1.cuda(), b1.cuda()), (w2.cuda(), b2.cuda())])
    assert packs.layout == 1
    (wb1, bb1, f1, bw1), (wb2, bb2, f2, bw2) = packs.refresh()
    f = ACTS[act]
    Xc = torch.complex(X.double().view(M, nb, 2, bs)[:, :, 0], X.double().view(M, nb, 2, bs)[:, :, 1])   # [M, nb, bs]
    W1c, W2c = torch.complex(w1[0].double(), w1[1].double()), torch.complex(w2[0].double(), w2[1].double())
    B1c, B2c = torch.complex(b1[0].double(), b1[1].double()), torch.complex(b2[0].double(), b2[1].double())
    pre_c = torch.einsum("mki,kio->mko", Xc, W1c) + B1c
    planar = lambda z: torch.stack([z.real, z.imag], dim=2).reshape(M, -1)      # [M, nb, 2, bs] -> [M, nb*2*bs]
    pre_ref = planar(pre_c)
    mid_ref = f(pre_ref)
    midc = torch.complex(mid_ref.view(M, nb, 2, bs)[:, :, 0], mid_ref.view(M, nb, 2, bs)[:, :, 1])
    Y_ref = planar(torch.einsum("mki,kio->mko", midc, W2c) + B2c)
    Y, pre, mid = ops.afno_mlp2(X.cuda(), f1, bb1, f2, bb2, nb, bs, ops.ACT_IDS[act], mode=0, want_pre=True,
                                want_mid=True, layout=1)
    assert_close(pre, pre_ref, "pre")
    assert_close(mid, mid_ref, "mid")
    assert_close(Y, Y_ref, "Y")
    # against the four-product kernel on the same weights (Wbig packs)
    W1T, W1B = ops.afno_block_weights(wb1)
    W2T, W2B = ops.afno_block_weights(wb2)
    Y4, pre4, mid4 = ops.afno_mlp2(X.cuda(), W1T, bb1, W2T, bb2, nb, bs, ops.ACT_IDS[act], mode=0, want_pre=True,
                                   want_mid=True)
    assert_close(Y, Y4.double(), "Y vs four-product kernel", rtol=2e-5, atol_scale=2e-5)
    # backward data path
    dO2 = rnd(M, nb * N, seed=6)
    pr = pre_ref.clone().requires_grad_(True)
    (f(pr)).backward(torch.ones_like(pr))
    dact = pr.grad
    Wbig1, Wbig2 = wb1.cpu().double(), wb2.cpu().double()                       # [nb, N, N], W[k][n]
    dmid_ref = torch.einsum("mko,kno->mkn", dO2.double().view(M, nb, N), Wbig2) * dact.view(M, nb, N)
    dS_ref = torch.einsum("mko,kno->mkn", dmid_ref, Wbig1)
    dS, o1, dmid = ops.afno_mlp2(dO2.cuda(), bw2, None, bw1, None, nb, bs, ops.ACT_IDS[act], mode=1,
                                 aux=pre_ref.float().contiguous().cuda(), want_mid=True, want_pre=True, layout=1)
    assert_close(dmid, dmid_ref.reshape(M, -1), "dO1pre")
    assert_close(dS, dS_ref.reshape(M, -1), "dS")
    assert_close(o1, f(pre_ref.float().double()), "act(aux) re-derived by the backward launch")     # == the forward's mid


@pytest.mark.parametrize("nb,bs,M,act", [(4, 128, 4608, "gelu"), (2, 128, 333, "gelu"), (8, 128, 100, "relu"),
                                         (1, 128, 16, "gelu"), (16, 96, 2176, "gelu"), (3, 96, 50, "gelu"),
                                         (16, 96, 8704 - 37, "gelu"), (5, 96, 129, "tanh")])
def test_afno_mlp6_bf16x6_form(ops, monkeypatch, nb, bs, M, act):
    """round 6: the fused complex MLP on the BF16 matrix cores at fp32 accuracy (csrc/afno_mlp6.hip: every operand split into
    three bf16 planes, six plane products, the hidden layer stays in registers in the transposed accumulator layout): forward
    (pre, mid, Y) and backward data path against float64 complex arithmetic at the SAME tolerance as the fp32 kernels, and
    against the three-product fp32 kernel; ragged row tiles; packs written by AfnoPacks under gemm_precision 'auto'"""
    set_tune(monkeypatch, mixer6=2)
    N = 2 * bs
    assert ops.afno_mlp6_supported(nb, bs)
    w1, w2 = rnd(2, nb, bs, bs, seed=2, scale=1.0 / math.sqrt(N)), rnd(2, nb, bs, bs, seed=3, scale=1.0 / math.sqrt(N))
    b1, b2 = rnd(2, nb, bs, seed=4, scale=0.3), rnd(2, nb, bs, seed=5, scale=0.3)
    X = rnd(M, nb * N, seed=1)
    with ops.precision_scope("auto", None):
        packs = ops.AfnoPacks([(w1.cuda(), b1.cuda()), (w2.cuda(), b2.cuda())])
        it1, it2 = packs.refresh()
    assert it1.p6 is not None and it2.p6 is not None
    (wb1, bb1, f1, bw1), (wb2, bb2, f2, bw2) = it1, it2
    f = ACTS[act]
    Xc = torch.complex(X.double().view(M, nb, 2, bs)[:, :, 0], X.double().view(M, nb, 2, bs)[:, :, 1])   # [M, nb, bs]
    W1c, W2c = torch.complex(w1[0].double(), w1[1].double()), torch.complex(w2[0].double(), w2[1].double())
    B1c, B2c = torch.complex(b1[0].double(), b1[1].double()), torch.complex(b2[0].double(), b2[1].double())
    pre_c = torch.einsum("mki,kio->mko", Xc, W1c) + B1c
    planar = lambda z: torch.stack([z.real, z.imag], dim=2).reshape(M, -1)      # [M, nb, 2, bs] -> [M, nb*2*bs]
    pre_ref = planar(pre_c)
    mid_ref = f(pre_ref)
    midc = torch.complex(mid_ref.view(M, nb, 2, bs)[:, :, 0], mid_ref.view(M, nb, 2, bs)[:, :, 1])
    Y_ref = planar(torch.einsum("mki,kio->mko", midc, W2c) + B2c)
    Y, pre, mid = ops.afno_mlp2(X.cuda(), it1.p6[0], bb1, it2.p6[0], bb2, nb, bs, ops.ACT_IDS[act], mode=0, want_pre=True,
                                want_mid=True, layout=2)
    assert_close(pre, pre_ref, "pre")
    assert_close(mid, mid_ref, "mid")
    assert_close(Y, Y_ref, "Y")
    # inference form (nothing but Y stored) gives the same Y
    Yi, _, _ = ops.afno_mlp2(X.cuda(), it1.p6[0], bb1, it2.p6[0], bb2, nb, bs, ops.ACT_IDS[act], mode=0, layout=2)
    assert torch.equal(Yi, Y)
    # against the fp32 three-product kernel on the same weights
    Y3, _, _ = ops.afno_mlp2(X.cuda(), f1, bb1, f2, bb2, nb, bs, ops.ACT_IDS[act], mode=0, layout=1)
    assert_close(Y, Y3.double(), "Y vs the fp32 kernel", rtol=2e-5, atol_scale=2e-5)
    # no worse than the fp32 kernel against float64 (both ~3e-7 norm-wise)
    e6 = (Y.double().cpu() - Y_ref).norm() / Y_ref.norm()
    e3 = (Y3.double().cpu() - Y_ref).norm() / Y_ref.norm()
    assert e6 <= 2.0 * e3 + 1e-7, (float(e6), float(e3))
    # backward data path
    dO2 = rnd(M, nb * N, seed=6)
    pr = pre_ref.clone().requires_grad_(True)
    (f(pr)).backward(torch.ones_like(pr))
    dact = pr.grad
    Wbig1, Wbig2 = wb1.cpu().double(), wb2.cpu().double()                       # [nb, N, N], W[k][n]
    dmid_ref = torch.einsum("mko,kno->mkn", dO2.double().view(M, nb, N), Wbig2) * dact.view(M, nb, N)
    dS_ref = torch.einsum("mko,kno->mkn", dmid_ref, Wbig1)
    dS, o1, dmid = ops.afno_mlp2(dO2.cuda(), it2.p6[1], None, it1.p6[1], None, nb, bs, ops.ACT_IDS[act], mode=1,
                                 aux=pre_ref.float().contiguous().cuda(), want_mid=True, want_pre=True, layout=2)
    assert_close(dmid, dmid_ref.reshape(M, -1), "dO1pre")
    assert_close(dS, dS_ref.reshape(M, -1), "dS")
    assert_close(o1, f(pre_ref.float().double()), "act(aux) re-derived by the backward launch")     # == the forward's mid


@pytest.mark.parametrize("nb,bs,Mm", [(4, 128, 4608), (2, 64, 32 * 13), (8, 128, 32 * 9),
                                      # round 3: bs = 96 (DPOT-Large, N = 192) on the 192 x 192-tile kernel
                                      (16, 96, 2176), (3, 96, 32 * 5), (16, 96, 32)])
def test_afno_wgrad2_both_layers_one_launch(ops, nb, bs, Mm):
    """dpot_afno_wgrad2: the weight + bias gradients of both AFNO MLP layers from one launch of the weight-gradient kernel
    (2*nb independent N x N problems) + one un-packing reduce, against float64 complex arithmetic"""
    N = 2 * bs
    sk = ops.afno_wgrad2_splitk(Mm, nb, bs)
    assert sk >= 1
    S, dO1, O1, dO2 = (rnd(Mm, nb * N, seed=k) for k in (1, 2, 3, 4))
    dw1, dw2 = (torch.full((2, nb, bs, bs), float("nan"), device="cuda") for _ in range(2))
    db1, db2 = (torch.full((2, nb, bs), float("nan"), device="cuda") for _ in range(2))
    ops.afno_wgrad2(S.cuda(), dO1.cuda(), O1.cuda(), dO2.cuda(), nb, bs, dw1, db1, dw2, db2, sk)

    def ref(A, Bm):
        Ac = A.double().view(Mm, nb, 2, bs)
        Bc = Bm.double().view(Mm, nb, 2, bs)
        Ar, Ai, Br, Bi = Ac[:, :, 0], Ac[:, :, 1], Bc[:, :, 0], Bc[:, :, 1]
        e = lambda x, y: torch.einsum("mki,mko->kio", x, y)
        return torch.stack([e(Ar, Br) + e(Ai, Bi), e(Ar, Bi) - e(Ai, Br)]), torch.stack([Br.sum(0), Bi.sum(0)])

    for (dw, db, A, Bm, nm) in ((dw1, db1, S, dO1, "layer 1"), (dw2, db2, O1, dO2, "layer 2")):
        rw, rb = ref(A, Bm)
        assert_close(dw, rw, f"dw {nm}", rtol=2e-5, atol_scale=2e-6)
        assert_close(db, rb, f"db {nm}", rtol=2e-5, atol_scale=2e-6)


@pytest.mark.parametrize("T,E,mh", [(8192, 512, 512), (32 * 11, 128, 384), (1024, 256, 128)])
def test_mlp_wgrad2_both_layers_one_launch(ops, T, E, mh):
    """dpot_mlp_wgrad2: dW2 = do2^T Hh, db2, dW1 = dHpre^T xn2 (stored un-transposed), db1 from one launch + one reduce"""
    sk = ops.mlp_wgrad2_splitk(T, E, mh)
    assert sk >= 1
    do2, Hh, xn2, dH = rnd(T, E, seed=1), rnd(T, mh, seed=2), rnd(T, E, seed=3), rnd(T, mh, seed=4)
    dW2 = torch.full((E, mh), float("nan"), device="cuda")
    dW1 = torch.full((mh, E), float("nan"), device="cuda")
    db2 = torch.full((E,), float("nan"), device="cuda")
    db1 = torch.full((mh,), float("nan"), device="cuda")
    ops.mlp_wgrad2(do2.cuda(), Hh.cuda(), xn2.cuda(), dH.cuda(), dW2, db2, dW1, db1, sk)
    assert_close(dW2, do2.double().t() @ Hh.double(), "dW2", rtol=2e-5, atol_scale=2e-6)
    assert_close(dW1, dH.double().t() @ xn2.double(), "dW1", rtol=2e-5, atol_scale=2e-6)
    assert_close(db2, do2.double().sum(0), "db2", rtol=2e-5, atol_scale=2e-6)
    assert_close(db1, dH.double().sum(0), "db1", rtol=2e-5, atol_scale=2e-6)


def test_bf16_panel_packed_epilogue_dact_with_colsum_and_strided_pack(ops):
    """the backward form the bf16 channel MLP uses: (dy W) * gelu'(aux) leaves the GEMM only as its two packs + the bias
    column sums (no fp32 store); and dpot_bf16_pack_both on a row window of a wider matrix (ld > K)"""
    M, N, K = 128, 256, 512
    dY, W, aux = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1.0 / math.sqrt(K)), rnd(M, N, seed=3)
    Wd = W.cuda()
    pk = ops.PanelPacks([(Wd, N, K, K, False)], bf16=True)
    pk.refresh()
    Ap = ops.bf16_pack_rows(dY.cuda())
    ref, _ = ops.gemm_bf16p(Ap, pk.bufs[0], M, N, K, act=1, mode=ops.EPI_DACT, aux=aux.cuda())
    c, pre, pr, pt, cs = ops.gemm_bf16p_packed(Ap, pk.bufs[0], M, N, K, act=1, mode=ops.EPI_DACT, aux=aux.cuda(),
                                               pack_rows=True, pack_trans=True, colsum=True, store=False)
    assert c is None and pre is None
    assert torch.equal(pr, ops.bf16_pack_rows(ref)) and torch.equal(pt, ops.bf16_pack_rows(ref, trans=True))
    assert_close(cs, ref.double().sum(0), "colsum", rtol=2e-5, atol_scale=2e-6)
    wide = rnd(192, 1024, seed=5).cuda()
    win = wide[64:, 256:768]                                   # [128, 512] window, ld = 1024
    r1, t1, c1 = ops.bf16_pack_both(win, want_colsum=True)
    r2, t2, c2 = ops.bf16_pack_both(win.contiguous(), want_colsum=True)
    assert torch.equal(r1, r2) and torch.equal(t1, t2) and torch.equal(c1, c2)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,act", [(32, 512, 512, 1), (5, 12, 512, 0), (33, 1024, 1024, 1), (64, 100, 1536, 0), (1, 7, 2048, 1)])
def test_small_linear_vs_fp64(ops, M, N, K, act):
    """few-row Linear (cls_head on the token mean): one wave per output column, against float64"""
    torch.manual_seed(M + N)
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    assert ops.small_linear_supported(M, N, K)
    y, pre = ops.linear_fwd(x, W, b, act=act, save_pre=True)
    ref = x.double() @ W.double().t() + b.double()
    assert_close(pre, ref, "small_linear pre")
    assert_close(y, torch.nn.functional.gelu(ref) if act else ref, "small_linear act")
    # the GEMM path on the same operands (what larger batches use) agrees
    old = os.environ.get("DPOT_TUNE")
    os.environ["DPOT_TUNE"] = (old + "," if old else "") + "fused_small=0"
    try:
        y2, _ = ops.linear_fwd(x, W, b, act=act, save_pre=True)
    finally:
        if old is None:
            del os.environ["DPOT_TUNE"]
        else:
            os.environ["DPOT_TUNE"] = old
    assert_close(y2, y.double(), "small_linear vs GEMM")


@pytest.mark.gpu
def test_groupnorm_deferred_param_grads_match(ops):
    """deferred parameter-gradient partials of two GroupNorm layers, reduced by one launch == the immediate reduction"""
    torch.manual_seed(3)
    B, T, E, G = 6, 64, 256, 8
    outs = []
    for add in (None, torch.randn(B, T, E, device="cuda")):
        x = torch.randn(B, T, E, device="cuda"); dy = torch.randn(B, T, E, device="cuda"); gw = torch.randn(E, device="cuda")
        _, mean, rstd = ops.groupnorm_fwd(x, gw, torch.zeros(E, device="cuda"), G)
        dx0, dg0, db0 = ops.groupnorm_bwd(dy, x, mean, rstd, gw, G, add=add)
        dx1, part = ops.groupnorm_bwd(dy, x, mean, rstd, gw, G, add=add, defer=True)
        assert torch.equal(dx0, dx1)
        outs.append((part, dg0, db0))
    slot = torch.zeros(E, device="cuda")
    res = ops.groupnorm_param_grads([(outs[0][0], slot, None), (outs[1][0], None, None)])
    assert res[0][0].data_ptr() == slot.data_ptr()
    for (dg, db), (_, dg0, db0) in zip(res, outs):
        assert torch.equal(dg, dg0) and torch.equal(db, db0)


@pytest.mark.gpu
def test_bf16_panel_pair_launch_matches_single(ops):
    """two weight-gradient products in one launch (csrc/gemm_bf16p.hip pair kernel) == the two single launches, bit for bit
    up to the split-K summation order of the singles (compared against the products of the bf16-rounded operands)"""
    torch.manual_seed(11)
    T, n0, k0, n1, k1 = 1024, 256, 512, 512, 256
    dy0 = torch.randn(T, n0, device="cuda"); x0 = torch.randn(T, k0, device="cuda")
    dy1 = torch.randn(T, n1, device="cuda"); x1 = torch.randn(T, k1, device="cuda")
    packs = [ops.bf16_pack_rows(t, trans=True) for t in (dy0, x0, dy1, x1)]
    C0, C1 = ops.gemm_bf16p_pair(packs[0], packs[1], n0, k0, packs[2], packs[3], n1, k1, T, splitk=1)
    r = lambda t: t.bfloat16().double()
    assert_close(C0, r(dy0).t() @ r(x0), "pair product 0")
    assert_close(C1, r(dy1).t() @ r(x1), "pair product 1")
    S0, _ = ops.gemm_bf16p(packs[0], packs[1], n0, k0, T, splitk=1)
    S1, _ = ops.gemm_bf16p(packs[2], packs[3], n1, k1, T, splitk=1)
    assert torch.equal(C0, S0) and torch.equal(C1, S1)
    assert ops.gemm_bf16p_pair_wanted(1024, 4096, 4096, 1024, 8192)
    # with a common split-K factor (few tiles, long K: DPOT-S): partial sums + fixed-order reduction
    Z0, Z1 = ops.gemm_bf16p_pair(packs[0], packs[1], n0, k0, packs[2], packs[3], n1, k1, T, splitk=2)
    assert_close(Z0, r(dy0).t() @ r(x0), "split pair product 0")
    assert_close(Z1, r(dy1).t() @ r(x1), "split pair product 1")
    # DPOT-L's weight-gradient shapes (1536 x 6144 and 6144 x 1536): 576 tiles of 128 x 256 = three rounds of 256 CUs, so
    # the pair launch runs them as 768 tiles of 128 x 192 (2 x 3 compute waves) - same products, same k order: bit-equal to
    # the single launches on 256-wide tiles
    T, n0, k0, n1, k1 = 512, 1536, 6144, 6144, 1536
    dy0 = torch.randn(T, n0, device="cuda"); x0 = torch.randn(T, k0, device="cuda")
    dy1 = torch.randn(T, n1, device="cuda"); x1 = torch.randn(T, k1, device="cuda")
    packs = [ops.bf16_pack_rows(t, trans=True) for t in (dy0, x0, dy1, x1)]
    assert ops.gemm_bf16p_pair_wanted(n0, k0, n1, k1, T)
    C0, C1 = ops.gemm_bf16p_pair(packs[0], packs[1], n0, k0, packs[2], packs[3], n1, k1, T, splitk=1)
    assert_close(C0, r(dy0).t() @ r(x0), "192-tile pair product 0")
    assert_close(C1, r(dy1).t() @ r(x1), "192-tile pair product 1")
    S0, _ = ops.gemm_bf16p(packs[0], packs[1], n0, k0, T, splitk=1)
    S1, _ = ops.gemm_bf16p(packs[2], packs[3], n1, k1, T, splitk=1)
    assert torch.equal(C0, S0) and torch.equal(C1, S1)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,E,G", [(4, 1024, 1536, 8), (2, 200, 96, 8), (1, 4096, 768, 8)])
def test_groupnorm_chunked_vs_fp64(ops, monkeypatch, B, T, E, G):
    """few, large (sample, group) slabs: the chunked kernels (statistics merged through a workspace) against float64
    and against the one-workgroup-per-slab kernels (compared via NULL workspace)"""
    torch.manual_seed(B + T)
    x = torch.randn(B, T, E, device="cuda") * 1.7 + 0.8
    gw = torch.randn(E, device="cuda"); gb = torch.randn(E, device="cuda")
    dy = torch.randn(B, T, E, device="cuda"); add = torch.randn(B, T, E, device="cuda")
    assert ops.groupnorm_ws_elems(B, T, E, G) > 0
    y, mean, rstd = ops.groupnorm_fwd(x, gw, gb, G)
    xd = x.double().view(B, T, G, E // G)
    mu = xd.mean(dim=(1, 3), keepdim=True); var = xd.var(dim=(1, 3), unbiased=False, keepdim=True)
    xh = ((xd - mu) / torch.sqrt(var + 1e-5)).view(B, T, E)
    assert_close(y, xh * gw.double() + gb.double(), "chunked groupnorm fwd")
    assert_close(mean, mu.view(B, G), "mean"); assert_close(rstd, (1 / torch.sqrt(var + 1e-5)).view(B, G), "rstd")
    dx, dg, db = ops.groupnorm_bwd(dy, x, mean, rstd, gw, G, add=add)
    xr = x.double().requires_grad_(True)
    gwr = gw.double().requires_grad_(True)
    yr = torch.nn.functional.group_norm(xr.transpose(1, 2), G, gwr, gb.double(), 1e-5).transpose(1, 2)
    yr.backward(dy.double())
    assert_close(dx, xr.grad + add.double(), "chunked groupnorm dx")
    assert_close(dg, gwr.grad, "dgamma"); assert_close(db, dy.double().sum(dim=(0, 1)), "dbeta")
    # the one-workgroup-per-slab kernels on the same data
    y0, mean0, rstd0 = ops.groupnorm_fwd(x, gw, gb, G, chunked=False)
    assert_close(y, y0.double(), "chunked vs slab kernels")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["large_mean", "outlier_first", "tiny_variance", "offset_border"])
def test_groupnorm_chunked_statistics_hard_cases(ops, case):
    """the one-sweep chunk statistics (sums around a pivot, csrc/norm.hip gn_chunk_stats_kernel) where a naive sum of squares
    fails: mean >> sigma, an outlier as the very first element of a chunk (it enters the pivot), nearly constant data, and
     a border region at the START of every chunk's sample that sits ~100 sigma away from the rest - the pivot is
    sampled with a stride over the whole chunk, so such a region enters it only in proportion to its size"""
    B, T, E, G = 2, 2048, 768, 8
    torch.manual_seed(5)
    x = torch.randn(B, T, E, device="cuda")
    if case == "large_mean":
        x = x * 0.05 + 300.0
    elif case == "outlier_first":
        x[:, 0, 0] = 1.0e4
        x[:, T // 2, 96] = -3.0e3
    elif case == "offset_border":
        x = x * 0.01
        x[:, :32, :] += 1.0                      # the first 32 tokens of every sample: +100 sigma (a masked / constant border)
    else:
        x = x * 1e-4 + 2.0
    gw = torch.ones(E, device="cuda"); gb = torch.zeros(E, device="cuda")
    assert ops.groupnorm_ws_elems(B, T, E, G) > 0
    y, mean, rstd = ops.groupnorm_fwd(x, gw, gb, G)
    xd = x.double().view(B, T, G, E // G)
    mu = xd.mean(dim=(1, 3)); var = xd.var(dim=(1, 3), unbiased=False)
    assert_close(mean, mu, f"mean ({case})", rtol=1e-6, atol_scale=1e-6)
    # rstd: eps = 1e-5 dominates the tiny-variance case
    assert_close(rstd, 1 / torch.sqrt(var + 1e-5), f"rstd ({case})", rtol=2e-5, atol_scale=2e-5)


@pytest.mark.parametrize("M", [8192, 8160])
def test_gemm_bf16_panel_large_shape(ops, M):
    """the bf16 panel kernel on a many-tile shape (512 tiles: two rounds of workgroups; also the
    L2-aware super-block tile order; launches with packed outputs run the two-workgroups-per-CU kernel from 512 tiles on -
    M = 8160 leaves its last row of tiles 96 rows tall).  bf16 x bf16 products are exact and the accumulation is fp32, so
    against an fp64 product of the bf16-ROUNDED operands the result must agree to fp32 accumulation accuracy - a
    mis-mapped tile shows as an O(1) error; the packed outputs must be the bf16 rounding of the activated output, the
    act' pack the derivative of the activation at the kernel's own pre-activation"""
    N, K = 2048, 256                               # 64 x 8 tiles = 512
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1.0 / math.sqrt(K)), rnd(N, seed=3, scale=0.3)
    pk = ops.PanelPacks([(W.cuda(), N, K, K, False)], bf16=True)
    pk.refresh()
    Ap = ops.bf16_pack_rows(A.cuda())
    Ab, Wb = A.bfloat16().double(), W.bfloat16().double()
    ref = Ab @ Wb.t() + b.double()
    y, pre = ops.gemm_bf16p(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), act=1, mode=ops.EPI_ACT, save_pre=True)
    assert_close(pre, ref, "pre-activation", rtol=2e-6, atol_scale=2e-6)
    assert_close(y, torch.nn.functional.gelu(ref), "output", rtol=4e-6, atol_scale=4e-6)
    y2, D, pr, pt, cs = ops.gemm_bf16p_packed(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), act=1, mode=ops.EPI_ACT,
                                              save_dact=True, pack_rows=True, pack_trans=True, colsum=True)
    assert torch.equal(y, y2)
    assert torch.equal(_unpack_rows(pr, M, N), y.bfloat16().float())
    assert torch.equal(_unpack_rows(pt, N, M), y.t().contiguous().bfloat16().float())
    assert_close(cs, y.double().sum(0), "column sums", rtol=1e-5, atol_scale=1e-5)
    p64 = pre.double().cpu().requires_grad_(True)
    torch.nn.functional.gelu(p64).sum().backward()
    assert ((_unpack_frag(D, M, N).double().cpu() - p64.grad).abs() <= 2.0 ** -8 * p64.grad.abs() + 1e-6).all()
    # the act'-product launch (fc2 data gradient form) on the same many-tile grid, without an fp32 output
    dY = rnd(M, K, seed=5)
    dYp = ops.bf16_pack_rows(dY.cuda())
    lin, _ = ops.gemm_bf16p(dYp, pk.bufs[0], M, N, K)
    want = lin * _unpack_frag(D, M, N)
    _, _, pr2, pt2, cs2 = ops.gemm_bf16p_packed(dYp, pk.bufs[0], M, N, K, act=1, mode=ops.EPI_DACT, dact=D, pack_rows=True,
                                                pack_trans=True, colsum=True, store=False)
    assert torch.equal(_unpack_rows(pr2, M, N), want.bfloat16().float())
    assert torch.equal(_unpack_rows(pt2, N, M), want.t().contiguous().bfloat16().float())
    assert_close(cs2, want.double().sum(0), "column sums of the act' product", rtol=1e-5, atol_scale=1e-5)
    res = rnd(M, N, seed=7).cuda()
    z, _ = ops.gemm_bf16p(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), res=res)
    assert_close(z, ref + res.double().cpu(), "residual epilogue", rtol=2e-6, atol_scale=2e-6)


@pytest.mark.parametrize("B,h,E,nb", [(2, 32, 384, 4), (4, 32, 1536, 16), (3, 16, 512, 4)])
def test_groupnorm_applied_on_the_load_of_its_consumer(ops, B, h, E, nb):
    """GroupNorm folded into the kernels that read its output (round 3): rfft2 / irfft2 with `norm` operands (the transform's
    input and the AFNO residual are GroupNorm1(x), never written) and the bf16 pack pass of the channel MLP's input
    (GroupNorm2(y1) never written in fp32) use the expression of the GroupNorm apply kernels - results are BIT-identical
    to the two-launch forms; statistics-only GroupNorm == the statistics of the full kernel"""
    tok = h * h
    mx, my = h, h // 2 + 1
    x = (rnd(B, tok, E, seed=1) * 1.3 + 0.2).cuda()
    g, b = (1 + 0.3 * rnd(E, seed=2)).cuda(), (0.2 * rnd(E, seed=3)).cuda()
    xn, mean, rstd = ops.groupnorm_fwd(x, g, b)
    if ops.groupnorm_stats_supported(B, tok, E):
        m2, r2 = ops.groupnorm_stats(x, g, b)
        assert torch.equal(m2, mean) and torch.equal(r2, rstd)
    assert ops.rfft2_norm_supported(h, h, E)
    S_ref = ops.rfft2(xn, h, h, nb, mx, my, 0)
    S = ops.rfft2(x, h, h, nb, mx, my, 0, norm=(mean, rstd, g, b))
    assert torch.equal(S, S_ref)
    y_ref = ops.irfft2(S_ref, B, h, h, E, nb, mx, my, 1, res=xn)
    y = ops.irfft2(S_ref, B, h, h, E, nb, mx, my, 1, res=x, res_norm=(mean, rstd, g, b))
    assert torch.equal(y, y_ref)
    M = B * tok
    if ops.bf16_pack_both_supported(M, E) and tok % 64 == 0:
        pr0, pt0, _ = ops.bf16_pack_both(xn.view(M, E))
        pr1, pt1, _ = ops.bf16_pack_both(x.view(M, E), norm=(mean, rstd, g, b, tok))
        assert torch.equal(pr0, pr1) and torch.equal(pt0, pt1)
    if ops.gn_dft_supported(h, h, E):
        g2, b2 = (1 + 0.3 * rnd(E, seed=4)).cuda(), (0.2 * rnd(E, seed=5)).cuda()
        y1, xn2, mm, rr = ops.irfft2_gn(S_ref, x, mean, rstd, g, b, g2, b2, h, h, nb, mx, my)
        y1b, none, mmb, rrb = ops.irfft2_gn(S_ref, x, mean, rstd, g, b, g2, b2, h, h, nb, mx, my, want_xn2=False)
        assert none is None and torch.equal(y1, y1b) and torch.equal(mm, mmb) and torch.equal(rr, rrb)
        pr0, pt0, _ = ops.bf16_pack_both(xn2.view(M, E))
        pr1, pt1, _ = ops.bf16_pack_both(y1.view(M, E), norm=(mm, rr, g2, b2, tok))
        assert torch.equal(pr0, pr1) and torch.equal(pt0, pt1)


def test_gemm_bf16_panel_192_wide_tiles(ops):
    """DPOT-L at batch 4: tokens 4096, E = 1536 -> fc2 forward / fc1 data gradient have 32 x 6 = 192 tiles of 128 x 256, so
    the library runs them as 32 x 8 = 256 tiles of 128 x 192 (2 x 3 compute waves, 20 KiB slabs).  Same checks as the
    256-wide kernel: fp64 product of the bf16-rounded operands, residual epilogue, packed outputs, act' pack"""
    M, N, K = 4096, 1536, 512
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1.0 / math.sqrt(K)), rnd(N, seed=3, scale=0.3)
    pk = ops.PanelPacks([(W.cuda(), N, K, K, False)], bf16=True)
    pk.refresh()
    Ap = ops.bf16_pack_rows(A.cuda())
    ref = A.bfloat16().double() @ W.bfloat16().double().t() + b.double()
    res = rnd(M, N, seed=7).cuda()
    z, _ = ops.gemm_bf16p(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), res=res)
    assert_close(z, ref + res.double().cpu(), "residual epilogue", rtol=2e-6, atol_scale=2e-6)
    y, pre = ops.gemm_bf16p(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), act=1, mode=ops.EPI_ACT, save_pre=True)
    assert_close(pre, ref, "pre-activation", rtol=2e-6, atol_scale=2e-6)
    y2, D, pr, pt, cs = ops.gemm_bf16p_packed(Ap, pk.bufs[0], M, N, K, bias=b.cuda(), act=1, mode=ops.EPI_ACT,
                                              save_dact=True, pack_rows=True, pack_trans=True, colsum=True)
    assert torch.equal(y, y2)
    assert torch.equal(_unpack_rows(pr, M, N), y.bfloat16().float())
    assert torch.equal(_unpack_rows(pt, N, M), y.t().contiguous().bfloat16().float())
    assert_close(cs, y.double().sum(0), "column sums", rtol=1e-5, atol_scale=1e-5)
    p64 = pre.double().cpu().requires_grad_(True)
    torch.nn.functional.gelu(p64).sum().backward()
    assert ((_unpack_frag(D, M, N).double().cpu() - p64.grad).abs() <= 2.0 ** -8 * p64.grad.abs() + 1e-6).all()
    # weight-gradient form with split-K on the same column count (K = tokens)
    dy, x = rnd(M, 256, seed=8), rnd(M, N, seed=9)
    dyT, xT = ops.bf16_pack_rows(dy.cuda(), trans=True), ops.bf16_pack_rows(x.cuda(), trans=True)
    g, _ = ops.gemm_bf16p(dyT, xT, 256, N, M)
    assert_close(g, dy.bfloat16().double().t() @ x.bfloat16().double(), "weight-gradient form", rtol=4e-6, atol_scale=4e-6)


@pytest.mark.parametrize("E,nb,modes", [(512, 4, 32), (1024, 8, 32), (512, 4, 5)])
def test_groupnorm_dft_fused_kernels_vs_separate(ops, E, nb, modes):
    """csrc/gn_dft.hip (round 3): each GroupNorm + DFT pair as ONE kernel against the two separate kernels it
    replaces (which are pinned by the golden / oracle tests): forward pair (norm1 + rfft2, irfft2 + x_orig + norm2) and
    backward pair (norm2 backward + adjoint rfft2, adjoint irfft2 + skip + norm1 backward + outer skip); 64 and 128
    channels per group, full and truncated mode sets.  fp32 re-association only: rtol 2e-5 of the tensor scale."""
    B, h = 3, 16
    if not ops.gn_dft_supported(h, h, E):
        pytest.skip("fused GroupNorm-DFT kernels switched off (DPOT_TUNE gn_fuse=0): the separate kernels run")
    mx, my = min(modes, h), min(modes, h // 2 + 1)
    x = (rnd(B, h * h, E, seed=1) * 1.7 + 0.4).cuda()
    g1, b1 = (1 + 0.3 * rnd(E, seed=2)).cuda(), (0.2 * rnd(E, seed=3)).cuda()
    g2, b2 = (1 + 0.3 * rnd(E, seed=4)).cuda(), (0.2 * rnd(E, seed=5)).cuda()
    tol = dict(rtol=2e-5, atol_scale=2e-5)
    # K1: norm1 + rfft2
    xn1, m1, r1 = ops.groupnorm_fwd(x, g1, b1)
    S_ref = ops.rfft2(xn1, h, h, nb, mx, my, 0)
    S, m1f, r1f = ops.gn_rfft2(x, g1, b1, h, h, nb, mx, my)
    assert_close(m1f, m1, "mean1", **tol)
    assert_close(r1f, r1, "rstd1", **tol)
    assert_close(S, S_ref, "gn_rfft2 spectrum", **tol)
    # K2: irfft2 + x_orig + norm2
    O2 = (rnd(B * mx * my, 2 * E, seed=6) * 0.8).cuda()
    y1_ref = ops.irfft2(O2, B, h, h, E, nb, mx, my, 1, res=xn1)
    xn2_ref, m2, r2 = ops.groupnorm_fwd(y1_ref, g2, b2)
    y1, xn2, m2f, r2f = ops.irfft2_gn(O2, x, m1, r1, g1, b1, g2, b2, h, h, nb, mx, my)
    assert_close(y1, y1_ref, "irfft2_gn y1", **tol)
    assert_close(xn2, xn2_ref, "irfft2_gn xn2", **tol)
    assert_close(m2f, m2, "mean2", **tol)
    assert_close(r2f, r2, "rstd2", **tol)
    # K3: norm2 backward + rfft2 with the adjoint column weights
    dxn2 = rnd(B, h * h, E, seed=7).cuda()
    dy1_ref, part2_ref = ops.groupnorm_bwd(dxn2, y1_ref, m2, r2, g2, defer=True)
    dO2_ref = ops.rfft2(dy1_ref, h, h, nb, mx, my, 1)
    dy1, part2, dO2 = ops.gn_bwd_rfft2(dxn2, y1_ref, m2, r2, g2, h, h, nb, mx, my, col_weights=1)
    assert_close(dy1, dy1_ref, "gn_bwd_rfft2 dx", **tol)
    assert_close(part2, part2_ref, "gn_bwd_rfft2 partials", **tol)
    assert_close(dO2, dO2_ref, "gn_bwd_rfft2 spectrum", **tol)
    # K4: adjoint irfft2 + skip + norm1 backward + outer skip
    dS = (rnd(B * mx * my, 2 * E, seed=8) * 0.8).cuda()
    dout = rnd(B, h * h, E, seed=9).cuda()
    dxn1_ref = ops.irfft2(dS, B, h, h, E, nb, mx, my, 0, res=dy1_ref)
    dx_ref, part1_ref = ops.groupnorm_bwd(dxn1_ref, x, m1, r1, g1, add=dout, defer=True)
    dx, part1 = ops.irfft2_gn_bwd(dS, dy1_ref, x, m1, r1, g1, h, h, nb, mx, my, add=dout, col_weights=0)
    assert_close(dx, dx_ref, "irfft2_gn_bwd dx", **tol)
    assert_close(part1, part1_ref, "irfft2_gn_bwd partials", **tol)
    dx0, _ = ops.irfft2_gn_bwd(dS, dy1_ref, x, m1, r1, g1, h, h, nb, mx, my, add=None, col_weights=0)
    assert_close(dx0, dx_ref - dout, "irfft2_gn_bwd without the outer skip", **tol)


def test_groupnorm_rfft2_fused_with_a_large_group_mean(ops):
    """gn_rfft2 transforms x itself and applies GroupNorm to the spectrum; with |mean| >> std the FFT's
    round-off would scale with |mean| (log10(|mean|/std) digits lost in every bin).  The kernel transforms x - x[0]
    per channel instead.  mean / std = 100 and 1000, against float64 GroupNorm -> rfft2: the fused kernel must be as
    accurate as the separate GroupNorm-then-rfft2 kernels (whose input to the FFT is already normalised)"""
    B, h, E, nb = 2, 16, 512, 4
    if not ops.gn_dft_supported(h, h, E):
        pytest.skip("fused GroupNorm-DFT kernels switched off")
    g1, b1 = (1 + 0.3 * rnd(E, seed=2)).cuda(), (0.2 * rnd(E, seed=3)).cuda()
    for ratio in (1e2, 1e3):
        x = (rnd(B, h * h, E, seed=1) + ratio).cuda()
        xd = x.double().view(B, h * h, 8, E // 8)
        mu = xd.mean(dim=(1, 3), keepdim=True)
        var = xd.var(dim=(1, 3), unbiased=False, keepdim=True)
        xn = ((xd - mu) / torch.sqrt(var + 1e-5)).view(B, h, h, E) * g1.double() + b1.double()
        F = torch.fft.rfft2(xn, dim=(1, 2), norm="ortho")                                     # [B, h, wf, E]
        ref = torch.stack([F.real, F.imag], dim=-2).view(B, h, 9, 2, nb, E // nb).permute(0, 1, 2, 4, 3, 5)
        ref = ref.reshape(B * h * 9, 2 * E)
        xn1, _, _ = ops.groupnorm_fwd(x, g1, b1)
        e_sep = ((ops.rfft2(xn1, h, h, nb, h, 9, 0).double() - ref).norm() / ref.norm()).item()
        S, _, _ = ops.gn_rfft2(x, g1, b1, h, h, nb, h, 9)
        e_fused = ((S.double() - ref).norm() / ref.norm()).item()
        print(f"[gn_rfft2, mean/std = {ratio:g}] norm-wise error vs float64: fused {e_fused:.2e}, separate kernels {e_sep:.2e}")
        # the input's own representation error is eps * ratio (a float32 x cannot carry more): both paths sit there
        assert e_fused <= max(2.0 * e_sep, 3e-7 * ratio)


@pytest.mark.parametrize("E,nb,B,norm,save,act", [(512, 4, 3, True, True, "gelu"), (1024, 8, 2, True, True, "gelu"),
                                                  (512, 4, 2, False, True, "gelu"), (1024, 8, 3, True, False, "silu"),
                                                  (512, 4, 1, True, False, "gelu")])
def test_afno_layer_one_launch_vs_three_launches(ops, monkeypatch, E, nb, B, norm, save, act):
    """csrc/afno_fused.hip (round 5): norm1 -> rfft2 -> both MLP layers -> irfft2 + x_orig -> norm2 in ONE launch
    (register FFTs, spectrum / hidden layer as the MFMA operand in LDS) against (a) the three launches it replaces -
    gn_rfft2, afno_mlp2 (three-product kernel), irfft2_gn, themselves pinned by the golden / oracle tests - on every tensor
    both produce (S, layer-1 pre-activation, y1, xn2, both statistics) and (b) float64 torch.fft of the same layer
    (models/dpot.py:59-102 + GroupNorm).  64 / 128 channels per group, the norm-free form (the reference's AFNO2D module
    alone) and the inference form (S / pre not written)."""
    set_tune(monkeypatch, afno_layer=1)
    h, G = 16, 8
    mx, my, bs = 16, 9, E // nb
    if not ops.afno_fused_supported(h, h, E, nb, mx, my, G=G if norm else 0):
        pytest.skip("one-launch AFNO layer not available")
    a = ops.ACT_IDS[act]
    x = (rnd(B, h * h, E, seed=1) * 1.3 + 0.25).cuda()
    g1, b1 = (1 + 0.3 * rnd(E, seed=2)).cuda(), (0.2 * rnd(E, seed=3)).cuda()
    g2, b2 = (1 + 0.3 * rnd(E, seed=4)).cuda(), (0.2 * rnd(E, seed=5)).cuda()
    w1, w2 = (rnd(2, nb, bs, bs, seed=6) * 0.09).cuda(), (rnd(2, nb, bs, bs, seed=7) * 0.09).cuda()
    bb1, bb2 = (rnd(2, nb, bs, seed=8) * 0.1).cuda(), (rnd(2, nb, bs, seed=9) * 0.1).cuda()
    packed = ops.AfnoPacks([(w1, bb1), (w2, bb2)]).refresh()
    assert getattr(packed[0], "layout", 0) == 1
    n = (g1, b1, g2, b2) if norm else (None, None, None, None)
    S, pre, y1, xn2, m1, r1, m2, r2 = ops.afno_fused_fwd(x, n[0], n[1], packed[0][2], packed[0][1], packed[1][2],
                                                         packed[1][1], n[2], n[3], h, h, nb, mx, my, a, save=save)
    tol = dict(rtol=2e-5, atol_scale=2e-5)
    # (a) the three launches
    if norm:
        S3, m1_3, r1_3 = ops.gn_rfft2(x, g1, b1, h, h, nb, mx, my)
    else:
        S3 = ops.rfft2(x, h, h, nb, mx, my, 0)
    O2, pre3, _ = ops.afno_mlp2(S3, packed[0][2], packed[0][1], packed[1][2], packed[1][1], nb, bs, a, mode=0,
                                want_pre=True, layout=1)
    if norm:
        y1_3, xn2_3, m2_3, r2_3 = ops.irfft2_gn(O2, x, m1_3, r1_3, g1, b1, g2, b2, h, h, nb, mx, my)
    else:
        y1_3 = ops.irfft2(O2, B, h, h, E, nb, mx, my, 1, res=x)
    if save:
        assert_close(S, S3, "spectrum", **tol)
        assert_close(pre, pre3, "layer-1 pre-activation", **tol)
    else:
        assert S is None and pre is None
    assert_close(y1, y1_3, "y1", **tol)
    if norm:
        assert_close(xn2, xn2_3, "xn2", **tol)
        for t, t3, nm in ((m1, m1_3, "mean1"), (r1, r1_3, "rstd1"), (m2, m2_3, "mean2"), (r2, r2_3, "rstd2")):
            assert_close(t, t3, nm, **tol)
    else:
        assert xn2 is None and m1 is None and m2 is None
    # (b) float64 reference of the layer
    xd = x.double().cpu()

    def gn(t, g, b):
        td = t.view(B, h * h, G, E // G)
        mu, var = td.mean(dim=(1, 3), keepdim=True), td.var(dim=(1, 3), unbiased=False, keepdim=True)
        return ((td - mu) / torch.sqrt(var + 1e-5)).view(B, h * h, E) * g.double().cpu() + b.double().cpu()

    xn = gn(xd, g1, b1) if norm else xd
    F = torch.fft.rfft2(xn.view(B, h, h, E), dim=(1, 2), norm="ortho").view(B, h, 9, nb, bs)
    W1 = torch.complex(w1[0].double().cpu(), w1[1].double().cpu())
    W2 = torch.complex(w2[0].double().cpu(), w2[1].double().cpu())
    c1 = torch.complex(bb1[0].double().cpu(), bb1[1].double().cpu())
    c2 = torch.complex(bb2[0].double().cpu(), bb2[1].double().cpu())
    o1 = torch.einsum("bxykI,kIO->bxykO", F, W1) + c1
    f = ACTS[act]
    o1 = torch.complex(f(o1.real), f(o1.imag))
    o2 = torch.einsum("bxykI,kIO->bxykO", o1, W2) + c2
    yref = torch.fft.irfft2(o2.reshape(B, h, 9, E), s=(h, h), dim=(1, 2), norm="ortho").reshape(B, h * h, E) + xn
    assert_close(y1, yref, "y1 vs float64")
    if norm:
        assert_close(xn2, gn(yref, g2, b2), "xn2 vs float64")
        # the same launch writing GroupNorm2(y1) as the channel MLP's two bf16 operand packs == the separate pack pass over y1
        res = ops.afno_fused_fwd(x, g1, b1, packed[0][2], packed[0][1], packed[1][2], packed[1][1], g2, b2, h, h, nb, mx, my, a,
                                 save=save, want_xn2=False, want_packs=True)
        # (a separate instantiation of the kernel: same arithmetic, the compiler's FMA contraction may differ in the last bit)
        assert_close(res[2], y1, "y1 of the pack-emitting instantiation", rtol=1e-6, atol_scale=1e-6)
        assert res[3] is None
        xp3, xpT3, _ = ops.bf16_pack_both(res[2].view(B * h * h, E), norm=(res[6], res[7], g2, b2, h * h))
        assert torch.equal(res[8].view(torch.int16), xp3.view(torch.int16)), "row-form pack of GroupNorm2(y1)"
        assert torch.equal(res[9].view(torch.int16), xpT3.view(torch.int16)), "transposed pack of GroupNorm2(y1)"


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,E,add", [(3, 256, 1024, True), (2, 128, 1024, False), (1, 256, 2048, True),
                                       (4, 1024, 1536, True), (16, 1024, 1536, False), (2, 512, 768, True)])
def test_groupnorm_bwd_writes_the_gradient_packs(ops, B, T, E, add):
    """round 5: the GroupNorm backward kernel that produces a Block's input gradient also writes it as the bf16 operands of the
    previous Block's channel-MLP backward (csrc/norm.hip groupnorm_bwd_cached_kernel<.., PK>; DPOT-L's chunked slabs:
    gn_chunk_bwd_apply_pk_kernel): dx and the parameter-gradient partials bit-identical to the plain kernels, the packs and
    column sums equal to a separate bf16_pack_both pass over dx"""
    G = 8
    rows = ops.groupnorm_bwd_packs_rows(B, T, E, G)
    if rows == 0:
        pytest.skip("pack-emitting GroupNorm backward not available for this shape")
    torch.manual_seed(B + T + E)
    x = torch.randn(B, T, E, device="cuda") * 1.5 + 0.3
    dy = torch.randn(B, T, E, device="cuda")
    gw, gb = torch.randn(E, device="cuda"), torch.randn(E, device="cuda")
    res = torch.randn(B, T, E, device="cuda") if add else None
    _, mean, rstd = ops.groupnorm_fwd(x, gw, gb, G)
    dx0, part0 = ops.groupnorm_bwd(dy, x, mean, rstd, gw, G, add=res, defer=True)
    dx, part, pr, pt, cs = ops.groupnorm_bwd_packs(dy, x, mean, rstd, gw, G, add=res)
    assert torch.equal(dx, dx0) and torch.equal(part, part0)
    assert cs.shape == (B * rows, E)
    pr0, pt0, cs0 = ops.bf16_pack_both(dx0.view(B * T, E), want_colsum=True)
    assert torch.equal(pr.view(torch.int16), pr0.view(torch.int16)), "row-form pack of dx"
    assert torch.equal(pt.view(torch.int16), pt0.view(torch.int16)), "transposed pack of dx"
    assert_close(cs.sum(0), cs0, "column sums of dx", rtol=1e-5, atol_scale=1e-5)
    assert_close(cs.view(B, rows, E).sum(1), dx0.double().sum(1), "per-sample column sums of dx", rtol=1e-5, atol_scale=1e-5)
