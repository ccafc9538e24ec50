"""Autograd functions of the DPOT step, one per stage, each with a hand-written backward that calls the HIP
kernels directly (no autograd tracing inside a stage, no eager-PyTorch math):

  EmbedFn   patch embed (8x8/8 conv as GEMM + act) -> folded [1x1 conv + pos + TimeAggregator] GEMM  models/dpot.py:373-384
  BlockFn   GroupNorm -> AFNO mixer -> GroupNorm -> channel MLP -> +residual                        models/dpot.py:165-180
  HeadFn    out_layer (ConvTranspose as GEMM + pixel tail) and cls_head                             models/dpot.py:394-398
  RelL2Fn   masked relative L2 loss                                                                 utils/criterion.py:38-59

Internal activation layout: channels-last tokens [B, h*w, E] (row-major [B*h*w, E]).

Parameter gradients: when the parameters live in a train.FlatParams buffer, the backward kernels write them straight
into their slot of the flat gradient buffer (see _Sink) and autograd receives None - no per-parameter accumulate
kernels, and the data-parallel reducer is notified the moment a parameter's gradient is final.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Optional

import torch

from . import ops
from .ops import EPI_ACT, EPI_DACT

Tensor = torch.Tensor


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class _Sink:
    """Where the gradient of one parameter goes.

    forward:  ``_Sink(param, needed)`` registers one pending use if ``param`` is bound to a FlatParams slot.
    backward: ``out()`` is the buffer the producing kernel should write (the slot itself for the first contribution of
              a step, None = "allocate a temporary" otherwise); ``done(t)`` finalises: returns ``t`` for plain
              autograd, or adds/marks the slot, notifies the owner when the last pending use has delivered, and
              returns None (autograd then has nothing to accumulate)."""
    __slots__ = ("fp", "i")

    def __init__(self, param: Optional[Tensor], needed: bool):
        s = getattr(param, "_dpot_grad_slot", None) if (needed and param is not None) else None
        self.fp, self.i = s if s is not None else (None, -1)
        if self.fp is not None:
            self.fp.pending[self.i] += 1

    def out(self) -> Optional[Tensor]:
        if self.fp is None or self.fp.filled[self.i]:
            return None
        return self.fp.grad_views[self.i]

    def done(self, t: Optional[Tensor]) -> Optional[Tensor]:
        if self.fp is None or t is None:
            return t
        fp, i = self.fp, self.i
        view = fp.grad_views[i]
        if t.data_ptr() != view.data_ptr():
            view.add_(t.reshape(view.shape))          # produced elsewhere (or a further use in this step): accumulate
        fp.filled[i] = True
        fp.pending[i] -= 1
        if fp.pending[i] == 0:
            fp.fire(i)
        return None

    def skip(self) -> None:
        """this use delivers no gradient (its branch of the backward did not run): release the pending count so the
        parameter can still be reported final (data-parallel bucket accounting)"""
        if self.fp is None:
            return
        fp, i = self.fp, self.i
        fp.pending[i] -= 1
        if fp.pending[i] == 0 and fp.filled[i]:
            fp.fire(i)


def _epoch_of(sinks):
    """(FlatParams, its weights epoch) of the first flat-bound parameter, or None.  DPOTNet keeps its derived weights
    (AFNO packs, panel packs, the de-embed matrix) in PERSISTENT buffers that every forward outside a weights_scope
    rewrites in place, and the stage backwards read them again without autograd version tracking; train.FusedAdam bumps
    the epoch when it changes the parameters.  (Parameters changed by other means are the caller's responsibility:
    run backward before modifying them.)"""
    for sk in sinks:
        if sk.fp is not None:
            return sk.fp, sk.fp.epoch
    return None


def _check_epoch(ctx, what: str) -> None:
    rec = ctx.weights_epoch
    if rec is not None and rec[0].epoch != rec[1]:
        raise RuntimeError(f"{what}.backward: the parameters were updated (optimiser step) after this forward ran; its "
                           "packed weights have been overwritten.  Run backward before the step, or re-run the forward.")


def _sinks(ctx, params, first_index: int):
    return [_Sink(p, ctx.needs_input_grad[first_index + k]) for k, p in enumerate(params)]


# ======================================================================================================
def embed_grid_matrix(gx, gy, gt, X: int, Y: int, T: int, Cc: int, P: int):
    """[tok*T, 3*P*P]: the unit-grid columns of the patch matrix (input channels C..C+2 = x, y, t coordinates,
    models/dpot.py:356-357) - batch independent, constant for a model"""
    z = torch.zeros(1, X, Y, T, Cc, dtype=torch.float32, device=gx.device)
    return ops.patchify(z, gx, gy, gt, P)[:, Cc * P * P:].contiguous()


def embed_layout_jobs(pos, w0, b0, w2, b2):
    """the layout pieces of embed_derived as ops.LayoutJobs entries, in the order (w0p, b0p, w2p, posb)"""
    hid, E = w0.shape[0], w2.shape[0]
    K0 = w0[0].numel()
    hidp = _pad4(hid)
    tok = pos.shape[2] * pos.shape[3]
    return [(w0, None, (1, hidp, K0), (1, hid, K0), (0, K0, 1)),               # zero rows hid..hidp
            (b0, None, (1, 1, hidp), (1, 1, hid), (0, 0, 1)),
            (w2, None, (1, E, hidp), (1, E, hid), (0, hid, 1)),                # zero cols hid..hidp
            (pos, b2, (1, tok, E), (1, tok, E), (0, 1, tok))]                  # pos^T [tok, E] + conv bias


def embed_derived(pos, w0, b0, w2, b2, taw, tagamma, tt, T: int, grid=None, layouts=None):
    """weight-only products of the embed stage (they depend on no activation, so a T_ar-step rollout computes them
    once per optimiser step, see DPOTNet.weights_scope): padded conv weights, pos+bias, the cos-scaled aggregation
    weights and the folded [1x1 conv -> TimeAggregator] matrices V / c described in EmbedFn.forward.  layouts: the four
    tensors of `embed_layout_jobs`, already refreshed (DPOTNet: one launch for all layout pieces of the model)"""
    hid, E = w0.shape[0], w2.shape[0]
    K0 = w0[0].numel()
    hidp = _pad4(hid)
    tok = pos.shape[2] * pos.shape[3]
    dev = pos.device
    if layouts is not None:
        w0p, b0p, w2p, posb = layouts[0].view(hidp, K0), layouts[1].view(hidp), layouts[2].view(E, hidp), layouts[3].view(tok, E)
    else:
        w0p = ops.copy2d_pad(w0, hid, K0, hidp, K0)                                # zero rows hid..hidp
        b0p = ops.copy2d_pad(b0, 1, hid, 1, hidp).view(hidp)
        w2p = ops.copy2d_pad(w2, E, hid, E, hidp)                                  # zero cols hid..hidp
        posT = ops.transpose2d(pos, 1, E, tok).view(tok, E)                        # [tok, E]
        posb = ops.bias_add(posT, b2)                                              # pos + conv bias
    ws = ops.timeagg_scale_w(taw, tagamma, tt) if tagamma is not None else taw
    V = torch.empty(T * hidp, E, dtype=torch.float32, device=dev)
    ops.gemm(w2p, ws, V, hidp, E, E, transA=True, lda=hidp, ldb=E, ldc=E, batch=T, strideA=0, strideB=E * E,
             strideC=hidp * E)
    wsum = ops.colsum(ws, T, E * E).view(E, E)
    cc = torch.empty(tok, E, dtype=torch.float32, device=dev)
    ops.gemm(posb, wsum, cc, tok, E, E, lda=E, ldb=E, ldc=E)
    wfrag = bt = None
    if grid is not None:
        # implicit-GEMM patch embedding (csrc/embed.hip): data-channel weights in fragment order, and the unit-grid
        # channels folded into a bias table  bt[(tok,t), n] = b0[n] + sum_{c>=C,i,j} grid[(tok,t),(c,i,j)] w0[n,c,i,j]
        kg = grid.shape[1]
        wfrag = ops.embed_pack_w0(w0)
        bt = torch.empty(grid.shape[0], hidp, dtype=torch.float32, device=dev)
        ops.gemm(grid, w0p[:, K0 - kg:], bt, grid.shape[0], hidp, kg, transB=True, lda=kg, ldb=K0, ldc=hidp, bias=b0p)
    return w0p, b0p, w2p, posb, ws, V, wsum, cc, wfrag, bt, grid


class EmbedFn(torch.autograd.Function):
    """x[B,X,Y,T,C] -> latent [B, h*w, E]"""

    @staticmethod
    def forward(ctx, x, pos, w0, b0, w2, b2, taw, tagamma, gx, gy, gt, tt, P: int, act: int, derived=None):
        ops.capture_precision(ctx)       # re-applied around backward (ops.with_ctx_precision)
        x = x.contiguous()
        B, X, Y, T, Cc = x.shape
        h, w = X // P, Y // P
        tok = h * w
        hid, E = w0.shape[0], w2.shape[0]
        K0 = (Cc + 3) * P * P
        hidp = _pad4(hid)
        dev = x.device
        M = B * tok

        implicit = derived is not None and derived[8] is not None
        A0 = None if implicit else ops.patchify(x, gx, gy, gt, P)               # [M0, K0], rows (b,px,py,t)
        if derived is None:
            derived = embed_derived(pos, w0, b0, w2, b2, taw, tagamma, tt, T)
        w0p, b0p, w2p, posb, ws, V, wsum, cc, wfrag, bt, grid = derived
        if implicit:   # rows gathered from x inside the kernel; grid channels + bias come from the table bt
            Hh, Hpre = ops.embed_fwd(x, wfrag, bt, hidp, act)
        else:
            Hh, Hpre = ops.linear_fwd(A0, w0p, b0p, act=act, save_pre=True)    # [M0, hidp]
        # Everything after the activation is LINEAR (1x1 conv hid->E, + pos_embed, TimeAggregator), and the hidden
        # width is only hid = out_channels*P+3 (35).  Instead of materialising z[M0, E] (168 MB at B=32) and
        # contracting it with ws[T*E, E] (K = 5120), the 1x1 conv is folded INTO the aggregation weights every step:
        #     V[t,h,j]  = sum_i w2[i,h] * ws[t,i,j]                         (T small GEMMs, 0.2 GFLOP)
        #     c[tok,j]  = sum_i (pos[tok,i] + b2[i]) * sum_t ws[t,i,j]       (token-dependent constant)
        #     y[m,j]    = sum_{t,h} H[m,(t,h)] * V[(t,h),j] + c[tok(m),j]    (K = T*hidp = 360 instead of 5120)
        # Same result up to fp32 re-association; 15x fewer FLOPs for this stage, and its backward.
        Yl = torch.empty(M, E, dtype=torch.float32, device=dev)
        ops.gemm(Hh, V, Yl, M, E, T * hidp, lda=T * hidp, ldb=E, ldc=E, res=cc, ldres=E, res_mod=tok)
        ctx.implicit = implicit
        ctx.save_for_backward(x if implicit else A0, Hpre, Hh, w0p, w2p, ws, V, wsum, posb, taw, tagamma, tt,
                              grid if implicit else x.new_empty(0))
        ctx.dims = (B, X, Y, T, Cc, P, h, w, hid, hidp, E, K0, act)
        ctx.sinks = _sinks(ctx, (pos, w0, b0, w2, b2, taw, tagamma), 1)
        ctx.weights_epoch = _epoch_of(ctx.sinks)
        return Yl.view(B, tok, E)

    @staticmethod
    @ops.with_ctx_precision
    def backward(ctx, dY):
        _check_epoch(ctx, "EmbedFn")
        A0, Hpre, Hh, w0p, w2p, ws, V, wsum, posb, taw, tagamma, tt, grid = ctx.saved_tensors
        B, X, Y, T, Cc, P, h, w, hid, hidp, E, K0, act = ctx.dims
        s_pos, s_w0, s_b0, s_w2, s_b2, s_taw, s_gamma = ctx.sinks
        tok = h * w
        M, M0 = B * tok, B * tok * T
        dev = dY.device
        dY = dY.contiguous().view(M, E)
        # y = H V + c  ->  dH = dY V^T (times act'), dV = H^T dY, dc = sum_b dY
        dHpre = torch.empty(M0, hidp, dtype=torch.float32, device=dev)         # viewed [M, T*hidp]
        ops.gemm(dY, V, dHpre, M, T * hidp, E, transB=True, lda=E, ldb=E, ldc=T * hidp, act=act, mode=EPI_DACT,
                 aux=Hpre, ldaux=T * hidp)
        dV = torch.empty(T * hidp, E, dtype=torch.float32, device=dev)
        ops.gemm(Hh, dY, dV, T * hidp, E, M, transA=True, lda=T * hidp, ldb=E, ldc=E,
                 splitk=ops.auto_splitk(T * hidp, E, M, tn=True))
        dc = ops.group_rowsum(dY, B, tok, 1, E)                                # [tok, E]
        # c = posb wsum  ->  dwsum = posb^T dc, dposb = dc wsum^T
        dwsum = torch.empty(E, E, dtype=torch.float32, device=dev)
        ops.gemm(posb, dc, dwsum, E, E, tok, transA=True, lda=E, ldb=E, ldc=E, splitk=ops.auto_splitk(E, E, tok, tn=True))
        dposb = torch.empty(tok, E, dtype=torch.float32, device=dev)
        ops.gemm(dc, wsum, dposb, tok, E, E, transB=True, lda=E, ldb=E, ldc=E)
        dpos = s_pos.done(ops.transpose2d(dposb, 1, tok, E, out=s_pos.out()).view(1, E, h, w))
        db2 = s_b2.done(ops.colsum(dposb, tok, E, out=s_b2.out()))
        # V_t = w2^T ws_t  ->  dw2 = sum_t ws_t dV_t^T,  dws_t = w2 dV_t (+ dwsum for every t)
        dw2_t = torch.empty(T, E, hidp, dtype=torch.float32, device=dev)
        ops.gemm(ws, dV, dw2_t, E, hidp, E, transB=True, lda=E, ldb=E, ldc=hidp, batch=T, strideA=E * E,
                 strideB=hidp * E, strideC=E * hidp)
        dw2p = ops.colsum(dw2_t, T, E * hidp).view(E, hidp)
        dw2 = s_w2.done(ops.copy2d_pad(dw2p, E, hidp, E, hid, out=s_w2.out()).view(E, hid, 1, 1))
        dws = s_taw.out() if tagamma is None else None
        dws = ops._out(dws, (T, E, E), dev)
        ops.gemm(w2p, dV, dws, E, E, hidp, lda=hidp, ldb=E, ldc=E, batch=T, strideA=0, strideB=hidp * E,
                 strideC=E * E, res=dwsum, ldres=E, strideRes=0)
        if tagamma is not None:
            dtaw, dgamma = ops.timeagg_scale_w_bwd(dws, taw, tagamma, tt, out_dw=s_taw.out(),
                                                   out_dgamma=s_gamma.out())
            dtaw, dgamma = s_taw.done(dtaw), s_gamma.done(dgamma)
        else:
            dtaw, dgamma = s_taw.done(dws), None
        # first (PxP / stride P) conv
        if ctx.implicit:
            # data channels: implicit weight-gradient GEMM over x (A0 holds x here); unit-grid channels: the grid matrix
            # is batch independent -> contract it with the batch-summed dHpre; both land in the gradient slot itself
            db0 = s_b0.done(ops.colsum(dHpre, M0, hid, ld=hidp, out=s_b0.out()))
            dw0 = ops._out(s_w0.out(), (hid, K0), dev)
            ops.embed_wgrad(A0, dHpre, dw0, hid)
            dHs = ops.group_rowsum(dHpre, B, tok * T, 1, hidp)                 # [tok*T, hidp]
            kg = grid.shape[1]
            ops.gemm(dHs, grid, dw0[:, K0 - kg:], hid, kg, tok * T, transA=True, lda=hidp, ldb=kg, ldc=K0)
            dw0 = s_w0.done(dw0.view(hid, Cc + 3, P, P))
        elif hid == hidp:
            dw0p, db0 = ops.linear_bwd_wb(dHpre, A0, None, s_b0.out())          # [hidp, K0] (padded: 16-byte loads)
            db0 = s_b0.done(db0)
        else:
            db0 = s_b0.done(ops.colsum(dHpre, M0, hid, ld=hidp, out=s_b0.out()))
            dw0p = ops.linear_bwd_weight(dHpre, A0)
        if not ctx.implicit:
            dw0 = s_w0.done(ops.copy2d_pad(dw0p, hidp, K0, hid, K0, out=s_w0.out()).view(hid, Cc + 3, P, P))
        dx = None
        if ctx.needs_input_grad[0]:
            dA0 = ops.linear_bwd_data(dHpre, w0p)                              # [M0, K0]
            dx = ops.unpatchify(dA0, B, X, Y, T, Cc, P)
        return dx, dpos, dw0, db0, dw2, db2, dtaw, dgamma, None, None, None, None, None, None, None


# ======================================================================================================
X6_MIN_FLOP = 3.0e9      # 'auto': GEMMs below this stay on the native fp32 kernels (launch / pack overheads dominate)


def mlp_pack_kind(E: int, mh: int, M: Optional[int] = None) -> Optional[str]:
    """which pre-packed-weight kernel family the channel MLP of a model uses under the current precision settings:
    'bf16' (opt-in reduced precision), 'bf16x6' (fp32-accurate operand split; precision 'bf16x6', or 'auto' for GEMMs of
    >= 3 GFLOP), 'f32' (csrc/gemm_panel.hip) or None (generic kernels)"""
    eff = ops.effective_mlp_precision()
    Mq = 128 if M is None else M          # M = None: the model deriving its packs before it has seen a batch
    bf_ok = ops.gemm_bf16p_supported(Mq, mh, E) and ops.gemm_bf16p_supported(Mq, E, mh)
    if bf_ok and eff == ops.GEMM_BF16:
        return "bf16"
    # (The fp32-accurate modes do NOT take the three-plane panel kernel: it reaches 154-190 TFLOP/s fp32-equivalent against
    # 147-178 for the register-staged split kernel (gemm_split.h) but needs a 10 B/element pack pass per activation operand that
    # costs more than it gains - profiles/r02_bf16x6p_bench.txt; the opt-in of rounds 2-5 is gone, ops.gemm_bf16p(planes=3) stays.)
    if eff == ops.GEMM_F32 and ops.panel_enabled() and ops.gemm_panel_supported(Mq, mh, E) \
            and ops.gemm_panel_supported(Mq, E, mh):
        return "f32"
    return None


def _mlp_panel_mode(mlp_pk, M, E, mh, mp) -> int:
    """which pre-packed-weight kernel the channel MLP runs on: 0 = none (generic split GEMM), 1 = fp32 panel
    (csrc/gemm_panel.hip), 2 = bf16 panel (csrc/gemm_bf16p.hip; plain bf16 or the fp32-accurate bf16x6 split, by the
    packs' kind).  The packs were made for the precision that was current when the weights were derived; a different
    precision at call time falls back to the generic kernels."""
    if mlp_pk is None:
        return 0
    kind = getattr(mlp_pk, "kind", "f32")
    if kind != mlp_pack_kind(E, mh, M):
        return 0
    if kind == "f32":
        return 1
    ok = ops.gemm_bf16p_supported(E, mh, M) and ops.gemm_bf16p_supported(mh, E, M)     # weight-gradient shapes
    return 2 if ok else 0


def _p6_of(packed, which: int):
    """the bf16x6 mixer packs (csrc/afno_mlp6.hip) of a filter's two AfnoItems - which = 0: forward operands, 1: backward-data
    operands - when the GEMM precision in effect asks for them ('auto' / 'bf16x6'), else None"""
    a, b = getattr(packed[0], "p6", None), getattr(packed[1], "p6", None)
    if a is None or b is None or a[which] is None or b[which] is None or not ops.afno_mlp6_wanted():
        return None
    return a[which], b[which]


def _mixer_core(S, packed, dims, afno_layout=None):
    """the block-diagonal complex 2-layer MLP on the kept modes (models/dpot.py:72-94): spectrum S [Mm, 2E] ->
    (O2, O1pre, O1)"""
    B, tok, E, h, w, nb, bs, mx, my, mh, act = dims
    Mm = B * mx * my
    dev = S.device
    (wb1, bb1, wb1T, _), (wb2, bb2, wb2T, _) = packed          # w*T: fragment-block-major W for the fused kernel
    if wb1T is not None:
        # both layers of the block-diagonal complex MLP in ONE launch; the activated spectrum never leaves the CU
        # except as the copy saved for the backward (csrc/afno_mlp.hip)
        # (layout 1: the (Wr, Wi) fragment packs of the three-product kernel; ops.AfnoItem carries the tag)
        lay = afno_layout if afno_layout is not None else getattr(packed[0], "layout", 0)
        p6 = _p6_of(packed, 0)
        if p6 is not None:       # fp32-accurate on the bf16 matrix cores (gemm_precision 'auto' / 'bf16x6'): same contract
            wb1T, wb2T, lay = p6[0], p6[1], 2
        # only the pre-activation is saved: the activated layer-1 output (operand of the layer-2 weight gradient) is
        # re-derived by the backward launch from it (O1 = None here; one spectrum-sized store and 18.9 MB per block less)
        O2, O1pre, O1 = ops.afno_mlp2(S, wb1T, bb1, wb2T, bb2, nb, bs, act, mode=0, want_pre=True, want_mid=False,
                                      layout=lay)
    else:
        O1 = torch.empty(Mm, 2 * E, dtype=torch.float32, device=dev)
        O1pre = torch.empty_like(O1)
        kw = dict(lda=2 * E, ldb=2 * bs, ldc=2 * E, batch=nb, strideA=2 * bs, strideB=4 * bs * bs, strideC=2 * bs,
                  strideBias=2 * bs, tag=1)
        ops.gemm(S, wb1, O1, Mm, 2 * bs, 2 * bs, bias=bb1, act=act, mode=EPI_ACT, preact=O1pre, ldpre=2 * E,
                 stridePre=2 * bs, **kw)
        O2 = torch.empty_like(O1)
        ops.gemm(O1, wb2, O2, Mm, 2 * bs, 2 * bs, bias=bb2, **kw)
    return O2, O1pre, O1


def _mixer_fwd(xn1, packed, dims, afno_layout=None, norm=None):
    """AFNO2D.forward on a [B, tok, E] field (models/dpot.py:51-110): rfft2 -> block-diagonal complex 2-layer MLP on the
    kept modes -> irfft2 + x_orig.  Returns (y1, S, O1pre, O1); shared by BlockFn and AFNO2DFn.
    norm = (mean, rstd, gamma, beta): xn1 is the UN-normalised block input and both DFT kernels apply GroupNorm1 on their
    loads (the transform's input and the residual) - the normalised field is never written"""
    B, tok, E, h, w, nb, bs, mx, my, mh, act = dims
    lay = afno_layout if afno_layout is not None else getattr(packed[0], "layout", 0)
    if norm is None and packed[0][2] is not None and ops.afno_fused_supported(h, w, E, nb, mx, my, G=0, B=B, layout=lay):
        # the whole mixer in ONE launch (csrc/afno_fused.hip), spectrum and hidden layer on chip
        S, O1pre, y1 = ops.afno_fused_fwd(xn1, None, None, packed[0][2], packed[0][1], packed[1][2], packed[1][1], None,
                                          None, h, w, nb, mx, my, act)[:3]
        return y1, S, O1pre, None
    S = ops.rfft2(xn1, h, w, nb, mx, my, 0, norm=norm)                     # [Mm, 2E]
    O2, O1pre, O1 = _mixer_core(S, packed, dims, afno_layout)
    y1 = ops.irfft2(O2, B, h, w, E, nb, mx, my, 1, res=xn1, res_norm=norm)  # + x_orig (the normalised input)
    return y1, S, O1pre, O1


def _mixer_core_bwd(dO2, S, O1pre, O1, wb1, wb2, dims, fused, afno_layout, sinks, pending=None):
    """backward of _mixer_core: dO2 [Mm, 2E] -> (dS, dw1, db1, dw2, db2); wb1 / wb2: the fragment-block-major W^T (fused
    kernel) or the plain Wbig (generic GEMM); sinks = (s_w1, s_b1, s_w2, s_b2).  pending (a dict, BlockFn): the fused
    weight-gradient launch leaves its split-K partials for the block's ONE finalising launch (pending["afno"] = the job)
    and the four gradients are returned as plain tensors - the caller calls the sinks' done() after that launch"""
    B, tok, E, h, w, nb, bs, mx, my, mh, act = dims
    s_w1, s_b1, s_w2, s_b2 = sinks
    Mm = B * mx * my
    dev = dO2.device
    kw = dict(lda=2 * E, ldb=2 * bs, ldc=2 * E, batch=nb, strideA=2 * bs, strideB=4 * bs * bs, strideC=2 * bs)
    sk = max(2, ops.auto_splitk(2 * bs, 2 * bs, Mm, nb, tn=True))
    wkw = dict(transA=True, lda=2 * E, ldb=2 * E, ldc=2 * bs, batch=nb, strideA=2 * bs, strideB=2 * bs,
               strideC=4 * bs * bs, splitk=sk, mode=ops.EPI_AFNO_WGRAD)
    # both weight gradients of the mixer in ONE launch (csrc/gemm_tn.hip, dpot_afno_wgrad2) once dO1pre exists
    sk2 = ops.afno_wgrad2_splitk(Mm, nb, bs) if fused and S.stride(0) == 2 * E else 0
    if fused:
        # data path of both layers in one launch: dO1pre = (dO2 W2^T) * act'(O1pre), dS = dO1pre W1^T
        # (wb1 / wb2 hold the fragment-block-major W^T here); the same launch re-derives O1 = act(O1pre) when the forward
        # did not keep it
        dS, O1r, dO1pre = ops.afno_mlp2(dO2, wb2, None, wb1, None, nb, bs, act, mode=1, aux=O1pre, want_mid=True,
                                        want_pre=O1 is None, layout=afno_layout)
        if O1 is None:
            O1 = O1r
    if not sk2:
        # wgrad of Wbig; its split-K reduction writes dw / db in the parameters' own [2, nb, bs, ...] layout
        dw2, db2 = ops._out(s_w2.out(), (2, nb, bs, bs), dev), ops._out(s_b2.out(), (2, nb, bs), dev)
        ops.gemm(O1, dO2, dw2, 2 * bs, 2 * bs, Mm, colsum_out=db2, colsum_of=2, **wkw)
        dw2, db2 = s_w2.done(dw2), s_b2.done(db2)
    if not fused:
        dO1pre = torch.empty(Mm, 2 * E, dtype=torch.float32, device=dev)
        ops.gemm(dO2, wb2, dO1pre, Mm, 2 * bs, 2 * bs, transB=True, act=act, mode=EPI_DACT, aux=O1pre,
                 ldaux=2 * E, strideAux=2 * bs, **kw)
    if sk2:
        dw1, db1 = ops._out(s_w1.out(), (2, nb, bs, bs), dev), ops._out(s_b1.out(), (2, nb, bs), dev)
        dw2, db2 = ops._out(s_w2.out(), (2, nb, bs, bs), dev), ops._out(s_b2.out(), (2, nb, bs), dev)
        if pending is not None:
            pending["afno"] = ops.afno_wgrad2(S, dO1pre, O1, dO2, nb, bs, dw1, db1, dw2, db2, sk2, defer=True)
        else:
            ops.afno_wgrad2(S, dO1pre, O1, dO2, nb, bs, dw1, db1, dw2, db2, sk2)
            dw1, db1, dw2, db2 = s_w1.done(dw1), s_b1.done(db1), s_w2.done(dw2), s_b2.done(db2)
    else:
        dw1, db1 = ops._out(s_w1.out(), (2, nb, bs, bs), dev), ops._out(s_b1.out(), (2, nb, bs), dev)
        ops.gemm(S, dO1pre, dw1, 2 * bs, 2 * bs, Mm, colsum_out=db1, colsum_of=2, **wkw)
        dw1, db1 = s_w1.done(dw1), s_b1.done(db1)
    if not fused:
        dS = torch.empty(Mm, 2 * E, dtype=torch.float32, device=dev)
        ops.gemm(dO1pre, wb1, dS, Mm, 2 * bs, 2 * bs, transB=True, **kw)
    return dS, dw1, db1, dw2, db2


def _mixer_bwd(dy1, S, O1pre, O1, wb1, wb2, dims, fused, afno_layout, sinks, pending=None):
    """backward of _mixer_fwd: returns (dxn1 = adjoint-rfft2(dS) + dy1, dw1, db1, dw2, db2); pending: see _mixer_core_bwd"""
    B, tok, E, h, w, nb, bs, mx, my, mh, act = dims
    dO2 = ops.rfft2(dy1, h, w, nb, mx, my, 1)                              # adjoint of irfft2
    dS, dw1, db1, dw2, db2 = _mixer_core_bwd(dO2, S, O1pre, O1, wb1, wb2, dims, fused, afno_layout, sinks, pending)
    dxn1 = ops.irfft2(dS, B, h, w, E, nb, mx, my, 0, res=dy1)              # adjoint of rfft2, + skip path
    return dxn1, dw1, db1, dw2, db2


class AFNO2DFn(torch.autograd.Function):
    """The AFNO mixer alone, as the reference's ``AFNO2D`` module computes it (models/dpot.py:51-110) on a channels-last
    field: x[B, h*w, E] -> irfft2(MLP(rfft2(x))) + x.  The very kernel sequence BlockFn runs (same helpers): the per-op
    golden vectors g1_afno_* drive the product mixer through this entry."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, h: int, w: int, nb: int, modes: int, act: int, packed=None):
        ops.capture_precision(ctx)       # re-applied around backward (ops.with_ctx_precision)
        x = x.contiguous()
        B, tok, E = x.shape
        bs = E // nb
        mx, my = min(modes, h), min(modes, w // 2 + 1)
        if packed is None:
            packed = tuple(ops.AfnoPacks([(w1, b1), (w2, b2)]).refresh())
        dims = (B, tok, E, h, w, nb, bs, mx, my, 0, act)
        y1, S, O1pre, O1 = _mixer_fwd(x, packed, dims)
        ctx.fused = packed[0][2] is not None
        wb1, wb2 = (packed[0][3], packed[1][3]) if ctx.fused else (packed[0][0], packed[1][0])
        ctx.afno_layout = getattr(packed[0], "layout", 0) if ctx.fused else 0
        ctx.p6b = _p6_of(packed, 1) if ctx.fused else None     # persistent buffers of the pack object (epoch-checked)
        ctx.save_for_backward(S, O1pre, O1, wb1, wb2)
        ctx.dims = dims
        ctx.sinks = _sinks(ctx, (w1, b1, w2, b2), 1)
        ctx.weights_epoch = _epoch_of(ctx.sinks)
        return y1

    @staticmethod
    @ops.with_ctx_precision
    def backward(ctx, dy):
        _check_epoch(ctx, "AFNO2DFn")
        S, O1pre, O1, wb1, wb2 = ctx.saved_tensors
        lay = ctx.afno_layout
        if ctx.p6b is not None:
            (wb1, wb2), lay = ctx.p6b, 2
        dx, dw1, db1, dw2, db2 = _mixer_bwd(dy.contiguous(), S, O1pre, O1, wb1, wb2, ctx.dims, ctx.fused, lay, ctx.sinks)
        return dx, dw1, db1, dw2, db2, None, None, None, None, None, None


def _block_parts(x, n1w, n1b, n2w, n2b, f1w, f1b, packed, dims, mp, need_out, f2w=None, f2b=None, mlp_pk=None,
                 afno_layout=None, save=True):
    """forward of one Block up to (and optionally including) the second channel-MLP GEMM; returns every intermediate
    the backward needs.  Called by BlockFn.forward, and again by BlockFn.backward when activations are recomputed."""
    B, tok, E, h, w, nb, bs, mx, my, mh, act = dims
    Mm, M = B * mx * my, B * tok
    dev = x.device
    panel = _mlp_panel_mode(mlp_pk, M, E, mh, mp)
    npl = mlp_pk.planes if panel == 2 else 0
    both = (panel == 2 and npl == 1 and ops.tune("pack_both") != 0
            and ops.bf16_pack_both_supported(M, E) and ops.bf16_pack_both_supported(M, mh))
    # GroupNorm applied ON THE LOAD of its consumer (DPOT_TUNE gn_fuse=0 disables): norm2 inside the pack pass of the bf16
    # channel MLP (the fp32 GroupNorm2(y1) is never written); norm1 inside both DFT kernels where they are not fused with
    # the statistics anyway (32x32 latent grid: statistics-only GroupNorm launches, no normalised tensor)
    onload = ops.tune("gn_fuse") != 0
    pack_norm = both and onload and tok % 64 == 0 and (E // 8) % 4 == 0
    lay = afno_layout if afno_layout is not None else getattr(packed[0], "layout", 0)
    fused_xp = None
    if packed[0][2] is not None and ops.afno_fused_supported(h, w, E, nb, mx, my, B=B, layout=lay):
        # the whole AFNO layer - norm1, rfft2, both MLP layers, irfft2, + x_orig, norm2 - in ONE launch (csrc/afno_fused.hip,
        # SURVEY 8 f4): spectrum and hidden layer stay on chip; save=False (nothing will run a backward on these
        # intermediates: inference, or a forward whose Block recomputes): S / O1pre are not even written
        # (round 5: with the bf16 channel MLP the same launch also writes GroupNorm2(y1) as the two bf16 operand packs -
        # DPOT_TUNE packs=0: the separate pack pass over y1 instead)
        fused_packs = pack_norm and ops.tune("packs") != 0
        res = ops.afno_fused_fwd(
            x, n1w, n1b, packed[0][2], packed[0][1], packed[1][2], packed[1][1], n2w, n2b, h, w, nb, mx, my, act, save=save,
            want_y1=save or (pack_norm and not fused_packs), want_xn2=not pack_norm, want_packs=fused_packs,
            packs_trans=save)
        S, O1pre, y1, xn2, mean1, rstd1, mean2, rstd2 = res[:8]
        fused_xp = res[8:] if fused_packs else None
        O1 = None
    elif ops.gn_dft_supported(h, w, E):
        # GroupNorm fused with the neighbouring DFT (csrc/gn_dft.hip): norm1 + rfft2, and irfft2 + x_orig + norm2 -
        # two launches around the mixer instead of four, GroupNorm1(x) never written
        S, mean1, rstd1 = ops.gn_rfft2(x, n1w, n1b, h, w, nb, mx, my)
        O2, O1pre, O1 = _mixer_core(S, packed, dims, afno_layout)
        y1, xn2, mean2, rstd2 = ops.irfft2_gn(O2, x, mean1, rstd1, n1w, n1b, n2w, n2b, h, w, nb, mx, my,
                                              want_xn2=not pack_norm)
        del O2
    else:
        stats = onload and ops.groupnorm_stats_supported(B, tok, E)
        if stats and ops.rfft2_norm_supported(h, w, E):
            mean1, rstd1 = ops.groupnorm_stats(x, n1w, n1b)
            y1, S, O1pre, O1 = _mixer_fwd(x, packed, dims, afno_layout, norm=(mean1, rstd1, n1w, n1b))
        else:
            xn1, mean1, rstd1 = ops.groupnorm_fwd(x, n1w, n1b)
            y1, S, O1pre, O1 = _mixer_fwd(xn1, packed, dims, afno_layout)
            del xn1
        pack_norm = pack_norm and stats
        if pack_norm:
            xn2 = None
            mean2, rstd2 = ops.groupnorm_stats(y1, n2w, n2b)
        else:
            xn2, mean2, rstd2 = ops.groupnorm_fwd(y1, n2w, n2b)
    if both:
        # plain-bf16 channel MLP, one pack pass per activation: the pass that packs xn2 / Hh as the A operand of the
        # next GEMM also writes the TRANSPOSED pack the weight gradient will need - that (bf16, half the bytes) is what
        # the backward keeps; the fp32 xn2 / Hh are dropped right here
        if fused_xp is not None:
            xp, xpT = fused_xp                      # written by the one-launch AFNO layer itself
        elif pack_norm:
            xp, xpT, _ = ops.bf16_pack_both(y1.view(M, E), norm=(mean2, rstd2, n2w, n2b, tok))
        else:
            xp, xpT, _ = ops.bf16_pack_both(xn2.view(M, E))
        del xn2
        # fc1: the epilogue emits the activated hidden layer directly in its packed forms (no fp32 copy of it exists)
        # (Hpre here = act'(pre-activation) as a bf16 pack: all the backward needs of it, at half the bytes of the fp32
        # pre-activation and without a second activation evaluation)
        _, Hpre, hp, hpT, _ = ops.gemm_bf16p_packed(xp, mlp_pk[0], M, mh, E, bias=f1b, act=act, mode=EPI_ACT,
                                                    save_dact=True, pack_rows=need_out, pack_trans=True, store=False)
        out = None
        if need_out:
            out, _ = ops.gemm_bf16p(hp, mlp_pk[2], M, E, mh, bias=f2b, res=x.view(M, E))
        return out, (mean1, rstd1, S, O1pre, O1, y1, mean2, rstd2, xpT, Hpre, hpT)
    if panel == 2:   # bf16 matrix cores: weights pre-packed bf16 once per step, activations packed in one pass each
        Hh, Hpre = ops.gemm_bf16p(ops.bf16_pack_rows(xn2.view(M, E), planes=npl), mlp_pk[0], M, mh, E, bias=f1b, act=act,
                                  mode=EPI_ACT, save_pre=True, planes=npl)
    elif panel:    # static-weight panel GEMM (csrc/gemm_panel.hip): weights pre-packed once per optimiser step
        Hh, Hpre = ops.gemm_panel(xn2.view(M, E), mlp_pk[0], mh, bias=f1b, act=act, mode=EPI_ACT, save_pre=True)
    else:
        Hh, Hpre = ops.linear_fwd(xn2.view(M, E), f1w, f1b, act=act, save_pre=True, precision=mp)
    out = None
    if need_out:
        if panel == 2:
            out, _ = ops.gemm_bf16p(ops.bf16_pack_rows(Hh, planes=npl), mlp_pk[2], M, E, mh, bias=f2b, res=x.view(M, E),
                                    planes=npl)
        elif panel:
            out, _ = ops.gemm_panel(Hh, mlp_pk[2], E, bias=f2b, res=x.view(M, E))
        else:
            out, _ = ops.linear_fwd(Hh, f2w, f2b, res=x.view(M, E), precision=mp)
    return out, (mean1, rstd1, S, O1pre, O1, y1, mean2, rstd2, xn2, Hpre, Hh)


# The gradient a Block's backward returns (dx) is the `dout` of the PREVIOUS Block's backward, whose bf16 channel MLP first packs
# it (row form, transposed form, bias column sums: one more pass over it).  Round 5: the GroupNorm1-backward kernel that produces
# dx writes those packs itself (ops.groupnorm_bwd_packs); they travel beside the autograd gradient in this side table, keyed by
# the gradient's address.  An entry HOLDS dx, so its memory cannot be handed to another tensor while the entry lives (a stale
# key can never alias a different gradient); the consumer pops its entry, at most two entries are kept.
_GRAD_PACKS: "OrderedDict[int, tuple]" = OrderedDict()


def _stash_grad_packs(dx: Tensor, dop: Tensor, dopT: Tensor, cs: Tensor) -> None:
    _GRAD_PACKS[dx.data_ptr()] = (dx, dx._version, dop, dopT, cs)
    while len(_GRAD_PACKS) > 2:
        _GRAD_PACKS.popitem(last=False)


def clear_grad_packs() -> None:
    """drop the side table (the step classes call this after every backward: an entry nobody took - frozen prefix, a tensor
    hook that replaced the gradient - would otherwise pin dx and both packs until two later stashes evict it)"""
    _GRAD_PACKS.clear()


def _take_grad_packs(do2: Tensor):
    e = _GRAD_PACKS.pop(do2.data_ptr(), None)
    if e is None:
        return None
    dx, ver, dop, dopT, cs = e
    if dx.numel() != do2.numel() or dx._version != ver:       # another tensor / modified in place since: pack it afresh
        return None
    return dop, dopT, cs


class BlockFn(torch.autograd.Function):
    """x[B,tok,E] -> x + MLP(GN2(GN1(x) + irfft2(Mix(rfft2(GN1(x))))))

    ``recompute=True`` (activation recomputation, for long auto-regressive rollouts at 256^2: SURVEY 7 risk "T_ar>1
    backward"): only the block INPUT is kept; the backward re-runs the forward kernels (everything but the last GEMM)
    to rebuild the 11 intermediates.  Same kernels, same order -> bit-identical gradients, ~1/20 of the memory."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, w1, b1, w2, b2, n2w, n2b, f1w, f1b, f2w, f2b, h: int, w: int, nb: int, modes: int,
                act: int, packed=None, recompute: bool = False, mlp_pk=None, grad_enabled: bool = True,
                emit_grad_packs: bool = False):
        x = x.contiguous()
        B, tok, E = x.shape
        bs = E // nb
        mx, my = min(modes, h), min(modes, w // 2 + 1)
        mh = f1w.shape[0]
        if packed is None:      # ((Wbig1, bbig1, blocked W1, blocked W1^T), (...)); normally packed by the model
            packed = (ops.afno_pack3(w1, b1), ops.afno_pack3(w2, b2))
        dims = (B, tok, E, h, w, nb, bs, mx, my, mh, act)
        mp = ops.mlp_precision()                                               # channel-MLP GEMM precision override
        # (no input needs a gradient or the caller runs under no_grad - inference - or the Block recomputes: the
        # intermediates are dropped right below.  grad_enabled comes from the CALLER: inside forward() autograd is always off,
        # and needs_input_grad mirrors requires_grad of the inputs whatever the grad mode)
        keep = grad_enabled and any(ctx.needs_input_grad) and not recompute
        out, parts = _block_parts(x, n1w, n1b, n2w, n2b, f1w, f1b, packed, dims, mp, True, f2w, f2b, mlp_pk, save=keep)
        ctx.mlp_pk = mlp_pk if _mlp_panel_mode(mlp_pk, B * tok, E, mh, mp) else None
        ctx.fused_mixer = packed[0][2] is not None
        # weights the backward data path multiplies by: blocked W^T (fused kernel) or the plain Wbig (generic GEMM)
        wb1, wb2 = (packed[0][3], packed[1][3]) if ctx.fused_mixer else (packed[0][0], packed[1][0])
        opt_t = (packed[0][2], packed[1][2]) if ctx.fused_mixer else ()
        if recompute:
            ctx.save_for_backward(x, wb1, packed[0][1], wb2, packed[1][1], n1w, n1b, n2w, n2b, f1w, f1b, f2w, *opt_t)
        else:
            ctx.save_for_backward(x, *parts, wb1, wb2, n1w, n2w, f1w, f2w)
        ctx.recompute = recompute
        ctx.emit_grad_packs = emit_grad_packs     # a Block precedes this one: its bf16 channel-MLP backward wants dx packed
        ctx.dims = dims
        ctx.afno_layout = getattr(packed[0], "layout", 0) if ctx.fused_mixer else 0
        # bf16x6 mixer packs (persistent buffers of the model's AfnoPacks, epoch-checked like every derived weight)
        ctx.p6f = _p6_of(packed, 0) if ctx.fused_mixer else None
        ctx.p6b = _p6_of(packed, 1) if ctx.fused_mixer else None
        ctx.mlp_precision = ops.effective_mlp_precision()    # a concrete code: the backward reproduces the forward's mode
        ctx.gemm_precision = ops._cur_gemm()
        ctx.sinks = _sinks(ctx, (n1w, n1b, w1, b1, w2, b2, n2w, n2b, f1w, f1b, f2w, f2b), 1)
        ctx.weights_epoch = _epoch_of(ctx.sinks)
        return out.view(B, tok, E)

    @staticmethod
    def backward(ctx, dout):
        # the modes the forward ran in (per-model attributes), whatever thread autograd runs this on
        with ops.precision_scope(getattr(ctx, "gemm_precision", None), ctx.mlp_precision):
            return BlockFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        _check_epoch(ctx, "BlockFn")
        mp = ctx.mlp_precision
        if ctx.recompute:
            x, wb1, bb1, wb2, bb2, n1w, n1b, n2w, n2b, f1w, f1b, f2w, *wts = ctx.saved_tensors
            wb1T, wb2T = wts if ctx.fused_mixer else (None, None)
            p6f = ctx.p6f if ctx.p6f is not None else (None, None)
            with torch.no_grad():
                _, parts = _block_parts(x, n1w, n1b, n2w, n2b, f1w, f1b,
                                        (ops.AfnoItem((wb1, bb1, wb1T, None), ctx.afno_layout, (p6f[0], None)),
                                         ops.AfnoItem((wb2, bb2, wb2T, None), ctx.afno_layout, (p6f[1], None))),
                                        ctx.dims, mp, False, mlp_pk=ctx.mlp_pk, afno_layout=ctx.afno_layout)
            mean1, rstd1, S, O1pre, O1, y1, mean2, rstd2, xn2, Hpre, Hh = parts
        else:
            (x, mean1, rstd1, S, O1pre, O1, y1, mean2, rstd2, xn2, Hpre, Hh, wb1, wb2, n1w, n2w, f1w,
             f2w) = ctx.saved_tensors
        B, tok, E, h, w, nb, bs, mx, my, mh, act = ctx.dims
        s_n1w, s_n1b, s_w1, s_b1, s_w2, s_b2, s_n2w, s_n2b, s_f1w, s_f1b, s_f2w, s_f2b = ctx.sinks
        afno_lay = ctx.afno_layout
        if ctx.p6b is not None:      # the mixer's data gradient on the bf16 matrix cores (bf16x6), csrc/afno_mlp6.hip
            (wb1, wb2), afno_lay = ctx.p6b, 2
        M, Mm = B * tok, B * mx * my
        dev = dout.device
        dout = dout.contiguous()
        do2 = dout.view(M, E)
        # one finalising launch per block (csrc/gemm_tn.hip block_finalize_kernel): the fused weight-gradient launches leave
        # their split-K partials, and those are reduced together with the GroupNorm parameter-gradient partials at the end
        pending = {} if ops.block_finalize_enabled() else None
        # channel MLP
        mlp_pk = ctx.mlp_pk
        bf16p = mlp_pk is not None and mlp_pk.kind != "f32"
        npl = mlp_pk.planes if bf16p else 0

        def wgrad(dy, xin, s_w, s_b, shape):
            n, k = dy.shape[1], xin.shape[1]
            if bf16p:   # dW[n, k] = dy^T xin on the bf16 matrix cores: both operands packed transposed (k-dim = tokens),
                #         split-K over the tokens with a fixed-order reduction; the bias gradient is a column sum
                dw, _ = ops.gemm_bf16p(ops.bf16_pack_rows(dy, trans=True, planes=npl),
                                       ops.bf16_pack_rows(xin, trans=True, planes=npl), n, k, M, out=s_w.out(),
                                       planes=npl)
                db = ops.colsum(dy, M, n, out=s_b.out())
            else:
                dw, db = ops.linear_bwd_wb(dy, xin, s_w.out(), s_b.out(), precision=mp)
            return s_w.done(dw.view(shape)), s_b.done(db)

        if bf16p and xn2.dtype == torch.bfloat16:
            # pack-both path (see _block_parts): xn2 / Hh ARE the transposed bf16 packs; each gradient is packed once, in
            # both forms, and its bias column sums come out of the same pass
            dcs = pending is not None          # bias column sums: partials now, summed by the block's finalising launch
            taken = _take_grad_packs(do2)
            if taken is not None:              # packed by the kernel that produced this gradient (the next Block's backward)
                dop, dopT, cs_b = taken             # cs_b: [B * token ranges, E] partial column sums
                df2b = ((cs_b, cs_b.shape[0], E, ops._out(s_f2b.out(), (E,), dev)) if dcs
                        else ops.colsum(cs_b, cs_b.shape[0], E, out=s_f2b.out()))
            else:
                dop, dopT, df2b = ops.bf16_pack_both(do2, want_colsum=True, colsum_out=s_f2b.out(), defer_colsum=dcs)
            # both weight gradients in ONE launch once dHpre's pack exists, when each alone would need split-K
            pair = ops.gemm_bf16p_pair_wanted(E, mh, mh, E, M)
            if not pair:
                df2w, _ = ops.gemm_bf16p(dopT, Hh, E, mh, M, out=s_f2w.out())
            # dHpre = (do2 W2) * act'(Hpre) leaves the GEMM as its packs + bias column sums only
            _, _, dhp, dhpT, df1b = ops.gemm_bf16p_packed(dop, mlp_pk[3], M, mh, E, act=act, mode=EPI_DACT, dact=Hpre,
                                                          pack_rows=True, pack_trans=True, colsum=True,
                                                          colsum_out=s_f1b.out(), store=False, defer_colsum=dcs)
            if dcs:
                pending["cs"] = [df2b, df1b]
                df2b, df1b = df2b[3], df1b[3]
            if pair:
                df2w, df1w = ops.gemm_bf16p_pair(dopT, Hh, E, mh, dhpT, xn2, mh, E, M, out0=s_f2w.out(), out1=s_f1w.out())
            else:
                df1w, _ = ops.gemm_bf16p(dhpT, xn2, mh, E, M, out=s_f1w.out())
            del dop, dopT
            df2w, df1w = s_f2w.done(df2w.view(E, mh, 1, 1)), s_f1w.done(df1w.view(mh, E, 1, 1))
            if not dcs:
                df2b, df1b = s_f2b.done(df2b), s_f1b.done(df1b)
            dxn2, _ = ops.gemm_bf16p(dhp, mlp_pk[1], M, E, mh)
            del dhp, dhpT
        else:
            # native fp32: both weight gradients in ONE launch once dHpre exists (csrc/gemm_tn.hip, dpot_mlp_wgrad2)
            skm = 0 if bf16p else ops.mlp_wgrad2_splitk(M, E, mh, mp)
            if skm and not (do2.is_contiguous() and Hh.is_contiguous() and xn2.is_contiguous()):
                skm = 0
            if not skm:
                df2w, df2b = wgrad(do2, Hh, s_f2w, s_f2b, (E, mh, 1, 1))
            if bf16p:
                dHpre, _ = ops.gemm_bf16p(ops.bf16_pack_rows(do2, planes=npl), mlp_pk[3], M, mh, E, act=act, mode=EPI_DACT,
                                          aux=Hpre, planes=npl)
            elif mlp_pk is not None:
                dHpre, _ = ops.gemm_panel(do2, mlp_pk[3], mh, act=act, mode=EPI_DACT, aux=Hpre)       # do2 W2, * act'(Hpre)
            else:
                dHpre = ops.linear_bwd_data(do2, f2w, act=act, aux=Hpre, precision=mp)  # [M, mh]
            if skm:
                df2w, df2b = ops._out(s_f2w.out(), (E, mh), dev), ops._out(s_f2b.out(), (E,), dev)
                df1w, df1b = ops._out(s_f1w.out(), (mh, E), dev), ops._out(s_f1b.out(), (mh,), dev)
                if pending is not None:      # partials only: reduced by the block's finalising launch below
                    pending["mlp"] = ops.mlp_wgrad2(do2, Hh, xn2.view(M, E), dHpre, df2w, df2b, df1w, df1b, skm, defer=True)
                else:
                    ops.mlp_wgrad2(do2, Hh, xn2.view(M, E), dHpre, df2w, df2b, df1w, df1b, skm)
                    df2w, df2b = s_f2w.done(df2w.view(E, mh, 1, 1)), s_f2b.done(df2b)
                    df1w, df1b = s_f1w.done(df1w.view(mh, E, 1, 1)), s_f1b.done(df1b)
            else:
                df1w, df1b = wgrad(dHpre, xn2.view(M, E), s_f1w, s_f1b, (mh, E, 1, 1))
            if bf16p:
                dxn2, _ = ops.gemm_bf16p(ops.bf16_pack_rows(dHpre, planes=npl), mlp_pk[1], M, E, mh, planes=npl)
            elif mlp_pk is not None:
                dxn2, _ = ops.gemm_panel(dHpre, mlp_pk[1], E)                      # dHpre W1
            else:
                dxn2 = ops.linear_bwd_data(dHpre, f1w, precision=mp)               # [M, E]
        # parameter-gradient partials of norm2 are reduced together with norm1's at the end of the block (one launch)
        if ops.gn_dft_supported(h, w, E):
            # norm2 backward + rfft2 (adjoint of the forward irfft2), then irfft2 (adjoint) + skip + norm1 backward + outer
            # skip: two launches around the mixer's backward (csrc/gn_dft.hip)
            dy1, gn2_part, dO2 = ops.gn_bwd_rfft2(dxn2.view(B, tok, E), y1, mean2, rstd2, n2w, h, w, nb, mx, my,
                                                  col_weights=1)
            dS, dw1, db1, dw2, db2 = _mixer_core_bwd(dO2, S, O1pre, O1, wb1, wb2, ctx.dims, ctx.fused_mixer,
                                                     afno_lay, (s_w1, s_b1, s_w2, s_b2), pending)
            if E // 8 <= 64:
                dx, gn1_part = ops.irfft2_gn_bwd(dS, dy1, x, mean1, rstd1, n1w, h, w, nb, mx, my, add=dout,
                                                 col_weights=0)
            else:
                # 128 channels per group (DPOT-S / -M): this pair reads four fields with 4-byte accesses and runs at 64.7 us
                # against 20.0 + 24.6 us for the separate kernels (profiles/r03_step_census_M_bf16_v1.txt) - not fused
                dxn1 = ops.irfft2(dS, B, h, w, E, nb, mx, my, 0, res=dy1)
                if (ctx.emit_grad_packs and ctx.needs_input_grad[0] and bf16p and xn2.dtype == torch.bfloat16 and ops.groupnorm_bwd_packs_supported(tok, E, B=B)
                        and ops.tune("packs") != 0):
                    # dx goes to the previous Block's bf16 channel-MLP backward: written here in its packed forms as well
                    dx, gn1_part, gp_r, gp_t, gp_cs = ops.groupnorm_bwd_packs(dxn1, x, mean1, rstd1, n1w, add=dout)
                    _stash_grad_packs(dx, gp_r, gp_t, gp_cs)
                else:
                    dx, gn1_part = ops.groupnorm_bwd(dxn1, x, mean1, rstd1, n1w, add=dout, defer=True)
        else:
            dy1, gn2_part = ops.groupnorm_bwd(dxn2.view(B, tok, E), y1, mean2, rstd2, n2w, defer=True)
            # AFNO mixer
            dxn1, dw1, db1, dw2, db2 = _mixer_bwd(dy1, S, O1pre, O1, wb1, wb2, ctx.dims, ctx.fused_mixer,
                                                  afno_lay, (s_w1, s_b1, s_w2, s_b2), pending)
            # (DPOT-L: the CHUNKED GroupNorm backward could write the gradient's packs too - see the 128-channel branch above - but
            # there the staging + two barriers per 32-token sub-tile cost what the saved pack pass did: DPOT-L 91.2 -> 90.9 ms,
            # L20 2.185 -> 2.211 s, profiles/r05_grad_packs_step_ab_L.txt; the opt-in of round 5 is gone)
            dx, gn1_part = ops.groupnorm_bwd(dxn1, x, mean1, rstd1, n1w, add=dout, defer=True)
        gn_jobs = [(gn1_part, s_n1w.out(), s_n1b.out()), (gn2_part, s_n2w.out(), s_n2b.out())]
        if pending:
            (dn1w, dn1b), (dn2w, dn2b) = ops.block_finalize(pending.get("afno"), pending.get("mlp"), gn_jobs,
                                                            pending.get("cs", ()))
            if "cs" in pending:
                df2b, df1b = s_f2b.done(df2b), s_f1b.done(df1b)
            if "mlp" in pending:
                df2w, df2b = s_f2w.done(df2w.view(E, mh, 1, 1)), s_f2b.done(df2b)
                df1w, df1b = s_f1w.done(df1w.view(mh, E, 1, 1)), s_f1b.done(df1b)
            if "afno" in pending:
                dw1, db1, dw2, db2 = s_w1.done(dw1), s_b1.done(db1), s_w2.done(dw2), s_b2.done(db2)
        else:
            (dn1w, dn1b), (dn2w, dn2b) = ops.groupnorm_param_grads(gn_jobs)
        dn1w, dn1b = s_n1w.done(dn1w), s_n1b.done(dn1b)
        dn2w, dn2b = s_n2w.done(dn2w), s_n2b.done(dn2b)
        return (dx, dn1w, dn1b, dw1, db1, dw2, db2, dn2w, dn2b, df1w, df1b, df2w, df2b, None, None, None, None, None,
                None, None, None, None, None)


# ======================================================================================================
def head_layout_jobs(o0b, o4w, o4b, P: int, old: int):
    """the small layout pieces of head_derived as ops.LayoutJobs entries, in the order (bexp, [w4p, b4p])"""
    co = o4w.shape[0]
    PP = P * P
    jobs = [(o0b, None, (1, PP, old), (1, PP, old), (0, 0, 1))]                # bias repeated per pixel of the patch
    if old == 32 and 0 < co <= 32:
        jobs += [(o4w, None, (1, 32, 32), (1, co, 32), (0, 32, 1)), (o4b, None, (1, 1, 32), (1, 1, co), (0, 0, 1))]
    return jobs


def head_derived(o0w, o0b, o4w, o4b, P: int, wt_out=None, layouts=None):
    """weight-only layouts of the de-embed stage: ConvTranspose2d(k=s=P) weight as the GEMM matrix whose columns are
    ordered (i, j, o) - the GEMM result IS the pixel-major [pixels, old] matrix -, its bias repeated per pixel, and
    the zero-padded last 1x1 conv of the fused tail (None when the fused tail does not apply).  layouts: the tensors of
    `head_layout_jobs`, already refreshed"""
    E, old = o0w.shape[0], o0w.shape[1]
    co = o4w.shape[0]
    PP = P * P
    wt = ops.transpose2d(o0w, E, old, PP, out=wt_out).view(E, PP * old)
    w4p = b4p = None
    if layouts is not None:
        bexp = layouts[0].view(PP * old)
        if len(layouts) > 1:
            w4p, b4p = layouts[1].view(32, 32), layouts[2].view(32)
        return wt, bexp, w4p, b4p
    bexp = ops.tile_vec(o0b, PP)
    if old == 32 and 0 < co <= 32:
        w4p, b4p = ops.out_tail_pad(o4w, o4b, co)
    return wt, bexp, w4p, b4p


class HeadFn(torch.autograd.Function):
    """x[B,tok,E] -> (pred [B,X,Y,T_out*C_out], cls_pred [B,n_cls])"""

    @staticmethod
    def forward(ctx, x, o0w, o0b, o2w, o2b, o4w, o4b, c0w, c0b, c2w, c2b, c4w, c4b, h: int, w: int, P: int, act: int,
                derived=None):
        ops.capture_precision(ctx)       # re-applied around backward (ops.with_ctx_precision)
        ctx.set_materialize_grads(False)      # an unused output (cls_pred in train_temporal.py:226) costs nothing
        x = x.contiguous()
        B, tok, E = x.shape
        old, co = o0w.shape[1], o4w.shape[0]
        PP = P * P
        M, Mp = B * tok, B * tok * PP
        dev = x.device
        fused = ops.out_tail_supported(old, co, Mp)
        if derived is None:
            derived = head_derived(o0w, o0b, o4w, o4b, P)
        wt, bexp, w4p, b4p = derived[:4]
        head_pk = derived[4] if len(derived) > 4 else None
        if head_pk is not None and not (ops.panel_enabled() and ops.gemm_panel_supported(M, PP * old, E)
                                        and ops.gemm_panel_supported(M, E, PP * old)):
            head_pk = None
        ctx.head_pk = head_pk
        if fused:
            # GEMM writes only the pre-activation; the whole per-pixel tail (act, 1x1, act, 1x1, pixel shuffle) is one
            # kernel that reads it once (csrc/tail.hip)
            U = V = Vpre = None
            Upre = torch.empty(M, PP * old, dtype=torch.float32, device=dev)
            if head_pk is not None:
                Upre, _ = ops.gemm_panel(x.view(M, E), head_pk[0], PP * old, bias=bexp)
            else:
                ops.gemm(x, wt, Upre, M, PP * old, E, lda=E, ldb=PP * old, ldc=PP * old, bias=bexp)
            pred = ops.out_tail_fwd(Upre, o2w, o2b, w4p, b4p, B, h, w, P, co, act)  # [B, X, Y, co]
            V = w4p                                                            # (saved slot reused: padded W4)
        else:
            if head_pk is not None:
                U, Upre = ops.gemm_panel(x.view(M, E), head_pk[0], PP * old, bias=bexp, act=act, mode=EPI_ACT,
                                         save_pre=True)
            else:
                U = torch.empty(M, PP * old, dtype=torch.float32, device=dev)
                Upre = torch.empty_like(U)
                ops.gemm(x, wt, U, M, PP * old, E, lda=E, ldb=PP * old, ldc=PP * old, bias=bexp, act=act,
                         mode=EPI_ACT, preact=Upre, ldpre=PP * old)
            V, Vpre = ops.linear_fwd(U.view(Mp, old), o2w, o2b, act=act, save_pre=True)
            Z, _ = ops.linear_fwd(V, o4w, o4b)                                 # [Mp, co]
            pred = ops.pixel_shuffle(Z, B, h, w, P, co)                        # [B, X, Y, co]
        # classification head
        cm = ops.token_mean(x)
        c1, c1pre = ops.linear_fwd(cm, c0w, c0b, act=act, save_pre=True)
        c2, c2pre = ops.linear_fwd(c1, c2w, c2b, act=act, save_pre=True)
        cls, _ = ops.linear_fwd(c2, c4w, c4b)
        ctx.save_for_backward(x, wt, U, Upre, V, Vpre, o2w, o2b, o4w, cm, c1, c1pre, c2, c2pre, c0w, c2w, c4w)
        ctx.dims = (B, tok, E, h, w, P, old, co, act)
        ctx.fused = fused
        # the cls_head sinks are registered lazily in backward: whether that branch runs is only known there
        ctx.sinks = _sinks(ctx, (o0w, o0b, o2w, o2b, o4w, o4b), 1)
        ctx.cls_params = (c0w, c0b, c2w, c2b, c4w, c4b)
        ctx.cls_needed = tuple(ctx.needs_input_grad[7:13])
        ctx.weights_epoch = _epoch_of(ctx.sinks)
        return pred, cls

    @staticmethod
    @ops.with_ctx_precision
    def backward(ctx, dpred, dcls):
        _check_epoch(ctx, "HeadFn")
        x, wt, U, Upre, V, Vpre, o2w, o2b, o4w, cm, c1, c1pre, c2, c2pre, c0w, c2w, c4w = ctx.saved_tensors
        B, tok, E, h, w, P, old, co, act = ctx.dims
        s_o0w, s_o0b, s_o2w, s_o2b, s_o4w, s_o4b = ctx.sinks
        PP = P * P
        M, Mp = B * tok, B * tok * PP
        dev = x.device
        o2w2, o4w2 = o2w.view(old, old), o4w.view(co, old)
        do0w = do0b = do2w = do2b = do4w = do4b = dx_out = None
        dc0w = dc0b = dc2w = dc2b = dc4w = dc4b = None
        if dpred is not None:
            # ---- out layer
            if ctx.fused:
                # the five small parameter gradients leave the partial-row reduction straight into their slots
                dUpre, (g2w, g4w, g2b, g0b, g4b) = ops.out_tail_bwd(
                    Upre, dpred.contiguous(), o2w, o2b, V, B, h, w, P, co, act,
                    outs=(s_o2w.out(), s_o4w.out(), s_o2b.out(), s_o0b.out(), s_o4b.out()))
                do2w = s_o2w.done(g2w.view(old, old, 1, 1))
                do4w = s_o4w.done(g4w.view(co, old, 1, 1))
                do2b, do0b, do4b = s_o2b.done(g2b), s_o0b.done(g0b), s_o4b.done(g4b)
            else:
                dZ = ops.pixel_shuffle(dpred.contiguous(), B, h, w, P, co, inverse=True)  # [Mp, co]
                dVpre = ops.linear_bwd_data(dZ, o4w2, act=act, aux=Vpre)                  # [Mp, old]
                do4w, do4b = ops.linear_bwd_wb(dZ, V, s_o4w.out(), s_o4b.out())
                do4w, do4b = s_o4w.done(do4w.view(co, old, 1, 1)), s_o4b.done(do4b)
                dUpre = ops.linear_bwd_data(dVpre, o2w2, act=act, aux=Upre.view(Mp, old))  # [Mp, old]
                do2w, do2b = ops.linear_bwd_wb(dVpre, U.view(Mp, old), s_o2w.out(), s_o2b.out())
                do2w, do2b = s_o2w.done(do2w.view(old, old, 1, 1)), s_o2b.done(do2b)
                do0b = s_o0b.done(ops.colsum(dUpre, Mp, old, out=s_o0b.out()))
            dU2 = dUpre.view(M, PP * old)
            if ctx.head_pk is not None:
                dx_out, _ = ops.gemm_panel(dU2, ctx.head_pk[1], E)
            else:
                dx_out = torch.empty(M, E, dtype=torch.float32, device=dev)
                ops.gemm(dU2, wt, dx_out, M, E, PP * old, transB=True, lda=PP * old, ldb=PP * old, ldc=E)
            dwt = torch.empty(E, PP * old, dtype=torch.float32, device=dev)
            ops.gemm(x, dU2, dwt, E, PP * old, M, transA=True, lda=E, ldb=PP * old, ldc=PP * old,
                     splitk=ops.auto_splitk(E, PP * old, M, tn=True))
            do0w = s_o0w.done(ops.transpose2d(dwt, E, PP, old, out=s_o0w.out()).view(E, old, P, P))
            dx_out = dx_out.view(B, tok, E)
        else:
            for sk in ctx.sinks:
                sk.skip()
        dx = dx_out
        if dcls is not None:
            # ---- cls head
            s_c0w, s_c0b, s_c2w, s_c2b, s_c4w, s_c4b = [_Sink(p, n) for p, n in zip(ctx.cls_params, ctx.cls_needed)]
            dcls = dcls.contiguous()
            dc2pre = ops.linear_bwd_data(dcls, c4w, act=act, aux=c2pre)
            dc4w, dc4b = ops.linear_bwd_wb(dcls, c2, s_c4w.out(), s_c4b.out())
            dc4w, dc4b = s_c4w.done(dc4w), s_c4b.done(dc4b)
            dc1pre = ops.linear_bwd_data(dc2pre, c2w, act=act, aux=c1pre)
            dc2w, dc2b = ops.linear_bwd_wb(dc2pre, c1, s_c2w.out(), s_c2b.out())
            dc2w, dc2b = s_c2w.done(dc2w), s_c2b.done(dc2b)
            dcm = ops.linear_bwd_data(dc1pre, c0w)
            dc0w, dc0b = ops.linear_bwd_wb(dc1pre, cm, s_c0w.out(), s_c0b.out())
            dc0w, dc0b = s_c0w.done(dc0w), s_c0b.done(dc0b)
            dx = ops.token_mean_bwd(dcm, tok, add=dx_out)
        return (dx, do0w, do0b, do2w, do2b, do4w, do4b, dc0w, dc0b, dc2w, dc2b, dc4w, dc4b, None, None, None, None,
                None)


# ======================================================================================================
class RelL2Fn(torch.autograd.Function):
    """sum_b sum_c ||(x-y) m||_2 / (||y m||_2 + 1e-8) / n_channels_b   (SimpleLpLoss(size_average=False))"""

    @staticmethod
    def forward(ctx, x, y, mask: Optional[Tensor]):
        x, y = ops._req(x.contiguous(), "pred"), ops._req(y.contiguous(), "target")
        B, Cc = x.shape[0], x.shape[-1]
        Tt = x.shape[-2] if x.dim() >= 3 else 1
        S = x.numel() // (B * Cc)
        if mask is not None:
            mask = mask.contiguous()
            if mask.numel() == x.numel():
                Tt_m = 1
            elif mask.numel() * Tt == x.numel():
                Tt_m = Tt
            else:
                raise ValueError(f"mask shape {tuple(mask.shape)} does not broadcast over {tuple(x.shape)}")
        else:
            Tt_m = 1
        loss, stats = ops.rel_l2_fwd(x, y, mask, B, S, Cc, Tt_m)
        ctx.save_for_backward(x, y, mask if mask is not None else x.new_empty(0), stats)
        ctx.meta = (B, S, Cc, Tt_m, mask is not None)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        x, y, mask, stats = ctx.saved_tensors
        B, S, Cc, Tt_m, has_mask = ctx.meta
        dx = ops.rel_l2_bwd(x, y, mask if has_mask else None, stats, g.contiguous().view(1), B, S, Cc, Tt_m)
        return dx, None, None


class AdaINFn(torch.autograd.Function):
    """lat * scale[b,:] + shift[b,:]   (models/dpot.py:386-387, the normalize=True branch)"""

    @staticmethod
    def forward(ctx, lat, scale, shift):
        lat, scale, shift = lat.contiguous(), scale.contiguous(), shift.contiguous()
        ctx.save_for_backward(lat, scale)
        return ops.scale_shift(lat, scale, shift)

    @staticmethod
    def backward(ctx, g):
        lat, scale = ctx.saved_tensors
        return ops.scale_shift_bwd(g.contiguous(), lat, scale)


def rel_l2_loss(x: Tensor, y: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    return RelL2Fn.apply(x, y, mask)
