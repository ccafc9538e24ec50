"""Tensor-level wrappers over the C ABI (include/dpot_hip.h).

PyTorch is used here only as the owner of device memory and of the HIP stream: every function takes
contiguous fp32 CUDA tensors, passes ``data_ptr()`` / sizes / the current stream to libdpot_hip.so and returns
tensors.  There is deliberately no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import contextlib
import functools
import threading
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import GemmDesc, check

Tensor = torch.Tensor

ACT_IDS = {None: 0, "none": 0, "gelu": 1, "tanh": 2, "sigmoid": 3, "relu": 4, "leaky_relu": 5, "softplus": 6,
           "ELU": 7, "silu": 8}
EPI_LINEAR, EPI_ACT, EPI_DACT, EPI_AFNO_WGRAD = 0, 1, 2, 3
GEMM_F32, GEMM_BF16X6, GEMM_AUTO, GEMM_BF16 = 0, 1, 2, 3
_PRECISIONS = {"f32": GEMM_F32, "bf16x6": GEMM_BF16X6, "auto": GEMM_AUTO, "bf16": GEMM_BF16}
# how every GEMM forms its fp32 products (include/dpot_hip.h: dpot_gemm_desc.precision);
# DPOT_GEMM_PRECISION = f32 | bf16x6 | auto
_gemm_precision = _PRECISIONS[os.environ.get("DPOT_GEMM_PRECISION", "f32")]


def set_gemm_precision(name: str) -> None:
    """'f32' = native fp32 MFMA, 'bf16x6' = fp32 emulated by 3-way bf16 operand splitting (fp32-level accuracy),
    'auto' = per shape, whichever of the two is faster (the library's rule)"""
    global _gemm_precision
    _gemm_precision = _PRECISIONS[name]


def gemm_precision() -> str:
    return {v: k for k, v in _PRECISIONS.items()}[_cur_gemm()]


# precision of the channel-MLP GEMMs only (Block.mlp, models/dpot.py:157-161); None = follow the global setting.
# BASELINE.json configs[2] asks for a "bf16 channel-MLP on MFMA": 'bf16x6' here puts exactly those GEMMs on the bf16
# matrix cores at fp32 accuracy; 'bf16' is the plain reduced-precision form (operands rounded to bf16, fp32
# accumulation - NOT within the 1e-4 parity tolerance, opt-in only).   DPOT_MLP_PRECISION = f32 | bf16x6 | auto | bf16
_mlp_precision = _PRECISIONS[os.environ["DPOT_MLP_PRECISION"]] if os.environ.get("DPOT_MLP_PRECISION") else None


def set_mlp_precision(name: Optional[str]) -> None:
    global _mlp_precision
    _mlp_precision = None if name is None else _PRECISIONS[name]


# per-THREAD overrides of the two process defaults above, set by the scopes below (round 5; ADVICE r4: the per-model
# precision used to be a temporary write to the process-global, visible to a backward of another model running on autograd's
# device thread at the same time).  A model applies its attributes around its own forward / weight derivation; every autograd
# Function captures the effective values in its context and re-applies them around its backward, whatever thread runs it.
_tls = threading.local()
_UNSET = object()


def _cur_gemm() -> int:
    v = getattr(_tls, "gemm", None)
    return _gemm_precision if v is None else v


def _cur_mlp() -> Optional[int]:
    v = getattr(_tls, "mlp", _UNSET)
    return _mlp_precision if v is _UNSET else v


def _code(prec) -> int:
    return _PRECISIONS[prec] if isinstance(prec, str) else int(prec)


@contextlib.contextmanager
def mlp_precision_scope(prec):
    """the channel-MLP precision of ONE model while its kernels are being enqueued: `DPOTNet.mlp_precision` (a
    per-model attribute: 'f32' | 'bf16x6' | 'auto' | 'bf16' | None = the process default above) is applied around the
    model's forward, its weight derivation and - through the value captured in the autograd context - its backward, so two
    models of one process can run different modes and nothing leaks to the next caller or to another thread.  prec: a name,
    a precision code or None (= leave the current setting)."""
    if prec is None:
        yield
        return
    prev = getattr(_tls, "mlp", _UNSET)
    _tls.mlp = _code(prec)
    try:
        yield
    finally:
        if prev is _UNSET:
            del _tls.mlp
        else:
            _tls.mlp = prev


@contextlib.contextmanager
def gemm_precision_scope(prec):
    """the same for the GEMM precision of everything OUTSIDE the channel MLP (`DPOTNet.gemm_precision`: 'f32' | 'bf16x6' |
    'auto' | None = the process default of set_gemm_precision / DPOT_GEMM_PRECISION)"""
    if prec is None:
        yield
        return
    prev = getattr(_tls, "gemm", None)
    _tls.gemm = _code(prec)
    try:
        yield
    finally:
        _tls.gemm = prev


@contextlib.contextmanager
def precision_scope(gemm=None, mlp=None):
    with gemm_precision_scope(gemm), mlp_precision_scope(mlp):
        yield


def with_ctx_precision(backward):
    """decorator for torch.autograd.Function.backward: re-apply the precisions the forward captured in ctx.gemm_precision /
    ctx.mlp_precision (see capture_precision) - autograd runs backward on its own thread, long after the model's scope"""
    @functools.wraps(backward)
    def wrapper(ctx, *grads):
        with precision_scope(getattr(ctx, "gemm_precision", None), getattr(ctx, "mlp_precision", None)):
            return backward(ctx, *grads)
    return wrapper


def capture_precision(ctx) -> None:
    """store the precisions in effect on this thread in an autograd context (forward side of with_ctx_precision)"""
    ctx.gemm_precision = _cur_gemm()
    # the EFFECTIVE channel-MLP precision as a concrete code: an unset override (None = "follow the GEMM precision")
    # would make the backward's scope a no-op and let it read whatever the autograd thread's default is by then
    ctx.mlp_precision = effective_mlp_precision()


def effective_mlp_precision() -> int:
    """precision the channel-MLP GEMMs run in: the override if set, else the GEMM precision in effect"""
    m = _cur_mlp()
    return m if m is not None else _cur_gemm()


def mlp_precision() -> Optional[int]:
    return _cur_mlp()


# ---- DPOT_TUNE: the ONE switchboard of the fallback-path selectors (round 6; rounds 1-5 grew 43 separate DPOT_* variables) ------
# DPOT_TUNE="key=val,key=val" (integers).  Every key selects a PREVIOUS-GENERATION path that stays under the GPU test gate
# (tests/test_gpu_optout.py runs parity subsets under them); the defaults are what measured fastest.  The C side reads the same
# variable once per process (csrc/core.hip dpot::tune); this side reads it per call.  Besides DPOT_TUNE the package reads only
# DPOT_HIP_LIB (library path), DPOT_GEMM_PRECISION and DPOT_MLP_PRECISION (process defaults of the precision modes).
TUNE_KEYS = {
    "mixer": (3, "AFNO mixer MLP: 3 = three-product fused kernel (afno_mlp3), 4 = four-product fused kernel (afno_mlp2), "
                 "0 = two generic GEMM launches"),
    "afno_layer": (-1, "one-launch AFNO layer forward (afno_fused_fwd): -1 = where it measured faster (>= 80 % full last "
                       "round of (sample, block) workgroups), 0 = never, 1 = wherever supported"),
    "gn_fuse": (1, "0 = GroupNorm as its own kernels: not fused with the mixer's DFTs, never applied on the load of its consumer"),
    "panel": (1, "0 = generic GEMM kernels instead of the pre-packed-weight panel GEMM, the token-contraction weight-gradient "
                 "kernels (gemm_tn) and their two-layer launches"),
    "wgrad_gauss": (1, "0 = four-product AFNO weight gradients instead of the three-product forms"),
    "bf16p_bd": (1, "0 = the LDS-DMA bf16 GEMM kernels of rounds 2-3 for every launch instead of the B-direct kernels"),
    "bf16p_shape": (1, "0 = none of the bf16 GEMM shape heuristics: 128 x 256 tiles only, column-major tile order, eight waves, "
                       "un-paired weight gradients"),
    "packs": (1, "0 = bf16 packs by separate pack passes only: none from the one-launch AFNO layer, the GroupNorm backward or Adam"),
    "pack_both": (1, "0 = bf16 channel MLP without the pack-both path (one pack pass per operand form, fp32 pre-activation saved)"),
    "fused_small": (1, "0 = separate small launches: per-block reduce launches, eight layout launches, torch ops for the cls head"),
    "embed_implicit": (1, "0 = explicit patch matrix + GEMM instead of the implicit-GEMM patch embedding"),
    "mixer6": (1, "bf16x6 mixer kernel afno_mlp6 under gemm_precision 'auto' / 'bf16x6': 1 = where it measured faster (96 channels "
                  "per block: DPOT-L), 2 = wherever supported (also 128), 0 = never (the fp32 matrix-core kernels)"),
}


def _parse_tune():
    out = {}
    for kv in os.environ.get("DPOT_TUNE", "").split(","):
        kv = kv.strip()
        if not kv:
            continue
        k, _, v = kv.partition("=")
        if k not in TUNE_KEYS:
            raise ValueError(f"DPOT_TUNE: unknown key {k!r} (known: {', '.join(TUNE_KEYS)})")
        out[k] = int(v)
    return out


def tune(key: str, default: Optional[int] = None) -> int:
    """value of `key` in DPOT_TUNE, else its default (TUNE_KEYS)"""
    d = TUNE_KEYS[key][0] if default is None else default
    return _parse_tune().get(key, d)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: Tensor, name: str = "tensor") -> Tensor:
    if not t.is_cuda:
        raise _lib.DpotHipError(f"{name} must live on the MI355X (got a {t.device} tensor): dpot_amd has no CPU path")
    if t.dtype != torch.float32:
        raise _lib.DpotHipError(f"{name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.DpotHipError(f"{name} must be contiguous")
    return t


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------------
def auto_splitk(M: int, N: int, K: int, batch: int = 1, precision: Optional[int] = None, tn: bool = False) -> int:
    """split-K factor for a GEMM; tn: a weight gradient (transA, not transB) - in native fp32 those run on the kernel
    of csrc/gemm_tn.hip, which wants one 128x128 workgroup per CU"""
    lib = _lib.load()
    prec = _cur_gemm() if precision is None else precision
    if tn and prec == GEMM_F32:
        s = lib.dpot_gemm_tn_splitk(M, N, K, batch)
        if s > 0:
            return s
    return lib.dpot_gemm_auto_splitk2(M, N, K, batch, prec)


def gemm(A: Tensor, B: Tensor, C_: Tensor, M: int, N: int, K: int, *, transA: bool = False, transB: bool = False,
         lda: int, ldb: int, ldc: int, batch: int = 1, strideA: int = 0, strideB: int = 0, strideC: int = 0,
         bias: Optional[Tensor] = None, strideBias: int = 0, act: int = 0, mode: int = EPI_LINEAR,
         aux: Optional[Tensor] = None, ldaux: int = 0, strideAux: int = 0,
         preact: Optional[Tensor] = None, ldpre: int = 0, stridePre: int = 0,
         res: Optional[Tensor] = None, ldres: int = 0, res_div: int = 0, res_mod: int = 0, strideRes: int = 0,
         accumulate: bool = False, splitk: Optional[int] = None, tile: int = 0, tag: int = 0,
         colsum_out: Optional[Tensor] = None, colsum_of: int = 0, strideColsum: int = 0,
         precision: Optional[int] = None) -> Tensor:
    """C = epilogue(A @ B) on the matrix cores; see include/dpot_hip.h for the exact semantics."""
    lib = _lib.load()
    d = GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C_.data_ptr()
    d.M, d.N, d.K, d.batch = M, N, K, batch
    d.transA, d.transB = int(transA), int(transB)
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.strideA, d.strideB, d.strideC = strideA, strideB, strideC
    d.bias, d.strideBias = _p(bias), strideBias
    d.act, d.epi_mode = act, mode
    d.aux, d.ldaux, d.strideAux = _p(aux), ldaux, strideAux
    d.preact, d.ldpre, d.stridePre = _p(preact), ldpre, stridePre
    d.res, d.ldres, d.res_div, d.res_mod, d.strideRes = _p(res), ldres, res_div, res_mod, strideRes
    d.accumulate = int(accumulate)
    d.tile = tile
    d.tag = tag
    d.precision = _cur_gemm() if precision is None else precision
    if splitk is None:
        # launches that cannot fill the chip (<= 128 output tiles: the embed fold's T small products, the cls head,
        # pos-embed terms) are latency bound - spread their K over more workgroups; big grids stay un-split
        small = ((M + 63) // 64) * ((N + 63) // 64) * batch <= 128
        splitk = auto_splitk(M, N, K, batch, precision=d.precision) if small else 1
    d.colsum_out, d.colsum_of, d.strideColsum = _p(colsum_out), (colsum_of if colsum_out is not None else 0), strideColsum
    ws = None
    if splitk > 1:
        L = M if d.colsum_of == 1 else N if d.colsum_of == 2 else 0
        ws = torch.empty(splitk * batch * (M * N + L), dtype=torch.float32, device=A.device)
        d.splitk, d.workspace = splitk, ws.data_ptr()
    else:
        d.splitk, d.workspace = 1, None
    check(lib.dpot_gemm_f32(C.byref(d), _stream()), "gemm_f32")
    return C_


# ------------------------------------------------------------------------------------------------------
# panel GEMM with pre-packed static weights (csrc/gemm_panel.hip)
# ------------------------------------------------------------------------------------------------------
def panel_enabled() -> bool:
    """the panel kernel is an fp32 kernel: used while the global GEMM precision is 'f32' (the channel-MLP override of
    set_mlp_precision is checked where it applies, functional._mlp_panel_ok)"""
    return tune("panel") != 0 and _cur_gemm() == GEMM_F32


def gemm_panel_supported(M: int, N: int, K: int) -> bool:
    return bool(_lib.load().dpot_gemm_panel_supported(M, N, K))


class PanelPacks:
    """Fragment-block-major copies of a set of static weights, refreshed by ONE launch (dpot_panel_pack_weights) from a
    job table that lives in device memory.  jobs = [(src tensor, rows, K, ld, trans)]; the destination buffers are
    allocated once, so the table stays valid while the sources do not move."""

    def __init__(self, jobs, bf16: bool = False, planes: int = 1):
        import numpy as np
        dev = jobs[0][0].device
        self.bf16 = bf16
        self.planes = planes if bf16 else 0       # 1: plain bf16, 3: bf16x6 split (fp32-accurate)
        if bf16:       # csrc/gemm_bf16p.hip: rows padded to 32, bf16 elements, `planes` planes per 16-k block
            lib = _lib.load()
            self.bufs = [torch.empty(lib.dpot_bf16_packed_elems(rows, K, planes), dtype=torch.bfloat16, device=dev)
                         for _, rows, K, _, _ in jobs]
        else:
            self.bufs = [torch.empty(rows * K, dtype=torch.float32, device=dev) for _, rows, K, _, _ in jobs]
        self.key = tuple(j[0].data_ptr() for j in jobs)
        host = np.zeros(len(jobs) * C.sizeof(_lib.PackJob), dtype=np.uint8)
        tab = (_lib.PackJob * len(jobs)).from_buffer(host)
        for i, ((src, rows, K, ld, trans), dst) in enumerate(zip(jobs, self.bufs)):
            tab[i].src, tab[i].dst, tab[i].rows, tab[i].K, tab[i].ld, tab[i].trans = src.data_ptr(), dst.data_ptr(), rows, K, ld, int(trans)
        self.table = torch.from_numpy(host).to(dev)
        self.n = len(jobs)
        self.max_elems = max(rows * K for _, rows, K, _, _ in jobs)
        self.jobs = list(jobs)                    # (the table holds raw pointers: this also keeps the sources alive)
        # round 6: an optimiser that writes these packs itself (train.FusedAdam -> dpot_adam_step_packs) records here what the
        # packs are fresh FOR - (its FlatParams, that buffer's epoch, the sources' tensor versions); refresh() is then a no-op
        # until anything else moves the parameters (FlatParams.epoch bumps, load_state_dict / in-place writes bump _version)
        self.fresh_for = None

    def _fresh_key(self, fp):
        return (id(fp), fp.epoch, tuple(j[0]._version for j in self.jobs[::2]))

    def is_fresh(self) -> bool:
        ff = self.fresh_for
        return ff is not None and ff[1] == self._fresh_key(ff[0])

    def mark_fresh(self, fp) -> None:
        self.fresh_for = (fp, self._fresh_key(fp))

    def refresh(self, force: bool = False) -> None:
        if not force and self.is_fresh():
            return
        lib = _lib.load()
        if self.bf16:
            check(lib.dpot_bf16_pack_jobs(self.table.data_ptr(), self.n, self.max_elems, self.planes, _stream()),
                  "bf16_pack_jobs")
        else:
            check(lib.dpot_panel_pack_weights(self.table.data_ptr(), self.n, self.max_elems, _stream()), "pack_weights")


class LayoutJobs:
    """Small weight-only layout products (zero-padded copies, small transposes, bias broadcasts, "+ bias") refreshed by ONE
    launch (dpot_layout_jobs) from a device-resident table.  jobs = [(src, add | None, (d0, d1, d2), (v0, v1, v2),
    (s0, s1, s2))]: out[i0, i1, i2] = (inside v ? src.flat[i0 s0 + i1 s1 + i2 s2] : 0) + (add[i2] if add is not None).
    self.out[i]: persistent [d0, d1, d2] tensors (the table stays valid while the sources do not move: self.key)."""

    def __init__(self, jobs):
        import numpy as np
        dev = jobs[0][0].device
        self.out = [torch.empty(d, dtype=torch.float32, device=dev) for _, _, d, _, _ in jobs]
        self.key = LayoutJobs.key_of(jobs)
        self.sources = [(j[0], j[1]) for j in jobs]        # the table holds raw pointers: keep the tensors alive
        host = np.zeros(len(jobs) * C.sizeof(_lib.LayoutJob), dtype=np.uint8)
        tab = (_lib.LayoutJob * len(jobs)).from_buffer(host)
        for i, ((src, add, d, v, st), dst) in enumerate(zip(jobs, self.out)):
            t = tab[i]
            t.src, t.add, t.dst = src.data_ptr(), (add.data_ptr() if add is not None else None), dst.data_ptr()
            t.d0, t.d1, t.d2 = d
            t.v0, t.v1, t.v2 = v
            t.s0, t.s1, t.s2 = st
        self.table = torch.from_numpy(host).to(dev)
        self.n = len(jobs)
        self.max_elems = max(d[0] * d[1] * d[2] for _, _, d, _, _ in jobs)

    @staticmethod
    def key_of(jobs):
        """every raw pointer the device table would hold (src AND add of each job): the cache key of a LayoutJobs"""
        return tuple(j[0].data_ptr() for j in jobs) + tuple(j[1].data_ptr() for j in jobs if j[1] is not None)

    def refresh(self):
        check(_lib.load().dpot_layout_jobs(self.table.data_ptr(), self.n, self.max_elems, _stream()), "layout_jobs")
        return self.out


def gemm_panel(A: Tensor, Wpacked: Tensor, N: int, *, bias: Optional[Tensor] = None, act: int = 0,
               mode: int = EPI_LINEAR, aux: Optional[Tensor] = None, res: Optional[Tensor] = None,
               save_pre: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
    """C[M,N] = epilogue(A[M,K] @ Wt^T), Wt packed by PanelPacks (rows = N).  Returns (C, pre | None)."""
    M, K = A.shape
    out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    pre = torch.empty_like(out) if save_pre else None
    check(_lib.load().dpot_gemm_panel(A.data_ptr(), A.stride(0), Wpacked.data_ptr(), _p(bias), _p(aux),
                                      aux.stride(0) if aux is not None else 0, _p(res),
                                      res.stride(0) if res is not None else 0, _p(pre), N, out.data_ptr(), N, M, N, K,
                                      act, mode, _stream()), "gemm_panel")
    return out, pre


def gemm_bf16p_supported(M: int, N: int, K: int) -> bool:
    return bool(_lib.load().dpot_gemm_bf16p_supported(M, N, K))


class MlpPacks(tuple):
    """(W1, W1^T, W2, W2^T) packed for one of the pre-packed-weight GEMM kernels; kind = 'f32' (csrc/gemm_panel.hip),
    'bf16' (csrc/gemm_bf16p.hip, 1 plane: reduced precision) or 'bf16x6' (3 planes: fp32-accurate)"""

    def __new__(cls, bufs, kind: str):
        self = super().__new__(cls, bufs)
        self.kind = kind
        return self

    @property
    def planes(self) -> int:
        return {"f32": 0, "bf16": 1, "bf16x6": 3}[self.kind]


def bf16_pack_rows(x: Tensor, trans: bool = False, planes: int = 1) -> Tensor:
    """fp32 [M, K] -> bf16 fragment-block-major operand of gemm_bf16p (one pass; rows padded to 32 with zeros).
    trans: pack x^T instead (x stored [K, rows]: weight gradients - rows = features, k = tokens);
    planes = 3: the three-plane bf16x6 split"""
    lib = _lib.load()
    if trans:
        K, M = x.shape
    else:
        M, K = x.shape
    out = torch.empty(lib.dpot_bf16_packed_elems(M, K, planes), dtype=torch.bfloat16, device=x.device)
    check(lib.dpot_bf16_pack_rows(x.data_ptr(), x.stride(0), M, K, int(trans), planes, out.data_ptr(), _stream()),
          "bf16_pack_rows")
    return out


def bf16_pack_both_supported(rows: int, K: int) -> bool:
    return bool(_lib.load().dpot_bf16_pack_both_supported(rows, K))


def bf16_pack_both(x: Tensor, want_rows: bool = True, want_trans: bool = True, colsum_out: Optional[Tensor] = None,
                   want_colsum: bool = False, norm=None, defer_colsum: bool = False):
    """one pass over x [M, K] fp32 -> (row-form pack | None, transposed pack | None, column sums | None): the two operand
    forms the bf16 channel MLP needs of an activation (data GEMM and weight gradient) and its bias gradient.
    norm = (mean [B,G], rstd [B,G], gamma [K], beta [K], rows_per_sample): the packs of GroupNorm(x) with those statistics
    (rows_per_sample % 64 == 0), applied on the load - no column sums in that form"""
    lib = _lib.load()
    M, K = x.shape
    pr = torch.empty(lib.dpot_bf16_packed_elems(M, K, 1), dtype=torch.bfloat16, device=x.device) if want_rows else None
    pt = torch.empty(lib.dpot_bf16_packed_elems(K, M, 1), dtype=torch.bfloat16, device=x.device) if want_trans else None
    if norm is not None:
        mean, rstd, gamma, beta, rps = norm
        assert not want_colsum
        check(lib.dpot_bf16_pack_both_norm(x.data_ptr(), x.stride(0), M, K, mean.data_ptr(), rstd.data_ptr(),
                                           gamma.data_ptr(), beta.data_ptr(), rps, mean.shape[1], _p(pr), _p(pt), _stream()),
              "bf16_pack_both_norm")
        return pr, pt, None
    part = torch.empty(M // 64, K, dtype=torch.float32, device=x.device) if want_colsum else None
    check(lib.dpot_bf16_pack_both(x.data_ptr(), x.stride(0), M, K, _p(pr), _p(pt), _p(part), _stream()), "bf16_pack_both")
    if want_colsum and defer_colsum:      # the column sums are left as partials: (part, rows, N, out) for `block_finalize`
        return pr, pt, (part, M // 64, K, _out(colsum_out, (K,), x.device))
    cs = colsum(part, M // 64, K, out=colsum_out) if want_colsum else None
    return pr, pt, cs


def gemm_bf16p(Ap: Tensor, Wp: Tensor, M: int, N: int, K: int, *, bias: Optional[Tensor] = None, act: int = 0,
               mode: int = EPI_LINEAR, aux: Optional[Tensor] = None, res: Optional[Tensor] = None,
               save_pre: bool = False, out: Optional[Tensor] = None, splitk: Optional[int] = None,
               planes: int = 1) -> Tuple[Tensor, Optional[Tensor]]:
    """C[M,N] fp32 = epilogue(A @ Wt^T) on the bf16 matrix cores; Ap / Wp: packed bf16 operands (bf16_pack_rows, or a
    bf16 PanelPacks buffer) with the same number of planes (1: plain bf16, 3: bf16x6 = fp32-accurate).
    splitk=None: the library's choice (weight gradients use split-K).  Returns (C, pre | None)."""
    C_, pre, _, _, _ = gemm_bf16p_packed(Ap, Wp, M, N, K, bias=bias, act=act, mode=mode, aux=aux, res=res,
                                         save_pre=save_pre, out=out, splitk=splitk, planes=planes)
    return C_, pre


def gemm_bf16p_pair_wanted(M0: int, N0: int, M1: int, N1: int, K: int) -> bool:
    return bool(_lib.load().dpot_gemm_bf16p_pair_wanted(M0, N0, M1, N1, K))


BF16P_KERNEL_KINDS = {0: "dpot::gemm_bf16p_kernel (8 compute + 4 loader waves, LDS-DMA)",
                      1: "dpot::gemm_bf16p_duo_kernel (two 8-wave workgroups per CU)",
                      2: "dpot::gemm_bf16p_bd_kernel<8,1,3> (B-direct: W fragments straight from global memory into registers, "
                         "only the A panel through LDS; eight 128 x 32 waves)",
                      3: "dpot::gemm_bf16p_bd_kernel<8,2,3> (B-direct: W fragments straight from global memory into registers, "
                         "only the A panel through LDS; four 128 x 64 waves, two workgroups per CU)",
                      4: "dpot::gemm_bf16x6p_kernel (fp32-accurate three-plane split)"}


def gemm_bf16p_kernel_name(M: int, N: int, K: int, splitk: int = 1, planes: int = 1, packed_outputs: bool = False) -> str:
    """the kernel dpot_gemm_bf16p runs for this shape, from the library's own selection (dpot_gemm_bf16p_kernel_kind)"""
    k = _lib.load().dpot_gemm_bf16p_kernel_kind(M, N, K, splitk, planes, int(packed_outputs))
    return BF16P_KERNEL_KINDS.get(k & 7, f"kind {k}") + (" on 128 x 192 tiles" if k >= 8 else "")


def gemm_bf16p_pair(A0: Tensor, W0: Tensor, M0: int, N0: int, A1: Tensor, W1: Tensor, M1: int, N1: int, K: int,
                    out0: Optional[Tensor] = None, out1: Optional[Tensor] = None,
                    splitk: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """(A0 W0^T [M0,N0], A1 W1^T [M1,N1]) on the bf16 matrix cores in ONE launch (packed plain-bf16 operands, common K,
    one common split-K factor - None: the library's choice): the two channel-MLP weight gradients of a block."""
    lib = _lib.load()
    C0 = _out(out0, (M0, N0), A0.device)
    C1 = _out(out1, (M1, N1), A1.device)
    sk = max(1, lib.dpot_gemm_bf16p_pair_splitk(M0, N0, M1, N1, K)) if splitk is None else splitk
    ws = torch.empty(sk * (M0 * N0 + M1 * N1), dtype=torch.float32, device=A0.device) if sk > 1 else None
    check(lib.dpot_gemm_bf16p_pair(A0.data_ptr(), W0.data_ptr(), C0.data_ptr(), N0, M0, N0, A1.data_ptr(), W1.data_ptr(),
                                   C1.data_ptr(), N1, M1, N1, K, sk, _p(ws), _stream()), "gemm_bf16p_pair")
    return C0, C1


def gemm_bf16p_packed(Ap: Tensor, Wp: Tensor, M: int, N: int, K: int, *, bias: Optional[Tensor] = None, act: int = 0,
                      mode: int = EPI_LINEAR, aux: Optional[Tensor] = None, res: Optional[Tensor] = None,
                      save_pre: bool = False, out: Optional[Tensor] = None, splitk: Optional[int] = None,
                      planes: int = 1, pack_rows: bool = False, pack_trans: bool = False, colsum: bool = False,
                      colsum_out: Optional[Tensor] = None, store: bool = True, save_dact: bool = False,
                      dact: Optional[Tensor] = None, defer_colsum: bool = False):
    """gemm_bf16p whose epilogue can ALSO emit the packed forms of the output and its column sums
    (pack_rows / pack_trans / colsum: planes == 1, M % 32 == 0, no split-K); store=False skips the fp32 output.
    save_dact (EPI_ACT): instead of the fp32 pre-activation, the second return value is act'(pre) as a packed bf16
    tensor - pass it as `dact` to the EPI_DACT launch of the backward (replaces aux).
    Always returns (C | None, pre or act' pack | None, row pack | None, transposed pack | None, column sums | None)."""
    lib = _lib.load()
    packs = pack_rows or pack_trans or colsum or save_dact or dact is not None
    C_ = _out(out, (M, N), Ap.device) if (store or not packs) else None
    pre = torch.empty(M, N, dtype=torch.float32, device=Ap.device) if save_pre else None
    if splitk is None:
        splitk = 1 if packs else lib.dpot_gemm_bf16p_splitk(M, N, K)
    ws = torch.empty(splitk * M * N, dtype=torch.float32, device=Ap.device) if splitk > 1 else None
    pr = torch.empty(lib.dpot_bf16_packed_elems(M, N, 1), dtype=torch.bfloat16, device=Ap.device) if pack_rows else None
    pt = torch.empty(lib.dpot_bf16_packed_elems(N, M, 1), dtype=torch.bfloat16, device=Ap.device) if pack_trans else None
    part = torch.empty(M // 32, N, dtype=torch.float32, device=Ap.device) if colsum else None
    dout = torch.empty(lib.dpot_bf16_packed_elems(M, N, 1), dtype=torch.bfloat16, device=Ap.device) if save_dact else None
    check(lib.dpot_gemm_bf16p(Ap.data_ptr(), Wp.data_ptr(), _p(bias), _p(aux),
                              aux.stride(0) if aux is not None else 0, _p(res),
                              res.stride(0) if res is not None else 0, _p(pre), N, _p(C_), N, M, N, K,
                              act, mode, planes, splitk, _p(ws), _p(pr), _p(pt), _p(part), _p(dout), _p(dact),
                              _stream()), "gemm_bf16p")
    if colsum and defer_colsum:           # (part, rows, N, out) for `block_finalize`
        cs = (part, M // 32, N, _out(colsum_out, (N,), Ap.device))
    else:
        cs = globals()["colsum"](part, M // 32, N, out=colsum_out) if colsum else None
    return C_, (dout if save_dact else pre), pr, pt, cs


def small_linear_supported(M: int, N: int, K: int) -> bool:
    return bool(_lib.load().dpot_small_linear_supported(M, N, K))


def linear_fwd(x: Tensor, W: Tensor, bias: Optional[Tensor], act: int = 0, save_pre: bool = False,
               res: Optional[Tensor] = None, res_div: int = 0, res_mod: int = 0,
               ldw: Optional[int] = None, precision: Optional[int] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """y[M,N] = act(x[M,K] @ W[N,K]^T + bias) (+ res)   -  nn.Linear / 1x1-conv semantics."""
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    pre = torch.empty_like(y) if save_pre else None
    lib = _lib.load()
    prec = _cur_gemm() if precision is None else precision
    if (res is None and M <= 128 and prec != GEMM_BF16 and lib.dpot_small_linear_supported(M, N, K)
            and tune("fused_small") != 0):
        # a handful of rows (the cls_head on the token mean): one pass over W, no split-K / reduce pair; exact fp32
        # (not under the opt-in bf16 operand rounding: that mode keeps its rounding at every batch size)
        check(lib.dpot_small_linear(x.data_ptr(), x.stride(0), W.data_ptr(), ldw or W.stride(0), _p(bias), y.data_ptr(),
                                    _p(pre), N, M, N, K, act, _stream()), "small_linear")
        return y, pre
    gemm(x, W, y, M, N, K, transB=True, lda=x.stride(0), ldb=ldw or W.stride(0), ldc=N, bias=bias, act=act,
         mode=EPI_ACT if act else EPI_LINEAR, preact=pre, ldpre=N, res=res, ldres=N, res_div=res_div,
         res_mod=res_mod, precision=precision)
    return y, pre


def linear_bwd_data(dy: Tensor, W: Tensor, act: int = 0, aux: Optional[Tensor] = None,
                    precision: Optional[int] = None) -> Tensor:
    """dx[M,K] = dy[M,N] @ W[N,K]  (optionally times act'(aux[M,K]))."""
    M, N = dy.shape
    K = W.shape[1]
    dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
    gemm(dy, W, dx, M, K, N, transB=False, lda=dy.stride(0), ldb=W.stride(0), ldc=K, act=act,
         mode=EPI_DACT if aux is not None else EPI_LINEAR, aux=aux, ldaux=K, precision=precision)
    return dx


def _out(out: Optional[Tensor], shape, device) -> Tensor:
    """`out` (a gradient slot inside the flat gradient buffer, see train.FlatParams) or a fresh tensor"""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    return out.view(shape)


def linear_bwd_weight(dy: Tensor, x: Tensor, out: Optional[Tensor] = None, n_rows: Optional[int] = None,
                      bias_out: Optional[Tensor] = None, precision: Optional[int] = None) -> Tensor:
    """dW[N,K] = dy[M,N]^T @ x[M,K]   (split-K over the token dimension); n_rows < N computes only dW[:n_rows].
    bias_out [N]: also db = colsum(dy), produced by the same kernel from the dy tiles it stages anyway."""
    M, N = dy.shape
    N = n_rows or N
    K = x.shape[1]
    dW = _out(out, (N, K), dy.device)
    gemm(dy, x, dW, N, K, M, transA=True, transB=False, lda=dy.stride(0), ldb=x.stride(0), ldc=K,
         splitk=auto_splitk(N, K, M, precision=precision, tn=True), colsum_out=bias_out, colsum_of=1,
         precision=precision)
    return dW


def linear_bwd_wb(dy: Tensor, x: Tensor, out_w: Optional[Tensor] = None, out_b: Optional[Tensor] = None,
                  precision: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """(dW, db) of y = x W^T + b in one launch (+ the split-K reduction): db rides on the dy tiles of the wgrad GEMM"""
    db = _out(out_b, (dy.shape[1],), dy.device)
    return linear_bwd_weight(dy, x, out=out_w, bias_out=db, precision=precision), db


# ------------------------------------------------------------------------------------------------------
# spectral ops
# ------------------------------------------------------------------------------------------------------
def rfft2_norm_supported(h: int, w: int, E: int) -> bool:
    return bool(_lib.load().dpot_rfft2_norm_supported(h, w, E))


def rfft2(x: Tensor, h: int, w: int, nb: int, mx: int, my: int, col_weights: int = 0, norm=None) -> Tensor:
    """x[B,h*w,E] -> spec[B*mx*my, 2E] (planar per channel block).
    norm = (mean [B,G], rstd [B,G], gamma [E], beta [E]): the transform of GroupNorm(x) with those statistics, applied on
    the load (rfft2_norm_supported grids)"""
    B, E = x.shape[0], x.shape[-1]
    spec = torch.empty(B * mx * my, 2 * E, dtype=torch.float32, device=x.device)
    if norm is not None:
        mean, rstd, gamma, beta = norm
        check(_lib.load().dpot_rfft2_norm(x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                          mean.shape[1], spec.data_ptr(), B, h, w, E, nb, mx, my, col_weights, _stream()),
              "rfft2_norm")
        return spec
    check(_lib.load().dpot_rfft2(x.data_ptr(), spec.data_ptr(), B, h, w, E, nb, mx, my, col_weights, _stream()),
          "rfft2")
    return spec


def irfft2(spec: Tensor, B: int, h: int, w: int, E: int, nb: int, mx: int, my: int, col_weights: int = 1,
           res: Optional[Tensor] = None, res_norm=None) -> Tensor:
    """res_norm = (mean, rstd, gamma, beta): the residual added is GroupNorm(res) with those statistics"""
    y = torch.empty(B, h * w, E, dtype=torch.float32, device=spec.device)
    if res_norm is not None:
        mean, rstd, gamma, beta = res_norm
        check(_lib.load().dpot_irfft2_norm(spec.data_ptr(), res.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                           gamma.data_ptr(), beta.data_ptr(), mean.shape[1], y.data_ptr(), B, h, w, E, nb,
                                           mx, my, col_weights, _stream()), "irfft2_norm")
        return y
    check(_lib.load().dpot_irfft2(spec.data_ptr(), _p(res), y.data_ptr(), B, h, w, E, nb, mx, my, col_weights,
                                  _stream()), "irfft2")
    return y


def afno_pack(w: Tensor, b: Tensor) -> Tuple[Tensor, Tensor]:
    _, nb, bs, _ = w.shape
    wbig = torch.empty(nb, 2 * bs, 2 * bs, dtype=torch.float32, device=w.device)
    bbig = torch.empty(nb, 2 * bs, dtype=torch.float32, device=w.device)
    check(_lib.load().dpot_afno_pack(w.data_ptr(), b.data_ptr(), wbig.data_ptr(), bbig.data_ptr(), nb, bs,
                                     _stream()), "afno_pack")
    return wbig, bbig


def afno_pack3(w: Tensor, b: Tensor):
    """(Wbig, bbig, blocked W | None, blocked W^T | None): the per-layer form of what afno_pack_multi returns"""
    _, nb, bs, _ = w.shape
    wbig, bbig = afno_pack(w, b)
    if afno_mlp2_supported(nb, bs):
        wf, wb = afno_block_weights(wbig)
        return wbig, bbig, wf, wb
    return wbig, bbig, None, None


def afno_pack_multi(pairs) -> list:
    """pairs = [(w [2,nb,bs,bs], b [2,nb,bs]), ...] of equal shapes -> [(wbig, bbig), ...], ONE launch for all of them"""
    n = len(pairs)
    _, nb, bs, _ = pairs[0][0].shape
    dev = pairs[0][0].device
    wbig = torch.empty(n, nb, 2 * bs, 2 * bs, dtype=torch.float32, device=dev)
    bbig = torch.empty(n, nb, 2 * bs, dtype=torch.float32, device=dev)
    arr = lambda ptrs: (C.c_void_p * n)(*ptrs)
    check(_lib.load().dpot_afno_pack_multi(arr([_req(w, "w").data_ptr() for w, _ in pairs]),
                                           arr([_req(b, "b").data_ptr() for _, b in pairs]),
                                           arr([wbig[i].data_ptr() for i in range(n)]),
                                           arr([bbig[i].data_ptr() for i in range(n)]), n, nb, bs, _stream()),
          "afno_pack_multi")
    # fragment-block-major copies for the fused 2-layer kernel (afno_mlp2): one launch for all layers
    if afno_mlp2_supported(nb, bs):
        wf, wb = afno_block_weights(wbig.view(n * nb, 2 * bs, 2 * bs))
        wf, wb = wf.view(n, nb, 2 * bs, 2 * bs), wb.view(n, nb, 2 * bs, 2 * bs)
        return [(wbig[i], bbig[i], wf[i], wb[i]) for i in range(n)]
    return [(wbig[i], bbig[i], None, None) for i in range(n)]


class AfnoItem(tuple):
    """(Wbig, bbig, fwd pack | None, bwd pack | None) of one AFNO layer + the layout of the packs (0: fragment-block-major
    Wbig for afno_mlp2's four-product kernel, 1: (Wr, Wi) fragments for the three-product kernel)"""

    def __new__(cls, items, layout: int = 0, p6=None):
        self = super().__new__(cls, items)
        self.layout = layout
        self.p6 = p6          # (fwd6, bwd6): the bf16x6 packs of csrc/afno_mlp6.hip (layout 2 of afno_mlp2), or None
        return self


class AfnoPacks:
    """Packed forms of ALL AFNO layers of a model, refreshed by ONE launch (dpot_afno_pack_all) from a device-resident
    job table: persistent output buffers, so the table is built once per parameter placement.
    pairs = [(w [2,nb,bs,bs], b [2,nb,bs]), ...];  self.items[i] = (wbig, bbig, fwd | None, bwd | None)"""

    def __init__(self, pairs):
        import numpy as np
        n = len(pairs)
        _, nb, bs, _ = pairs[0][0].shape
        dev = pairs[0][0].device
        N = 2 * bs
        self.nb, self.bs, self.n = nb, bs, n
        self.key = tuple(t.data_ptr() for p in pairs for t in p)
        fused = afno_mlp2_supported(nb, bs)
        self.layout = 1 if fused and afno_mlp3_supported(nb, bs) else 0      # three-product (Wr, Wi) fragment packs
        wbig = torch.empty(n, nb, N, N, dtype=torch.float32, device=dev)
        bbig = torch.empty(n, nb, N, dtype=torch.float32, device=dev)
        fwd = torch.empty_like(wbig) if fused else None
        bwd = torch.empty_like(wbig) if fused else None
        host = np.zeros(n * C.sizeof(_lib.AfnoPackJob), dtype=np.uint8)
        tab = (_lib.AfnoPackJob * n).from_buffer(host)
        for i, (w, b) in enumerate(pairs):
            tab[i].w, tab[i].b = _req(w, "w").data_ptr(), _req(b, "b").data_ptr()
            tab[i].wbig, tab[i].bbig = wbig[i].data_ptr(), bbig[i].data_ptr()
            tab[i].fwd = fwd[i].data_ptr() if fused else None
            tab[i].bwd = bwd[i].data_ptr() if fused else None
        self.table = torch.from_numpy(host).to(dev)
        # bf16x6 packs (csrc/afno_mlp6.hip): items alternate (first-layer weight, second-layer weight) - pairs of a filter
        self.wbig = wbig
        self.fwd6 = self.bwd6 = None               # allocated by the first refresh that wants them (3 bytes per weight element
        self._p6_ok = fused and n % 2 == 0 and afno_mlp6_supported(nb, bs)   # each: nothing for a model that stays in 'f32')
        self._fused = fused
        self._base = [(wbig[i], bbig[i], fwd[i] if fused else None, bwd[i] if fused else None) for i in range(n)]
        self.items = [AfnoItem(t, self.layout, None) for t in self._base]

    def refresh(self):
        lib = _lib.load()
        check(lib.dpot_afno_pack_all(self.table.data_ptr(), self.n, self.nb, self.bs, self.layout, _stream()),
              "afno_pack_all")
        if self._p6_ok and afno_mlp6_wanted():
            if self.fwd6 is None:
                ne = int(lib.dpot_afno_pack6_elems(self.nb, self.bs))
                dev = self.wbig.device
                self.fwd6 = torch.empty(self.n, ne, dtype=torch.int16, device=dev)
                self.bwd6 = torch.empty(self.n, ne, dtype=torch.int16, device=dev)
                self.items = [AfnoItem(t, self.layout, (self.fwd6[i], self.bwd6[i])) for i, t in enumerate(self._base)]
            check(lib.dpot_afno_pack6(self.wbig.data_ptr(), self.fwd6.data_ptr(), self.bwd6.data_ptr(), self.n, self.nb,
                                      self.bs, _stream()), "afno_pack6")
        return self.items


def afno_block_weights(wbig: Tensor) -> Tuple[Tensor, Tensor]:
    """wbig [J, N, N] (W[k][n]) -> (fwd, bwd): fragment-block-major W and W^T (dpot_afno_block_weights)"""
    J, N, _ = wbig.shape
    fwd, bwd = torch.empty_like(wbig), torch.empty_like(wbig)
    check(_lib.load().dpot_afno_block_weights(wbig.data_ptr(), fwd.data_ptr(), bwd.data_ptr(), J, N, _stream()),
          "afno_block_weights")
    return fwd, bwd


def mlp_wgrad2_splitk(T: int, E: int, mh: int, precision: Optional[int] = None) -> int:
    """split factor of the fused fc1 + fc2 weight gradient (0: not covered / not worthwhile -> two separate GEMMs)"""
    prec = _cur_gemm() if precision is None else precision
    return _lib.load().dpot_mlp_wgrad2_splitk(T, E, mh) if prec == GEMM_F32 else 0


def mlp_wgrad2(do2: Tensor, Hh: Tensor, xn2: Tensor, dHpre: Tensor, dW2: Tensor, db2: Tensor, dW1: Tensor, db1: Tensor,
               splitk: int, defer: bool = False):
    """both weight gradients of a channel MLP, one launch + one reduce.  defer=True: the launch writes its split-K partials
    only and the job (for `block_finalize`, which reduces it together with the block's other partials) is returned"""
    lib = _lib.load()
    T, E = do2.shape
    mh = Hh.shape[1]
    ws = torch.empty(lib.dpot_mlp_wgrad2_ws_elems(E, mh, splitk), dtype=torch.float32, device=do2.device)
    outs = (None,) * 4 if defer else (dW2.data_ptr(), db2.data_ptr(), dW1.data_ptr(), db1.data_ptr())
    check(lib.dpot_mlp_wgrad2(do2.data_ptr(), Hh.data_ptr(), xn2.data_ptr(), dHpre.data_ptr(), T, E, mh, *outs,
                              ws.data_ptr(), splitk, _stream()), "mlp_wgrad2")
    return (ws, splitk, E, mh, dW2, db2, dW1, db1) if defer else None


def afno_wgrad2_splitk(Mm: int, nb: int, bs: int) -> int:
    """split factor of the fused two-layer AFNO weight gradient (0: shape not covered -> two generic GEMM launches)"""
    return _lib.load().dpot_afno_wgrad2_splitk(Mm, nb, bs) if _cur_gemm() in (GEMM_F32, GEMM_AUTO) else 0


def afno_wgrad2(S: Tensor, dO1pre: Tensor, O1: Tensor, dO2: Tensor, nb: int, bs: int, dw1: Tensor, db1: Tensor,
                dw2: Tensor, db2: Tensor, splitk: int, defer: bool = False):
    """dw1 / db1 (from S, dO1pre) and dw2 / db2 (from O1, dO2) of an AFNO block's complex MLP: one launch + one reduce.
    defer=True: partials only; returns the job for `block_finalize`"""
    lib = _lib.load()
    Mm, ld = S.shape
    ws = torch.empty(lib.dpot_afno_wgrad2_ws_elems(nb, bs, splitk), dtype=torch.float32, device=S.device)
    outs = (None,) * 4 if defer else (dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr())
    check(lib.dpot_afno_wgrad2(S.data_ptr(), dO1pre.data_ptr(), O1.data_ptr(), dO2.data_ptr(), ld, Mm, nb, bs, *outs,
                               ws.data_ptr(), splitk, _stream()), "afno_wgrad2")
    return (ws, splitk, nb, bs, dw1, db1, dw2, db2) if defer else None


def block_finalize(afno_job, mlp_job, gn_jobs, cs_jobs=()):
    """ONE launch for the reductions that end a DPOT block's backward (csrc/gemm_tn.hip block_finalize_kernel): the deferred
    split-K partials of `afno_wgrad2` and / or `mlp_wgrad2` (jobs as those return them, or None) and the GroupNorm
    parameter-gradient partials gn_jobs = [(part [2,B,E], out_dgamma | None, out_dbeta | None)] (<= 2), and the deferred
    column sums cs_jobs of `bf16_pack_both` / `gemm_bf16p_packed` (defer_colsum=True; <= 2).  Returns the GroupNorm outputs
    [(dgamma, dbeta)]; the other outputs are the tensors handed to / returned by the deferred calls."""
    null = None
    a = afno_job if afno_job is not None else (None, 0, 0, 0, None, None, None, None)
    m = mlp_job if mlp_job is not None else (None, 0, 0, 0, None, None, None, None)
    n = len(gn_jobs)
    outs, B, E = [], 0, 0
    parts = dgs = dbs = null
    if n:
        _, B, E = gn_jobs[0][0].shape
        outs = [(_out(og, (E,), p.device), _out(ob, (E,), p.device)) for p, og, ob in gn_jobs]
        parts = (C.c_void_p * n)(*[p.data_ptr() for p, _, _ in gn_jobs])
        dgs = (C.c_void_p * n)(*[g.data_ptr() for g, _ in outs])
        dbs = (C.c_void_p * n)(*[b.data_ptr() for _, b in outs])
    nc = len(cs_jobs)                              # [(part [rows, N], rows, N, out [N])] (<= 2): deferred column sums
    cparts = couts = crows = ccols = null
    if nc:
        cparts = (C.c_void_p * nc)(*[j[0].data_ptr() for j in cs_jobs])
        couts = (C.c_void_p * nc)(*[j[3].data_ptr() for j in cs_jobs])
        crows = (C.c_int * nc)(*[j[1] for j in cs_jobs])
        ccols = (C.c_int * nc)(*[j[2] for j in cs_jobs])
    check(_lib.load().dpot_block_finalize(_p(a[0]), a[1], a[2], a[3], _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]),
                                          _p(m[0]), m[1], m[2], m[3], _p(m[4]), _p(m[5]), _p(m[6]), _p(m[7]),
                                          parts, dgs, dbs, n, B, E, cparts, couts, crows, ccols, nc, _stream()),
          "block_finalize")
    return outs


def block_finalize_enabled() -> bool:
    return tune("fused_small") != 0


def afno_mlp2_supported(nb: int, bs: int) -> bool:
    return bool(_lib.load().dpot_afno_mlp2_supported(nb, bs)) and tune("mixer") != 0


def afno_mlp3_supported(nb: int, bs: int) -> bool:
    """the three-product (Gauss / Karatsuba) form of the fused mixer kernel: bs == 128; DPOT_AFNO_3MULT=0 disables it"""
    return bool(_lib.load().dpot_afno_mlp3_supported(nb, bs)) and tune("mixer") == 3


def afno_mlp6_supported(nb: int, bs: int) -> bool:
    """the bf16x6 (fp32-accurate, bf16 matrix cores) form of the fused mixer kernel, csrc/afno_mlp6.hip: bs in {96, 128}"""
    return bool(_lib.load().dpot_afno_mlp6_supported(nb, bs)) and tune("mixer") != 0 and (tune("mixer6") == 2 or (tune("mixer6") == 1 and bs == 96))


def afno_mlp6_wanted() -> bool:
    """the mixer follows the GEMM precision in effect: 'auto' / 'bf16x6' -> the bf16x6 kernel (both are fp32-accurate, the
    choice is speed); 'f32' -> the native fp32 matrix-core kernels"""
    return _cur_gemm() in (GEMM_AUTO, GEMM_BF16X6)


def afno_mlp2(X: Tensor, WaT: Tensor, ba: Optional[Tensor], WbT: Tensor, bb: Optional[Tensor], nb: int, bs: int,
              act: int, mode: int = 0, aux: Optional[Tensor] = None, want_pre: bool = False, want_mid: bool = False,
              layout: int = 0):
    """both layers of the AFNO block-diagonal complex MLP in one launch (csrc/afno_mlp.hip).
    mode 0: pre = X Wa + ba, mid = act(pre), Y = mid Wb + bb;  mode 1: mid = (X Wa) * act'(aux), Y = mid Wb, and with
    want_pre the `pre` output is act(aux) (the forward's activated layer-1 output, re-derived for the weight gradient).
    X / outputs: [M, nb*2*bs]; WaT / WbT: [nb, 2bs, 2bs] fragment-block-major weights from afno_block_weights (`fwd`
    to multiply by W, `bwd` to multiply by W^T).  Returns (Y, pre | None, mid | None)."""
    M, ld = X.shape
    Y = torch.empty_like(X)
    pre = torch.empty_like(X) if want_pre else None
    mid = torch.empty_like(X) if want_mid else None
    if layout == 2:      # WaT / WbT: the bf16x6 packs of AfnoPacks (item.p6), csrc/afno_mlp6.hip
        check(_lib.load().dpot_afno_mlp6(X.data_ptr(), WaT.data_ptr(), _p(ba), WbT.data_ptr(), _p(bb), _p(aux), _p(pre),
                                         _p(mid), Y.data_ptr(), M, nb, bs, ld, ld, act, mode, _stream()), "afno_mlp6")
        return Y, pre, mid
    check(_lib.load().dpot_afno_mlp2(X.data_ptr(), WaT.data_ptr(), _p(ba), WbT.data_ptr(), _p(bb), _p(aux), _p(pre),
                                     _p(mid), Y.data_ptr(), M, nb, bs, ld, ld, act, mode, layout, _stream()),
          "afno_mlp2")
    return Y, pre, mid


def afno_unpack_grad(dwbig: Tensor, dbbig: Tensor, nb: int, bs: int, out_dw: Optional[Tensor] = None,
                     out_db: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    dw = _out(out_dw, (2, nb, bs, bs), dwbig.device)
    db = _out(out_db, (2, nb, bs), dwbig.device)
    check(_lib.load().dpot_afno_unpack_grad(dwbig.data_ptr(), dbbig.data_ptr(), dw.data_ptr(), db.data_ptr(), nb, bs,
                                            _stream()), "afno_unpack_grad")
    return dw, db


# ------------------------------------------------------------------------------------------------------
# GroupNorm
# ------------------------------------------------------------------------------------------------------
def _gn_workspace(B: int, T: int, E: int, G: int, device) -> Optional[Tensor]:
    """scratch for the chunked GroupNorm kernels (few, large (sample, group) slabs: DPOT-L at 256^2), or None"""
    n = _lib.load().dpot_groupnorm_ws_elems(B, T, E, G)
    return torch.empty(n, dtype=torch.float32, device=device) if n > 0 else None


def groupnorm_ws_elems(B: int, T: int, E: int, G: int) -> int:
    return int(_lib.load().dpot_groupnorm_ws_elems(B, T, E, G))


def groupnorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, G: int = 8, eps: float = 1e-5, chunked: bool = True):
    """chunked=False: never the chunked kernels (no workspace is passed)"""
    B, T, E = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(B, G, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B, G, dtype=torch.float32, device=x.device)
    ws = _gn_workspace(B, T, E, G, x.device) if chunked else None
    check(_lib.load().dpot_groupnorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                         mean.data_ptr(), rstd.data_ptr(), _p(ws), B, T, E, G, eps, _stream()),
          "groupnorm_fwd")
    return y, mean, rstd


def groupnorm_stats_supported(B: int, T: int, E: int, G: int = 8) -> bool:
    """statistics-only GroupNorm (groupnorm_stats): the chunked kernels (few, large (sample, group) slabs - DPOT-L)"""
    return _lib.load().dpot_groupnorm_ws_elems(B, T, E, G) > 0


def groupnorm_stats(x: Tensor, gamma: Tensor, beta: Tensor, G: int = 8, eps: float = 1e-5):
    """(mean, rstd) [B, G] of GroupNorm(G, E) over x [B, T, E] without writing the normalised tensor - the consumer applies
    the affine map on its load (rfft2(norm=...), bf16_pack_both(norm=...)); needs groupnorm_stats_supported"""
    B, T, E = x.shape
    mean = torch.empty(B, G, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B, G, dtype=torch.float32, device=x.device)
    ws = _gn_workspace(B, T, E, G, x.device)
    check(_lib.load().dpot_groupnorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, mean.data_ptr(),
                                         rstd.data_ptr(), _p(ws), B, T, E, G, eps, _stream()), "groupnorm_fwd (statistics)")
    return mean, rstd


def groupnorm_bwd(dy: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor, G: int = 8,
                  add: Optional[Tensor] = None, out_dgamma: Optional[Tensor] = None,
                  out_dbeta: Optional[Tensor] = None, defer: bool = False):
    """defer=False: returns (dx, dgamma, dbeta).  defer=True: returns (dx, part) - the per-sample partials [2,B,E] of the
    parameter gradients, to be reduced later by groupnorm_param_grads (several layers in one launch)"""
    B, T, E = x.shape
    dx = torch.empty_like(x)
    part = torch.empty(2, B, E, dtype=torch.float32, device=x.device)
    dgamma = None if defer else _out(out_dgamma, (E,), x.device)
    dbeta = None if defer else _out(out_dbeta, (E,), x.device)
    ws = _gn_workspace(B, T, E, G, x.device)
    check(_lib.load().dpot_groupnorm_bwd(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                         gamma.data_ptr(), _p(add), dx.data_ptr(), _p(dgamma), _p(dbeta),
                                         part.data_ptr(), _p(ws), B, T, E, G, _stream()), "groupnorm_bwd")
    return (dx, part) if defer else (dx, dgamma, dbeta)


def groupnorm_bwd_packs_rows(B: int, T: int, E: int, G: int = 8) -> int:
    """column-sum partial rows per sample groupnorm_bwd_packs writes for this shape (0: not supported)"""
    return int(_lib.load().dpot_groupnorm_bwd_packs_rows(B, T, E, G))


def groupnorm_bwd_packs_supported(T: int, E: int, G: int = 8, B: int = 1) -> bool:
    return groupnorm_bwd_packs_rows(B, T, E, G) > 0


def groupnorm_bwd_packs(dy: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor, G: int = 8,
                        add: Optional[Tensor] = None):
    """groupnorm_bwd(defer=True) that also writes dx as the bf16 operand packs of the previous block's channel-MLP backward:
    returns (dx, part [2,B,E], dx row-form pack, dx transposed pack, column sums of dx [B * rows, E] over `rows` token ranges
    per sample) - the last three are what bf16_pack_both(dx, want_colsum=True) would produce from a second pass over dx"""
    B, T, E = x.shape
    lib = _lib.load()
    rows = lib.dpot_groupnorm_bwd_packs_rows(B, T, E, G)
    dx = torch.empty_like(x)
    part = torch.empty(2, B, E, dtype=torch.float32, device=x.device)
    pr = torch.empty(lib.dpot_bf16_packed_elems(B * T, E, 1), dtype=torch.bfloat16, device=x.device)
    pt = torch.empty(lib.dpot_bf16_packed_elems(E, B * T, 1), dtype=torch.bfloat16, device=x.device)
    cs = torch.empty(B * max(rows, 1), E, dtype=torch.float32, device=x.device)
    ws = _gn_workspace(B, T, E, G, x.device)
    check(lib.dpot_groupnorm_bwd_packs(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                       _p(add), dx.data_ptr(), part.data_ptr(), pr.data_ptr(), pt.data_ptr(), cs.data_ptr(),
                                       _p(ws), B, T, E, G, _stream()), "groupnorm_bwd_packs")
    return dx, part, pr, pt, cs


def groupnorm_param_grads(jobs):
    """jobs = [(part [2,B,E], out_dgamma | None, out_dbeta | None)] (<= 4) -> [(dgamma, dbeta)], one launch"""
    n = len(jobs)
    _, B, E = jobs[0][0].shape
    outs = [(_out(og, (E,), p.device), _out(ob, (E,), p.device)) for p, og, ob in jobs]
    parts = (C.c_void_p * n)(*[p.data_ptr() for p, _, _ in jobs])
    dgs = (C.c_void_p * n)(*[g.data_ptr() for g, _ in outs])
    dbs = (C.c_void_p * n)(*[b.data_ptr() for _, b in outs])
    check(_lib.load().dpot_groupnorm_param_grads(parts, dgs, dbs, n, B, E, _stream()), "groupnorm_param_grads")
    return outs


# ------------------------------------------------------------------------------------------------------
# GroupNorm fused with the mixer's DFTs (csrc/gn_dft.hip)
# ------------------------------------------------------------------------------------------------------
def gn_dft_supported(h: int, w: int, E: int, G: int = 8) -> bool:
    """16x16 latent grid and 64 / 128 channels per group (DPOT-Tiny / -Small / -Medium at 128^2); DPOT_GN_DFT=0 disables"""
    return bool(_lib.load().dpot_gn_dft_supported(h, w, E, G))


def gn_rfft2(x: Tensor, gamma: Tensor, beta: Tensor, h: int, w: int, nb: int, mx: int, my: int, G: int = 8,
             eps: float = 1e-5):
    """(spec [B*mx*my, 2E] = rfft2(GroupNorm(x)), mean [B,G], rstd [B,G]) in one launch; GroupNorm(x) is not written"""
    B, _, E = x.shape
    spec = torch.empty(B * mx * my, 2 * E, dtype=torch.float32, device=x.device)
    mean = torch.empty(B, G, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B, G, dtype=torch.float32, device=x.device)
    check(_lib.load().dpot_gn_rfft2(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), spec.data_ptr(), mean.data_ptr(),
                                    rstd.data_ptr(), B, h, w, E, G, nb, mx, my, eps, _stream()), "gn_rfft2")
    return spec, mean, rstd


def irfft2_gn(spec: Tensor, x: Tensor, mean1: Tensor, rstd1: Tensor, g1: Tensor, b1: Tensor, g2: Tensor, b2: Tensor,
              h: int, w: int, nb: int, mx: int, my: int, G: int = 8, eps: float = 1e-5, col_weights: int = 1,
              want_xn2: bool = True):
    """(y1 = irfft2(spec) + GroupNorm1(x), xn2 = GroupNorm2(y1), mean2, rstd2) in one launch; want_xn2=False: statistics
    only (xn2 = None: the bf16 channel MLP packs GroupNorm2(y1) itself, bf16_pack_both(norm=...))"""
    B, _, E = x.shape
    y1, xn2 = torch.empty_like(x), (torch.empty_like(x) if want_xn2 else None)
    mean2 = torch.empty(B, G, dtype=torch.float32, device=x.device)
    rstd2 = torch.empty(B, G, dtype=torch.float32, device=x.device)
    check(_lib.load().dpot_irfft2_gn(spec.data_ptr(), x.data_ptr(), mean1.data_ptr(), rstd1.data_ptr(), g1.data_ptr(),
                                     b1.data_ptr(), g2.data_ptr(), b2.data_ptr(), y1.data_ptr(), _p(xn2),
                                     mean2.data_ptr(), rstd2.data_ptr(), B, h, w, E, G, nb, mx, my, col_weights, eps,
                                     _stream()), "irfft2_gn")
    return y1, xn2, mean2, rstd2


def afno_layer_mode() -> int:
    """DPOT_TUNE afno_layer: 0 = never the one-launch AFNO layer (csrc/afno_fused.hip), 1 = wherever it is supported, -1
    (default) = where it measured faster than the three-launch form (afno_layer_wins)"""
    return tune("afno_layer")


def afno_layer_wins(B: int, nb: int) -> bool:
    """one (sample, channel block) workgroup per CU and round: the one-launch layer costs ~100 us per round of 256
    workgroups whatever their number, the three launches ~125 us per 256 (profiles/r05_f4_fused_vs_3launch.txt: 128
    workgroups 84 vs 72 us - DPOT-Tiny at batch 32 stays on three launches -, 256: 98 vs 124, 512: 196 vs 240) - so it
    is selected where the last round is at least 80 % full"""
    n = B * nb
    rounds = (n + 255) // 256
    return n >= 0.8 * 256 * rounds


def afno_fused_supported(h: int, w: int, E: int, nb: int, mx: int, my: int, G: int = 8, B: Optional[int] = None,
                         layout: int = 1) -> bool:
    """one-launch AFNO layer forward: 16x16 latent grid, 128 channels per block, all modes kept, three-product packs"""
    mode = afno_layer_mode()
    if mode == 0 or layout != 1 or not _lib.load().dpot_afno_fused_supported(h, w, E, G, nb, mx, my):
        return False
    return mode == 1 or (B is not None and afno_layer_wins(B, nb))


def afno_fused_fwd(x: Tensor, g1: Optional[Tensor], b1: Optional[Tensor], WaT: Tensor, ba: Tensor, WbT: Tensor, bb: Tensor,
                   g2: Optional[Tensor], b2: Optional[Tensor], h: int, w: int, nb: int, mx: int, my: int, act: int,
                   G: int = 8, eps: float = 1e-5, save: bool = True, want_y1: bool = True, want_xn2: bool = True,
                   want_packs: bool = False, packs_trans: bool = True):
    """[GroupNorm1] -> rfft2 -> 2-layer complex MLP -> irfft2 + x_orig -> [GroupNorm2] in ONE launch (csrc/afno_fused.hip).
    Returns (S, O1pre, y1, xn2, mean1, rstd1, mean2, rstd2) - the tensors gn_rfft2 / afno_mlp2 / irfft2_gn return, same
    layouts; save=False (inference): S = O1pre = None; entries that do not apply are None.
    want_packs (with g2): two more entries (xp, xpT) = GroupNorm2(y1) as the bf16 operand packs of the channel MLP (what
    bf16_pack_both(y1, norm=...) returns), written by the same launch; packs_trans=False: xpT = None (inference: only the
    weight gradient reads the transposed form)"""
    B, tok, E = x.shape
    dev = x.device
    Mm = B * mx * my
    S = torch.empty(Mm, 2 * E, dtype=torch.float32, device=dev) if save else None
    pre = torch.empty(Mm, 2 * E, dtype=torch.float32, device=dev) if save else None
    want_xn2 = want_xn2 and g2 is not None
    want_packs = want_packs and g2 is not None
    y1 = torch.empty_like(x) if (want_y1 or not (want_xn2 or want_packs)) else None
    xn2 = torch.empty_like(x) if want_xn2 else None
    st = [torch.empty(B, G, dtype=torch.float32, device=dev) if g is not None else None for g in (g1, g1, g2, g2)]
    lib = _lib.load()
    xp = xpT = None
    if want_packs:
        n = lib.dpot_bf16_packed_elems(B * tok, E, 1)
        xp = torch.empty(n, dtype=torch.bfloat16, device=dev)
        xpT = torch.empty(lib.dpot_bf16_packed_elems(E, B * tok, 1), dtype=torch.bfloat16, device=dev) if packs_trans else None
    check(lib.dpot_afno_fused_fwd(x.data_ptr(), _p(g1), _p(b1), WaT.data_ptr(), _p(ba), WbT.data_ptr(), _p(bb),
                                  _p(g2), _p(b2), _p(S), _p(pre), _p(y1), _p(xn2), _p(st[0]), _p(st[1]),
                                  _p(st[2]), _p(st[3]), _p(xp), _p(xpT), B, h, w, E, G, nb, mx, my, act, eps, _stream()),
          "afno_fused_fwd")
    if want_packs:
        return S, pre, y1, xn2, st[0], st[1], st[2], st[3], xp, xpT
    return S, pre, y1, xn2, st[0], st[1], st[2], st[3]


def gn_bwd_rfft2(dy: Tensor, xin: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor, h: int, w: int, nb: int, mx: int,
                 my: int, G: int = 8, col_weights: int = 1):
    """(dx = GroupNorm backward of dy, part [2,B,E], spec = rfft2(dx; col_weights)) in one launch"""
    B, _, E = xin.shape
    dx = torch.empty_like(xin)
    part = torch.empty(2, B, E, dtype=torch.float32, device=xin.device)
    spec = torch.empty(B * mx * my, 2 * E, dtype=torch.float32, device=xin.device)
    check(_lib.load().dpot_gn_bwd_rfft2(dy.data_ptr(), xin.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                        dx.data_ptr(), part.data_ptr(), spec.data_ptr(), B, h, w, E, G, nb, mx, my,
                                        col_weights, _stream()), "gn_bwd_rfft2")
    return dx, part, spec


def irfft2_gn_bwd(spec: Tensor, res: Tensor, xin: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor, h: int, w: int,
                  nb: int, mx: int, my: int, add: Optional[Tensor] = None, G: int = 8, col_weights: int = 0):
    """(dx = GroupNorm backward of (irfft2(spec; col_weights) + res) (+ add), part [2,B,E]) in one launch"""
    B, _, E = xin.shape
    dx = torch.empty_like(xin)
    part = torch.empty(2, B, E, dtype=torch.float32, device=xin.device)
    check(_lib.load().dpot_irfft2_gn_bwd(spec.data_ptr(), res.data_ptr(), xin.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                         gamma.data_ptr(), _p(add), dx.data_ptr(), part.data_ptr(), B, h, w, E, G, nb, mx,
                                         my, col_weights, _stream()), "irfft2_gn_bwd")
    return dx, part


# ------------------------------------------------------------------------------------------------------
# data movement / small ops
# ------------------------------------------------------------------------------------------------------
def patchify(x: Tensor, gx: Tensor, gy: Tensor, gt: Tensor, P: int) -> Tensor:
    B, X, Y, T, Cc = x.shape
    K0 = (Cc + 3) * P * P
    A = torch.empty(B * (X // P) * (Y // P) * T, K0, dtype=torch.float32, device=x.device)
    check(_lib.load().dpot_patchify(x.data_ptr(), gx.data_ptr(), gy.data_ptr(), gt.data_ptr(), A.data_ptr(), B, X, Y,
                                    T, Cc, P, _stream()), "patchify")
    return A


def embed_supported(Cc: int, P: int, T: int, hid: int, w: int) -> bool:
    """shapes the implicit-GEMM patch embedding covers (csrc/embed.hip); DPOT_EMBED_IMPLICIT=0 disables it"""
    return (tune("embed_implicit") != 0
            and bool(_lib.load().dpot_embed_supported(Cc, P, T, hid, w)))


def embed_pack_w0(w0: Tensor) -> Tensor:
    lib = _lib.load()
    wf = torch.empty(lib.dpot_embed_wfrag_elems(), dtype=torch.float32, device=w0.device)
    check(lib.dpot_embed_pack_w0(w0.data_ptr(), w0.shape[0], wf.data_ptr(), _stream()), "embed_pack_w0")
    return wf


def embed_fwd(x: Tensor, wfrag: Tensor, btab: Tensor, hidp: int, act: int) -> Tuple[Tensor, Tensor]:
    """(Hh, Hpre) [B*tok*T, hidp] of the patch convolution, gathered from x[B,X,Y,T,4] (no patch matrix)"""
    B, X, Y, T, _ = x.shape
    M0 = B * (X // 8) * (Y // 8) * T
    hpre = torch.empty(M0, hidp, dtype=torch.float32, device=x.device)
    hh = torch.empty_like(hpre)
    check(_lib.load().dpot_embed_fwd(x.data_ptr(), wfrag.data_ptr(), btab.data_ptr(), hpre.data_ptr(), hh.data_ptr(),
                                     B, X, Y, T, hidp, act, _stream()), "embed_fwd")
    return hh, hpre


def embed_wgrad(x: Tensor, dhpre: Tensor, dw0: Tensor, hid: int) -> Tensor:
    """data-channel columns of the patch-conv weight gradient: dw0[hid, K0][:, :4*64] (deterministic two-launch sum)"""
    lib = _lib.load()
    B, X, Y, T, _ = x.shape
    ws = torch.empty(lib.dpot_embed_wgrad_ws_elems(B, X, Y), dtype=torch.float32, device=x.device)
    check(lib.dpot_embed_wgrad(x.data_ptr(), dhpre.data_ptr(), ws.data_ptr(), dw0.data_ptr(), dw0.stride(0), hid, B, X,
                               Y, T, dhpre.shape[1], _stream()), "embed_wgrad")
    return dw0


def unpatchify(dA: Tensor, B: int, X: int, Y: int, T: int, Cc: int, P: int) -> Tensor:
    dx = torch.empty(B, X, Y, T, Cc, dtype=torch.float32, device=dA.device)
    check(_lib.load().dpot_unpatchify(dA.data_ptr(), dx.data_ptr(), B, X, Y, T, Cc, P, _stream()), "unpatchify")
    return dx


def pixel_shuffle(z: Tensor, B: int, h: int, w: int, P: int, Cc: int, inverse: bool = False) -> Tensor:
    """inverse=False: z[(b,px,py,i,j),Cc] -> out[b,px*P+i,py*P+j,Cc];  inverse=True: the other way."""
    out = torch.empty(B * h * w * P * P, Cc, dtype=torch.float32, device=z.device)
    if not inverse:
        out = out.view(B, h * P, w * P, Cc)
        check(_lib.load().dpot_pixel_shuffle(z.data_ptr(), out.data_ptr(), B, h, w, P, Cc, 0, _stream()),
              "pixel_shuffle")
    else:
        check(_lib.load().dpot_pixel_shuffle(z.data_ptr(), out.data_ptr(), B, h, w, P, Cc, 1, _stream()),
              "pixel_shuffle")
    return out


def copy2d_pad(src: Tensor, sR: int, sC: int, dR: int, dC: int, out: Optional[Tensor] = None) -> Tensor:
    dst = _out(out, (dR, dC), src.device)
    check(_lib.load().dpot_copy2d_pad(src.data_ptr(), sR, sC, dst.data_ptr(), dR, dC, _stream()), "copy2d_pad")
    return dst


def transpose2d(src: Tensor, nbatch: int, R: int, Cn: int, out: Optional[Tensor] = None) -> Tensor:
    dst = _out(out, (nbatch, Cn, R), src.device)
    check(_lib.load().dpot_transpose2d(src.data_ptr(), dst.data_ptr(), nbatch, R, Cn, _stream()), "transpose2d")
    return dst


def colsum(X: Tensor, M: int, N: int, ld: Optional[int] = None, out: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    out = _out(out, (N,), X.device)
    part = torch.empty(lib.dpot_colsum_parts(M), N, dtype=torch.float32, device=X.device)
    check(lib.dpot_colsum(X.data_ptr(), M, N, ld or N, out.data_ptr(), part.data_ptr(), _stream()), "colsum")
    return out


def group_rowsum(X: Tensor, B: int, R: int, T: int, N: int) -> Tensor:
    out = torch.empty(R, N, dtype=torch.float32, device=X.device)
    check(_lib.load().dpot_group_rowsum(X.data_ptr(), out.data_ptr(), B, R, T, N, _stream()), "group_rowsum")
    return out


def token_mean(x: Tensor) -> Tensor:
    B, T, E = x.shape
    y = torch.empty(B, E, dtype=torch.float32, device=x.device)
    check(_lib.load().dpot_token_mean(x.data_ptr(), y.data_ptr(), B, T, E, _stream()), "token_mean")
    return y


def token_mean_bwd(dy: Tensor, T: int, add: Optional[Tensor] = None) -> Tensor:
    B, E = dy.shape
    dx = torch.empty(B, T, E, dtype=torch.float32, device=dy.device)
    check(_lib.load().dpot_token_mean_bwd(dy.data_ptr(), _p(add), dx.data_ptr(), B, T, E, _stream()),
          "token_mean_bwd")
    return dx


def bias_add(x: Tensor, v: Tensor) -> Tensor:
    """y[r, n] = x[r, n] + v[n]"""
    R, N = x.shape
    y = torch.empty_like(x)
    check(_lib.load().dpot_bias_add(x.data_ptr(), v.data_ptr(), y.data_ptr(), R, N, _stream()), "bias_add")
    return y


def tile_vec(v: Tensor, reps: int) -> Tensor:
    """v [N] repeated `reps` times -> [reps * N]"""
    N = v.numel()
    y = torch.empty(reps * N, dtype=torch.float32, device=v.device)
    check(_lib.load().dpot_bias_add(None, v.data_ptr(), y.data_ptr(), reps, N, _stream()), "tile_vec")
    return y


def scale_shift(x: Tensor, scale: Tensor, shift: Tensor) -> Tensor:
    B, T, E = x.shape
    y = torch.empty_like(x)
    check(_lib.load().dpot_scale_shift(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), B, T, E,
                                       _stream()), "scale_shift")
    return y


def scale_shift_bwd(dy: Tensor, x: Tensor, scale: Tensor):
    B, T, E = x.shape
    dx = torch.empty_like(x)
    dscale = torch.empty(B, E, dtype=torch.float32, device=x.device)
    dshift = torch.empty_like(dscale)
    check(_lib.load().dpot_scale_shift_bwd(dy.data_ptr(), x.data_ptr(), scale.data_ptr(), dx.data_ptr(),
                                           dscale.data_ptr(), dshift.data_ptr(), B, T, E, _stream()), "scale_shift_bwd")
    return dx, dscale, dshift


def timeagg_scale_w(w: Tensor, gamma: Tensor, tt: Tensor) -> Tensor:
    T, E, _ = w.shape
    ws = torch.empty_like(w)
    check(_lib.load().dpot_timeagg_scale_w(w.data_ptr(), gamma.data_ptr(), tt.data_ptr(), ws.data_ptr(), T, E,
                                           _stream()), "timeagg_scale_w")
    return ws


def timeagg_scale_w_bwd(dws: Tensor, w: Tensor, gamma: Tensor, tt: Tensor, out_dw: Optional[Tensor] = None,
                        out_dgamma: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    T, E, _ = w.shape
    dw = _out(out_dw, (T, E, E), w.device)
    dgamma = _out(out_dgamma, (1, E), w.device)
    check(_lib.load().dpot_timeagg_scale_w_bwd(dws.data_ptr(), w.data_ptr(), gamma.data_ptr(), tt.data_ptr(),
                                               dw.data_ptr(), dgamma.data_ptr(), T, E, _stream()),
          "timeagg_scale_w_bwd")
    return dw, dgamma


def out_tail_supported(old: int, co: int, n_pixels: int) -> bool:
    return old == 32 and 0 < co <= 32 and n_pixels % 32 == 0


def out_tail_pad(w4: Tensor, b4: Tensor, co: int) -> Tuple[Tensor, Tensor]:
    """W4 [co,32] -> [32,32], b4 [co] -> [32], zero padded (the tail kernels load them unconditionally)"""
    return copy2d_pad(w4, co, 32, 32, 32), copy2d_pad(b4, 1, co, 1, 32).view(32)


def out_tail_fwd(upre: Tensor, w2: Tensor, b2: Tensor, w4p: Tensor, b4p: Tensor, B: int, h: int, w: int, P: int,
                 co: int, act: int) -> Tensor:
    out = torch.empty(B, h * P, w * P, co, dtype=torch.float32, device=upre.device)
    check(_lib.load().dpot_out_tail_fwd(upre.data_ptr(), w2.data_ptr(), b2.data_ptr(), w4p.data_ptr(),
                                        b4p.data_ptr(), out.data_ptr(), B, h, w, P, co, act, _stream()), "out_tail_fwd")
    return out


def colsum_scatter(X: Tensor, M: int, N: int, segments) -> None:
    """column sums of X[M,N]; segments = [(first column, tensor)]: each tensor receives its columns (one launch pair)"""
    lib = _lib.load()
    n = len(segments)
    starts = (C.c_int * n)(*[s for s, _ in segments])
    lens = (C.c_int * n)(*[t.numel() for _, t in segments])
    dsts = (C.c_void_p * n)(*[t.data_ptr() for _, t in segments])
    part = torch.empty(lib.dpot_colsum_parts(M) * N, dtype=torch.float32, device=X.device)
    check(lib.dpot_colsum_scatter(X.data_ptr(), M, N, N, part.data_ptr(), n, starts, lens, dsts, _stream()),
          "colsum_scatter")


def out_tail_bwd(upre: Tensor, dout: Tensor, w2: Tensor, b2: Tensor, w4p: Tensor, B: int, h: int, w: int, P: int,
                 co: int, act: int, outs=None):
    """returns (dupre [pixels,32], (dW2 [32,32], dW4 [co,32], db2 [32], colsum(dupre) [32], db4 [co])); `outs`: the five
    destination tensors (gradient slots) or None entries for fresh ones - all written by one reduction"""
    lib = _lib.load()
    rows, cols = lib.dpot_out_tail_partial_rows(B, h, w, P), lib.dpot_out_tail_partial_cols()
    dupre = torch.empty_like(upre)
    part = torch.empty(rows, cols, dtype=torch.float32, device=upre.device)
    check(lib.dpot_out_tail_bwd(upre.data_ptr(), dout.data_ptr(), w2.data_ptr(), b2.data_ptr(), w4p.data_ptr(),
                                dupre.data_ptr(), part.data_ptr(), B, h, w, P, co, act, _stream()), "out_tail_bwd")
    shapes = ((32, 32), (co, 32), (32,), (32,), (co,))
    starts = (0, 1024, 2048, 2080, 2112)
    outs = list(outs) if outs is not None else [None] * 5
    res = [_out(o, shp, upre.device) for o, shp in zip(outs, shapes)]
    colsum_scatter(part, rows, cols, list(zip(starts, res)))
    return dupre, tuple(res)


# ------------------------------------------------------------------------------------------------------
# loss / optimiser
# ------------------------------------------------------------------------------------------------------
def rel_l2_fwd(x: Tensor, y: Tensor, mask: Optional[Tensor], B: int, S: int, Cc: int, Tt: int):
    nch = _lib.load().dpot_rel_l2_chunks(S, Cc)
    stats = torch.empty(1 + nch, B, Cc, 4, dtype=torch.float32, device=x.device)
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    check(_lib.load().dpot_rel_l2_fwd(x.data_ptr(), y.data_ptr(), _p(mask), stats.data_ptr(), loss.data_ptr(), B, S,
                                      Cc, Tt, _stream()), "rel_l2_fwd")
    return loss, stats


def rel_l2_bwd(x: Tensor, y: Tensor, mask: Optional[Tensor], stats: Tensor, gloss: Tensor, B: int, S: int, Cc: int,
               Tt: int) -> Tensor:
    dx = torch.empty_like(x)
    check(_lib.load().dpot_rel_l2_bwd(x.data_ptr(), y.data_ptr(), _p(mask), stats.data_ptr(), gloss.data_ptr(),
                                      dx.data_ptr(), B, S, Cc, Tt, _stream()), "rel_l2_bwd")
    return dx


def sumsq(g: Tensor, out: Tensor, part: Tensor, accumulate: bool = False) -> Tensor:
    check(_lib.load().dpot_sumsq(g.data_ptr(), g.numel(), out.data_ptr(), part.data_ptr(), int(accumulate),
                                 _stream()), "sumsq")
    return out


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, hyper: Tensor, sumsq_: Optional[Tensor],
              grad_scale: float = 1.0) -> None:
    check(_lib.load().dpot_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                     hyper.data_ptr(), _p(sumsq_), grad_scale, _stream()), "adam_step")


class AdamPackPlan:
    """Device tables of dpot_adam_step_packs for one (flat parameter buffer, PanelPacks of 1-plane bf16 weight packs): every
    weight W [R, K] (row-major inside the flat buffer) that the pack set holds as the pair (W, R, K, K, False) / (W, K, R, K,
    True) becomes a job whose 64 x 256 tiles run Adam AND write both packs; the rest of [0, n_active) becomes plain ranges of
    <= 2^20 elements.  None (`AdamPackPlan.build` returns None) when any weight of the set does not qualify."""
    MAX_RANGE = 1 << 20

    @staticmethod
    def build(flat: Tensor, n_active: int, pp: "PanelPacks"):
        import numpy as np
        lib = _lib.load()
        if not pp.bf16 or pp.planes != 1 or pp.n % 2:
            return None
        base, jobs = flat.data_ptr(), []
        for j in range(0, pp.n, 2):
            (w, rows, K, ld, tr), (w2, rows2, K2, ld2, tr2) = pp.jobs[j], pp.jobs[j + 1]
            off = (w.data_ptr() - base) // 4
            if (w2.data_ptr() != w.data_ptr() or tr or not tr2 or ld != K or ld2 != K or rows2 != K or K2 != rows
                    or (w.data_ptr() - base) % 16 or off < 0 or off + rows * K > n_active or not w.is_contiguous()
                    or w.numel() != rows * K or not lib.dpot_adam_pack_supported(rows, K)):
                return None
            jobs.append((off, rows, K, pp.bufs[j], pp.bufs[j + 1]))
        jobs.sort(key=lambda t: t[0])
        self = AdamPackPlan()
        self.pp = pp
        dev = flat.device
        ntiles, tile_job = 0, []
        host = np.zeros(len(jobs) * C.sizeof(_lib.AdamPackJob), dtype=np.uint8)
        tab = (_lib.AdamPackJob * len(jobs)).from_buffer(host)
        ranges, pos = [], 0
        for i, (off, R, K, drow, dtr) in enumerate(jobs):
            if off < pos:
                return None                                   # overlapping weights: not a layout this plan understands
            t = tab[i]
            t.off, t.dst_rows, t.dst_trans, t.R, t.K, t.tile0 = off, drow.data_ptr(), dtr.data_ptr(), R, K, ntiles
            nt = (R // 64) * (K // 256)
            tile_job += [i] * nt
            ntiles += nt
            ranges.append((pos, off - pos))
            pos = off + R * K
        ranges.append((pos, n_active - pos))
        split = []
        for st, ln in ranges:
            while ln > 0:
                c = min(ln, AdamPackPlan.MAX_RANGE)
                split.append((st, c))
                st, ln = st + c, ln - c
        self.ntiles, self.nranges = ntiles, len(split)
        self.max_range = max((c for _, c in split), default=0)
        self.jobs_dev = torch.from_numpy(host).to(dev)
        self.tile_job_dev = torch.tensor(tile_job, dtype=torch.int32, device=dev)
        rh = np.array(split, dtype=np.int64).reshape(-1, 2) if split else np.zeros((0, 2), dtype=np.int64)
        self.ranges_dev = torch.from_numpy(rh).to(dev)
        self.key = (base, n_active, id(pp))
        return self


def adam_step_packs(plan: AdamPackPlan, p: Tensor, g: Tensor, m: Tensor, v: Tensor, hyper: Tensor, sumsq_: Optional[Tensor],
                    grad_scale: float = 1.0) -> None:
    """dpot_adam_step over the flat buffers whose channel-MLP weights ALSO leave as their two bf16 packs (plan.pp's buffers)"""
    check(_lib.load().dpot_adam_step_packs(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), hyper.data_ptr(), _p(sumsq_),
                                           grad_scale, plan.jobs_dev.data_ptr(), plan.tile_job_dev.data_ptr(), plan.ntiles,
                                           plan.ranges_dev.data_ptr() if plan.nranges else None, plan.nranges, plan.max_range,
                                           _stream()), "adam_step_packs")


def adam_stage(hyper: Tensor, step: Tensor, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float,
               max_norm: float, advance: int = 1) -> None:
    """step[0] += advance; hyper <- {lr, betas, eps, wd, bias corrections(step), max_norm}; every value travels by
    value in the launch packet (no host buffer that a later step could overwrite)"""
    check(_lib.load().dpot_adam_stage(hyper.data_ptr(), step.data_ptr(), lr, beta1, beta2, eps, weight_decay,
                                      max_norm, advance, _stream()), "adam_stage")


_rng_states = {}


def rng_state(device) -> Tensor:
    """{seed, offset} of the in-kernel noise generator of `device` (int64[2]); seeded from torch.initial_seed()"""
    key = torch.device(device).index or 0
    if key not in _rng_states:
        _rng_states[key] = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64,
                                        device=device)
    return _rng_states[key]


def noise_inject(xx: Tensor, eps: Optional[Tensor], noise_scale: float, return_norms: bool = False):
    """xx + noise_scale * ||xx||_(X,Y,T) * eps;  eps=None: eps ~ N(0,1) is drawn inside the kernel.
    return_norms: also return the scratch tensor whose first B*C floats are the per-(b,c) norms (for the backward)"""
    B, Cc = xx.shape[0], xx.shape[-1]
    S = xx.numel() // (B * Cc)
    out = torch.empty_like(xx)
    lib = _lib.load()
    norms = torch.empty(B * Cc * (1 + lib.dpot_noise_chunks(S, Cc)), dtype=torch.float32, device=xx.device)
    if eps is None:
        check(lib.dpot_noise_inject_rng(xx.data_ptr(), out.data_ptr(), norms.data_ptr(), rng_state(xx.device).data_ptr(),
                                        noise_scale, B, S, Cc, _stream()), "noise_inject_rng")
    else:
        check(lib.dpot_noise_inject(xx.data_ptr(), eps.data_ptr(), out.data_ptr(), norms.data_ptr(), noise_scale,
                                    B, S, Cc, _stream()), "noise_inject")
    return (out, norms) if return_norms else out


def noise_inject_bwd(xx: Tensor, eps: Optional[Tensor], rng_snapshot: Optional[Tensor], g: Tensor, norms: Tensor,
                     noise_scale: float) -> Tensor:
    """d/dxx of noise_inject: g + noise_scale * xx / ||xx|| * sum(g * eps); eps given, or re-drawn from the
    generator state the forward used (rng_snapshot = clone of rng_state() taken right after the forward call)"""
    B, Cc = xx.shape[0], xx.shape[-1]
    S = xx.numel() // (B * Cc)
    lib = _lib.load()
    dx = torch.empty_like(xx)
    part = torch.empty(B * Cc * lib.dpot_noise_chunks(S, Cc), dtype=torch.float32, device=xx.device)
    check(lib.dpot_noise_inject_bwd(xx.data_ptr(), _p(eps), _p(rng_snapshot), g.data_ptr(), norms.data_ptr(),
                                    dx.data_ptr(), part.data_ptr(), noise_scale, B, S, Cc, _stream()),
          "noise_inject_bwd")
    return dx


def window_slide(xx: Tensor, im: Tensor) -> Tensor:
    """cat(xx[..., Tb:, :], im) along the time axis of [B,X,Y,T,C] (train_temporal.py:219)"""
    T, Cc = xx.shape[-2], xx.shape[-1]
    Tb = im.shape[-2]
    # the reference's torch.cat raises when the prediction does not continue the window (out_channels != in_channels,
    # other leading dims); the kernel would read `im` with the wrong stride instead
    if im.shape[:-2] != xx.shape[:-2] or im.shape[-1] != Cc or Tb > T:
        raise _lib.DpotHipError(f"window_slide: prediction {tuple(im.shape)} does not continue the window {tuple(xx.shape)} "
                                f"(needs equal leading dims and channel count, T_bundle <= T_in)")
    rows = xx.numel() // (T * Cc)
    out = torch.empty_like(xx)
    check(_lib.load().dpot_window_slide(xx.data_ptr(), im.data_ptr(), out.data_ptr(), rows, T, Tb, Cc, _stream()),
          "window_slide")
    return out


def window_slide_bwd(dout: Tensor, Tb: int, need_xx: bool, need_im: bool):
    T, Cc = dout.shape[-2], dout.shape[-1]
    if not 0 < Tb <= T:
        raise _lib.DpotHipError(f"window_slide_bwd: T_bundle {Tb} outside (0, {T}]")
    rows = dout.numel() // (T * Cc)
    dxx = torch.empty_like(dout) if need_xx else None
    dim = torch.empty(dout.shape[:-2] + (Tb, Cc), dtype=torch.float32, device=dout.device) if need_im else None
    check(_lib.load().dpot_window_slide_bwd(dout.data_ptr(), _p(dxx), _p(dim), rows, T, Tb, Cc, _stream()),
          "window_slide_bwd")
    return dxx, dim
