"""ctypes binding of libdpot_hip.so (include/dpot_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPOT_HIP_LIB") or os.path.join(HERE, "lib", "libdpot_hip.so")   # override: kernel experiments

c_fp = C.c_void_p       # device pointer (float*)
c_i = C.c_int
c_i64 = C.c_int64
c_f = C.c_float
c_d = C.c_double


class GemmDesc(C.Structure):
    """mirror of struct dpot_gemm_desc"""
    _fields_ = [
        ("A", c_fp), ("B", c_fp), ("C", c_fp),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("batch", C.c_int32),
        ("transA", C.c_int32), ("transB", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
        ("strideA", c_i64), ("strideB", c_i64), ("strideC", c_i64),
        ("bias", c_fp), ("strideBias", c_i64),
        ("act", C.c_int32), ("epi_mode", C.c_int32),
        ("aux", c_fp), ("ldaux", C.c_int32), ("strideAux", c_i64),
        ("preact", c_fp), ("ldpre", C.c_int32), ("stridePre", c_i64),
        ("res", c_fp), ("ldres", C.c_int32), ("res_div", C.c_int32), ("res_mod", C.c_int32),
        ("strideRes", c_i64),
        ("accumulate", C.c_int32), ("splitk", C.c_int32),
        ("workspace", c_fp),
        ("tile", C.c_int32),
        ("tag", C.c_int32),
        ("colsum_out", c_fp), ("strideColsum", c_i64), ("colsum_of", C.c_int32),
        ("precision", C.c_int32),
    ]


class PackJob(C.Structure):
    """mirror of struct dpot_pack_job"""
    _fields_ = [("src", c_fp), ("dst", c_fp), ("rows", C.c_int32), ("K", C.c_int32), ("ld", C.c_int32),
                ("trans", C.c_int32)]


class AdamPackJob(C.Structure):
    """mirror of struct dpot_adam_pack_job"""
    _fields_ = [("off", C.c_int64), ("dst_rows", c_fp), ("dst_trans", c_fp), ("R", C.c_int32), ("K", C.c_int32),
                ("tile0", C.c_int32), ("pad_", C.c_int32)]


class AdamRange(C.Structure):
    """mirror of struct dpot_adam_range"""
    _fields_ = [("start", C.c_int64), ("len", C.c_int64)]


class LayoutJob(C.Structure):
    """mirror of struct dpot_layout_job"""
    _fields_ = [("src", c_fp), ("add", c_fp), ("dst", c_fp), ("d0", C.c_int32), ("d1", C.c_int32), ("d2", C.c_int32),
                ("v0", C.c_int32), ("v1", C.c_int32), ("v2", C.c_int32), ("s0", C.c_int64), ("s1", C.c_int64),
                ("s2", C.c_int64)]


class AfnoPackJob(C.Structure):
    """mirror of struct dpot_afno_pack_job"""
    _fields_ = [("w", c_fp), ("b", c_fp), ("wbig", c_fp), ("bbig", c_fp), ("fwd", c_fp), ("bwd", c_fp)]


class SampleDesc(C.Structure):
    """mirror of struct dpot_sample_desc"""
    _fields_ = [("data", c_fp), ("H", C.c_int32), ("W", C.c_int32), ("T", C.c_int32), ("C", C.c_int32),
                ("t0", C.c_int32), ("reserved", C.c_int32)]


# name -> (restype, argtypes); every symbol include/dpot_hip.h declares
SIGNATURES = {
    "dpot_version": (c_i, []),
    "dpot_tune": (c_i, [C.c_char_p, c_i]),
    "dpot_last_error": (C.c_char_p, []),
    "dpot_gemm_f32": (c_i, [C.POINTER(GemmDesc), c_fp]),
    "dpot_gemm_workspace_bytes": (c_i64, [C.POINTER(GemmDesc)]),
    "dpot_gemm_auto_splitk": (c_i, [c_i, c_i, c_i, c_i]),
    "dpot_gemm_auto_splitk2": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "dpot_rfft2": (c_i, [c_fp, c_fp] + [c_i] * 8 + [c_fp]),
    "dpot_rfft2_norm_supported": (c_i, [c_i, c_i, c_i]),
    "dpot_rfft2_norm": (c_i, [c_fp] * 5 + [c_i, c_fp] + [c_i] * 8 + [c_fp]),
    "dpot_irfft2_norm": (c_i, [c_fp] * 6 + [c_i, c_fp] + [c_i] * 8 + [c_fp]),
    "dpot_irfft2": (c_i, [c_fp, c_fp, c_fp] + [c_i] * 8 + [c_fp]),
    "dpot_afno_pack": (c_i, [c_fp] * 4 + [c_i, c_i, c_fp]),
    "dpot_afno_pack_multi": (c_i, [C.POINTER(C.c_void_p)] * 4 + [c_i, c_i, c_i, c_fp]),
    "dpot_afno_unpack_grad": (c_i, [c_fp] * 4 + [c_i, c_i, c_fp]),
    "dpot_afno_mlp2_supported": (c_i, [c_i, c_i]),
    "dpot_afno_mlp2": (c_i, [c_fp] * 9 + [c_i] * 8 + [c_fp]),
    "dpot_afno_mlp3_supported": (c_i, [c_i, c_i]),
    "dpot_afno_mlp6_supported": (c_i, [c_i, c_i]),
    "dpot_afno_pack6_elems": (c_i64, [c_i, c_i]),
    "dpot_afno_pack6": (c_i, [c_fp] * 3 + [c_i] * 3 + [c_fp]),
    "dpot_afno_mlp6": (c_i, [c_fp] * 9 + [c_i] * 7 + [c_fp]),
    "dpot_afno_block_weights": (c_i, [c_fp] * 3 + [c_i, c_i, c_fp]),
    "dpot_afno_pack_all": (c_i, [c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "dpot_groupnorm_ws_elems": (c_i64, [c_i] * 4),
    "dpot_groupnorm_fwd": (c_i, [c_fp] * 7 + [c_i] * 4 + [c_f, c_fp]),
    "dpot_groupnorm_bwd": (c_i, [c_fp] * 11 + [c_i] * 4 + [c_fp]),
    "dpot_groupnorm_bwd_packs_supported": (c_i, [c_i] * 3),
    "dpot_groupnorm_bwd_packs_rows": (c_i, [c_i] * 4),
    "dpot_groupnorm_bwd_packs": (c_i, [c_fp] * 12 + [c_i] * 4 + [c_fp]),
    "dpot_groupnorm_param_grads": (c_i, [C.c_void_p] * 3 + [c_i] * 3 + [c_fp]),
    "dpot_afno_fused_supported": (c_i, [c_i] * 7),
    "dpot_afno_fused_fwd": (c_i, [c_fp] * 19 + [c_i] * 9 + [c_f, c_fp]),
    "dpot_gn_dft_supported": (c_i, [c_i] * 4),
    "dpot_gn_rfft2": (c_i, [c_fp] * 6 + [c_i] * 8 + [c_f, c_fp]),
    "dpot_irfft2_gn": (c_i, [c_fp] * 12 + [c_i] * 9 + [c_f, c_fp]),
    "dpot_gn_bwd_rfft2": (c_i, [c_fp] * 8 + [c_i] * 9 + [c_fp]),
    "dpot_irfft2_gn_bwd": (c_i, [c_fp] * 9 + [c_i] * 9 + [c_fp]),
    "dpot_patchify": (c_i, [c_fp] * 5 + [c_i] * 6 + [c_fp]),
    "dpot_unpatchify": (c_i, [c_fp] * 2 + [c_i] * 6 + [c_fp]),
    "dpot_pixel_shuffle": (c_i, [c_fp] * 2 + [c_i] * 6 + [c_fp]),
    "dpot_copy2d_pad": (c_i, [c_fp, c_i, c_i, c_fp, c_i, c_i, c_fp]),
    "dpot_transpose2d": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    "dpot_colsum_parts": (c_i, [c_i]),
    "dpot_colsum": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "dpot_colsum_scatter": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_void_p), c_fp]),
    "dpot_group_rowsum": (c_i, [c_fp, c_fp] + [c_i] * 4 + [c_fp]),
    "dpot_small_linear_supported": (c_i, [c_i] * 3),
    "dpot_small_linear": (c_i, [c_fp, c_i, c_fp, c_i, c_fp, c_fp, c_fp] + [c_i] * 5 + [c_fp]),
    "dpot_token_mean": (c_i, [c_fp, c_fp] + [c_i] * 3 + [c_fp]),
    "dpot_token_mean_bwd": (c_i, [c_fp] * 3 + [c_i] * 3 + [c_fp]),
    "dpot_bias_add": (c_i, [c_fp] * 3 + [c_i, c_i, c_fp]),
    "dpot_scale_shift": (c_i, [c_fp] * 4 + [c_i] * 3 + [c_fp]),
    "dpot_scale_shift_bwd": (c_i, [c_fp] * 6 + [c_i] * 3 + [c_fp]),
    "dpot_timeagg_scale_w": (c_i, [c_fp] * 4 + [c_i, c_i, c_fp]),
    "dpot_timeagg_scale_w_bwd": (c_i, [c_fp] * 6 + [c_i, c_i, c_fp]),
    "dpot_out_tail_partial_rows": (c_i, [c_i] * 4),
    "dpot_out_tail_partial_cols": (c_i, []),
    "dpot_out_tail_fwd": (c_i, [c_fp] * 6 + [c_i] * 6 + [c_fp]),
    "dpot_out_tail_bwd": (c_i, [c_fp] * 7 + [c_i] * 6 + [c_fp]),
    "dpot_rel_l2_chunks": (c_i, [c_i, c_i]),
    "dpot_rel_l2_fwd": (c_i, [c_fp] * 5 + [c_i] * 4 + [c_fp]),
    "dpot_rel_l2_bwd": (c_i, [c_fp] * 6 + [c_i] * 4 + [c_fp]),
    "dpot_sumsq": (c_i, [c_fp, c_i64, c_fp, c_fp, c_i, c_fp]),
    "dpot_adam_step": (c_i, [c_fp] * 4 + [c_i64, c_fp, c_fp, c_f, c_fp]),
    "dpot_adam_stage": (c_i, [c_fp, c_fp, c_f, c_d, c_d, c_f, c_f, c_f, c_i, c_fp]),
    "dpot_noise_chunks": (c_i, [c_i, c_i]),
    "dpot_noise_inject": (c_i, [c_fp] * 4 + [c_f] + [c_i] * 3 + [c_fp]),
    "dpot_noise_inject_rng": (c_i, [c_fp] * 4 + [c_f] + [c_i] * 3 + [c_fp]),
    "dpot_noise_inject_bwd": (c_i, [c_fp] * 7 + [c_f] + [c_i] * 3 + [c_fp]),
    "dpot_window_slide": (c_i, [c_fp] * 3 + [c_i64] + [c_i] * 3 + [c_fp]),
    "dpot_window_slide_bwd": (c_i, [c_fp] * 3 + [c_i64] + [c_i] * 3 + [c_fp]),
    "dpot_resize_pad_window": (c_i, [c_fp, c_i, c_fp, c_fp] + [c_i] * 6 + [c_fp]),
    "dpot_panel_pack_weights": (c_i, [c_fp, c_i, c_i, c_fp]),
    "dpot_layout_jobs": (c_i, [c_fp, c_i, c_i64, c_fp]),
    "dpot_bf16_packed_elems": (c_i64, [c_i, c_i, c_i]),
    "dpot_bf16_pack_rows": (c_i, [c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "dpot_bf16_pack_jobs": (c_i, [c_fp, c_i, c_i, c_i, c_fp]),
    "dpot_adam_pack_supported": (c_i, [c_i, c_i]),
    "dpot_adam_step_packs": (c_i, [c_fp] * 6 + [c_f, c_fp, c_fp, c_i, c_fp, c_i, c_i, c_fp]),
    "dpot_gemm_bf16p_supported": (c_i, [c_i, c_i, c_i]),
    "dpot_gemm_bf16p": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_i, c_fp, c_i, c_fp, c_i] + [c_i] * 7 + [c_fp] * 7),
    "dpot_gemm_bf16p_splitk": (c_i, [c_i, c_i, c_i]),
    "dpot_gemm_bf16p_pair_wanted": (c_i, [c_i] * 5),
    "dpot_gemm_bf16p_pair_splitk": (c_i, [c_i] * 5),
    "dpot_gemm_bf16p_kernel_kind": (c_i, [c_i] * 6),
    "dpot_gemm_bf16p_pair": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "dpot_gemm_tn_splitk": (c_i, [c_i] * 4),
    "dpot_mlp_wgrad2_splitk": (c_i, [c_i] * 3),
    "dpot_mlp_wgrad2_ws_elems": (c_i64, [c_i] * 3),
    "dpot_mlp_wgrad2": (c_i, [c_fp] * 4 + [c_i] * 3 + [c_fp] * 5 + [c_i, c_fp]),
    "dpot_afno_wgrad2_splitk": (c_i, [c_i] * 3),
    "dpot_afno_wgrad2_ws_elems": (c_i64, [c_i] * 3),
    "dpot_afno_wgrad2": (c_i, [c_fp] * 4 + [c_i] * 4 + [c_fp] * 5 + [c_i, c_fp]),
    "dpot_block_finalize": (c_i, [c_fp, c_i, c_i, c_i] + [c_fp] * 4 + [c_fp, c_i, c_i, c_i] + [c_fp] * 4 + [c_fp] * 3
                            + [c_i, c_i, c_i] + [c_fp] * 4 + [c_i, c_fp]),
    "dpot_bf16_pack_both_supported": (c_i, [c_i, c_i]),
    "dpot_bf16_pack_both": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "dpot_bf16_pack_both_norm": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_fp, c_fp]),
    "dpot_embed_supported": (c_i, [c_i] * 5),
    "dpot_embed_wfrag_elems": (c_i, []),
    "dpot_embed_pack_w0": (c_i, [c_fp, c_i, c_fp, c_fp]),
    "dpot_embed_fwd": (c_i, [c_fp] * 5 + [c_i] * 6 + [c_fp]),
    "dpot_embed_wgrad_ws_elems": (c_i, [c_i] * 3),
    "dpot_embed_wgrad": (c_i, [c_fp] * 4 + [c_i] * 7 + [c_fp]),
    "dpot_gemm_panel_supported": (c_i, [c_i, c_i, c_i]),
    "dpot_gemm_panel": (c_i, [c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_fp, c_i, c_fp, c_i, c_fp, c_i] + [c_i] * 5 + [c_fp]),
}

_lib = None
_lock = threading.Lock()


class DpotHipError(RuntimeError):
    pass


def lib_path() -> str:
    return LIB_PATH


def load():
    """dlopen libdpot_hip.so and bind every symbol.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise DpotHipError(
                f"{LIB_PATH} is missing - build it first (python -m dpot_amd.build, or __graft_entry__.build()). "
                "dpot_amd has no CPU / eager fallback by design.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().dpot_last_error()
        raise DpotHipError(f"libdpot_hip {what} failed (rc={rc}): {msg.decode() if msg else '?'}")
