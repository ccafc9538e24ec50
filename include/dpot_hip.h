/*
 * dpot_hip.h - C ABI of libdpot_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the DPOT
 * auto-regressive forward/backward step.
 *
 * The reference (HaoZhongkai/DPOT) has no FFI/plugin layer: its hot path is Python on top of PyTorch
 * ATen.  This header is therefore the "inner face" of the drop-in boundary: the
 * entry points a maintainer binds (ctypes - see INTEGRATION.md) to replace the ATen calls made by
 *   models/dpot.py:51-110   AFNO2D.forward      -> dpot_rfft2 / dpot_gemm_f32 / dpot_irfft2
 *   models/dpot.py:142,152  GroupNorm(8, width) -> dpot_groupnorm_fwd / _bwd
 *   models/dpot.py:157-161  Block.mlp (1x1 conv)-> dpot_gemm_f32 (bias + activation epilogues)
 *   models/dpot.py:198-202  PatchEmbed          -> dpot_patchify + dpot_gemm_f32
 *   models/dpot.py:226-234  TimeAggregator      -> dpot_timeagg_scale_w + dpot_gemm_f32
 *   models/dpot.py:315-321  out_layer           -> dpot_gemm_f32 + dpot_pixel_shuffle
 *   utils/criterion.py:38-59 SimpleLpLoss       -> dpot_rel_l2_fwd / _bwd
 *   utils/optimizer.py:9-52 adam()              -> dpot_sumsq + dpot_adam_step
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise
 *   - all tensors are fp32; activations are channels-last: [B, h, w, E] == row-major [B*h*w, E]
 *   - the caller owns every buffer (including workspaces); the library never allocates or frees
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*) and never synchronise, so every
 *     call is legal inside HIP stream capture (hipGraph)
 *   - return value: 0 on success, negative DPOT_E* on error; dpot_last_error() gives the message
 *   - re-entrant and stateless (no global state besides the thread-local error string)
 */
#ifndef DPOT_HIP_H
#define DPOT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPOT_OK 0
#define DPOT_EINVAL (-1)   /* bad shape / argument            */
#define DPOT_EUNSUP (-2)   /* unsupported size                */
#define DPOT_EHIP (-3)     /* HIP runtime error on launch     */

typedef void* dpot_stream_t; /* hipStream_t */

/* activation ids (models/dpot.py:19 ACTIVATION table) */
enum {
  DPOT_ACT_NONE = 0,
  DPOT_ACT_GELU = 1,      /* exact erf GELU (nn.GELU() default) */
  DPOT_ACT_TANH = 2,
  DPOT_ACT_SIGMOID = 3,
  DPOT_ACT_RELU = 4,
  DPOT_ACT_LEAKY_RELU = 5, /* slope 0.1 */
  DPOT_ACT_SOFTPLUS = 6,
  DPOT_ACT_ELU = 7,
  DPOT_ACT_SILU = 8
};

/* epilogue modes of dpot_gemm_f32 */
enum {
  DPOT_EPI_LINEAR = 0, /* v                                  */
  DPOT_EPI_ACT = 1,    /* act(v)                             */
  DPOT_EPI_DACT = 2,   /* v * act'(aux[m,n])  (backward)     */
  /* weight gradient of the AFNO block-diagonal complex MLP (split-K only, M = N = 2*bs, batch = nb): the split-K
   * reduction also undoes the Wbig = [[Wr, Wi], [-Wi, Wr]] packing - C receives dw[2, nb, bs, bs]
   * (dWr = TL + BR, dWi = TR - BL of the 2bs x 2bs product) and colsum_out (colsum_of = 2) receives db[2, nb, bs] */
  DPOT_EPI_AFNO_WGRAD = 3
};

int dpot_version(void);
/* value of `key` in the environment variable DPOT_TUNE="key=val,key=val" (integers), or dflt: the one switchboard of the
 * library's fallback-path selectors (keys: dpot_amd/ops.py TUNE_KEYS); read once per process */
int dpot_tune(const char* key, int dflt);
const char* dpot_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
 *
 *   for b in [0,batch):   C_b[m,n] = epilogue( sum_k A_b[m,k] * B_b[k,n] )
 *
 *   transA = 0: A_b is [M,K] row-major (lda = row stride)   transA = 1: A_b is [K,M] row-major
 *   transB = 0: B_b is [K,N] row-major (ldb = row stride)   transB = 1: B_b is [N,K] row-major
 *   X_b = X + b * strideX  (in floats)
 *
 *   epilogue, in this order:
 *       v  = acc
 *       v += bias[b*strideBias + n]                               (bias != NULL)
 *       preact[b*stridePre + m*ldpre + n] = v                      (preact != NULL)
 *       v  = act(v)               (epi_mode == DPOT_EPI_ACT)
 *       v  = v * act'(aux[m,n])   (epi_mode == DPOT_EPI_DACT; aux = saved pre-activation)
 *       v += res[b*strideRes + ((m / res_div) % res_mod) * ldres + n]   (res != NULL)
 *       v += C_b[m,n]             (accumulate != 0)
 *       C_b[m*ldc + n] = v
 *
 *   splitk > 1 partitions K over `splitk` workgroups per tile; partial sums go to `workspace`
 *   (splitk * batch * M * N floats) and a second kernel reduces them in a fixed order (deterministic)
 *   and applies the epilogue.
 * ------------------------------------------------------------------------------------------------ */
typedef struct dpot_gemm_desc {
  const float* A;
  const float* B;
  float* C;
  int32_t M, N, K;
  int32_t batch;
  int32_t transA, transB;
  int32_t lda, ldb, ldc;
  int64_t strideA, strideB, strideC;
  const float* bias;
  int64_t strideBias;
  int32_t act;      /* DPOT_ACT_*  */
  int32_t epi_mode; /* DPOT_EPI_*  */
  const float* aux;
  int32_t ldaux;
  int64_t strideAux;
  float* preact;
  int32_t ldpre;
  int64_t stridePre;
  const float* res;
  int32_t ldres;
  int32_t res_div, res_mod; /* 0 -> plain residual (res_div = 1, res_mod = infinity) */
  int64_t strideRes;
  int32_t accumulate;
  int32_t splitk;
  float* workspace;
  int32_t tile; /* 0 = auto, 64 or 128 = force BMxBN tile */
  int32_t tag;  /* 1 = launch the separately-named AFNO-mixer instantiation (profiling identity only) */
  /* fused bias gradient: colsum_out[b*strideColsum + j] = sum_k of operand column j;  colsum_of = 1: columns of A
   * (needs transA = 1, j < M), 2: columns of B (needs transB = 0, j < N), 0: off.  With split-K the workspace grows by
   * splitk*batch*L floats (dpot_gemm_workspace_bytes accounts for it). */
  float* colsum_out;
  int64_t strideColsum;
  int32_t colsum_of;
  /* how the fp32 products are formed (both accumulate in fp32 and meet the rtol 1e-4 contract):
   *   DPOT_GEMM_F32    v_mfma_f32_32x32x2_f32, native fp32 operands
   *   DPOT_GEMM_BF16X6 fp32 emulated on the bf16 matrix cores: each operand is split a = a1+a2+a3 (bf16 each, exact
   *                    to 2^-25) and the 6 partial products down to 2^-24 are accumulated - fp32-level accuracy at
   *                    2.67x the fp32 MFMA rate (csrc/gemm_split.h)
   *   DPOT_GEMM_AUTO   the library picks per shape (bf16x6 for products >= 3 GFLOP, where its longer pipeline pays)
   *   DPOT_GEMM_BF16   REDUCED precision, only on explicit request (BASELINE configs[2]: "bf16 channel-MLP on MFMA"):
   *                    operands rounded to bf16, one product per k-step, fp32 accumulation (relative error ~4e-3) */
  int32_t precision;
} dpot_gemm_desc;
enum { DPOT_GEMM_F32 = 0, DPOT_GEMM_BF16X6 = 1, DPOT_GEMM_AUTO = 2, DPOT_GEMM_BF16 = 3 };

int dpot_gemm_f32(const dpot_gemm_desc* d, dpot_stream_t stream);
/* bytes of workspace dpot_gemm_f32 needs for this descriptor (0 when splitk <= 1) */
int64_t dpot_gemm_workspace_bytes(const dpot_gemm_desc* d);
/* weight gradients (transA = 1, transB = 0, native fp32, M % 128 == N % 128 == 0, K % 16 == 0, split-K) run on the
 * dedicated kernel of csrc/gemm_tn.hip; this is the split count it wants for a shape (0: shape not eligible - use
 * dpot_gemm_auto_splitk2).  Any splitk > 1 is accepted, the result does not depend on which kernel ran beyond fp32
 * summation order. */
int dpot_gemm_tn_splitk(int M, int N, int K, int batch);
/* Both weight gradients of an AFNO block's complex MLP (models/dpot.py:72-94) in one launch + one fixed-order reduce
 * (csrc/gemm_tn.hip): per channel block k  dWbig1[k] = S[:,k]^T dO1pre[:,k], dWbig2[k] = O1[:,k]^T dO2[:,k], un-packed
 * into the parameters' own layout dw [2, nb, bs, bs] / db [2, nb, bs] (db = column sums of the dO operand).
 * S / dO1pre / O1 / dO2: [Mm, ld] planar-per-block spectra.  Needs 2*bs % 128 == 0 and Mm % 32 == 0
 * (dpot_afno_wgrad2_splitk returns 0 otherwise); workspace: dpot_afno_wgrad2_ws_elems(nb, bs, splitk) floats. */
/* Both weight gradients of a channel MLP y = fc2(act(fc1(x))) (models/dpot.py:157-161) in one launch + one reduce:
 * dW2 [E, mh] = do2^T Hh, db2 = colsum(do2), dW1 [mh, E] = dHpre^T xn2, db1 = colsum(dHpre); do2 / xn2 [T, E],
 * Hh / dHpre [T, mh], all contiguous.  E, mh % 128 == 0, T % 32 == 0 (dpot_mlp_wgrad2_splitk returns 0 otherwise, and
 * for layers whose single weight gradient already fills the chip). */
int dpot_mlp_wgrad2_splitk(int T, int E, int mh);
int64_t dpot_mlp_wgrad2_ws_elems(int E, int mh, int splitk);
int dpot_mlp_wgrad2(const float* do2, const float* Hh, const float* xn2, const float* dHpre, int T, int E, int mh,
                    float* dW2, float* db2, float* dW1, float* db1, float* workspace, int splitk, dpot_stream_t stream);
int dpot_afno_wgrad2_splitk(int Mm, int nb, int bs);
int64_t dpot_afno_wgrad2_ws_elems(int nb, int bs, int splitk);
int dpot_afno_wgrad2(const float* S, const float* dO1pre, const float* O1, const float* dO2, int ld, int Mm, int nb,
                     int bs, float* dw1, float* db1, float* dw2, float* db2, float* workspace, int splitk,
                     dpot_stream_t stream);
/* dpot_afno_wgrad2 / dpot_mlp_wgrad2 with ALL FOUR outputs NULL write their split-K partials to the workspace only; ONE
 * finalising launch per DPOT block then runs, in three slices of one grid, the fixed-order reductions that end the block's
 * backward (models/dpot.py:165-180): the AFNO weight-gradient partials (afno_ws != NULL; outputs as dpot_afno_wgrad2), the
 * channel-MLP ones (mlp_ws != NULL; as dpot_mlp_wgrad2) and the GroupNorm parameter gradients of gn_jobs <= 2 layers
 * (as dpot_groupnorm_param_grads) - same summation orders as the stand-alone reductions: bit-identical results - and the column
 * sums out[n] = sum_r part[r, n] of cs_jobs <= 2 partial matrices [cs_rows, cs_cols] (the bias gradients of the bf16 channel
 * MLP, whose pack pass / GEMM epilogue leave per-row-tile partial sums; fixed order: four accumulators over r mod 4). */
int dpot_block_finalize(const float* afno_ws, int afno_splitk, int nb, int bs, float* dw1, float* db1, float* dw2,
                        float* db2, const float* mlp_ws, int mlp_splitk, int E, int mh, float* dW2, float* dfb2, float* dW1,
                        float* dfb1, const float* const* gn_parts, float* const* gn_dgammas, float* const* gn_dbetas,
                        int gn_jobs, int B, int Egn, const float* const* cs_parts, float* const* cs_outs, const int* cs_rows,
                        const int* cs_cols, int cs_jobs, dpot_stream_t stream);
/* Small weight-only layout jobs (zero-padded copies, small transposes, bias broadcasts, "+ bias") in ONE launch from a DEVICE
 * table: dst[i0][i1][i2] (contiguous d0 x d1 x d2) = (inside v0 x v1 x v2 ? src[i0 s0 + i1 s1 + i2 s2] : 0) + (add ? add[i2] : 0).
 * The pieces DPOTNet derives from its parameters once per optimiser step (models/dpot.py:198-202 padded for the MFMA
 * kernels, :378 pos_embed + bias, :315-321 de-embed bias / tail weights).  max_elems = the largest d0*d1*d2.  72 bytes. */
typedef struct dpot_layout_job {
  const float* src;
  const float* add;     /* [d2] or NULL */
  float* dst;
  int32_t d0, d1, d2, v0, v1, v2;
  int64_t s0, s1, s2;
} dpot_layout_job;
int dpot_layout_jobs(const dpot_layout_job* jobs_dev, int njobs, int64_t max_elems, dpot_stream_t stream);
/* the split-K factor the library would pick for this shape (>= 1) */
int dpot_gemm_auto_splitk(int M, int N, int K, int batch);
/* the same for a given dpot_gemm_desc.precision (the bf16x6 kernel prefers fewer, larger workgroups) */
int dpot_gemm_auto_splitk2(int M, int N, int K, int batch, int precision);

/* ------------------------------------------------------------------------------------------------
 * rfft2 / irfft2, norm="ortho", over the two spatial axes of a channels-last field, done as two in-LDS
 * direct DFT passes (latent grids are 16x16 / 32x32: a dense DFT is cheaper than a strided FFT plan).
 *
 * spectrum layout ("planar per channel block"): spec[B, mx, my, nb, 2, bs]  (2 = re, im; bs = E/nb)
 *   -> row (b,kx,ky) of the [B*mx*my, 2E] matrix holds, per channel block, [re(bs) | im(bs)], which is
 *      exactly the A operand of the block-diagonal complex MLP written as a real GEMM with
 *      Wbig = [[Wr, Wi], [-Wi, Wr]].
 * Only modes kx < mx, ky < my are produced / consumed (models/dpot.py:70-94 keeps [:modes,:modes] of the
 * half spectrum; everything else is zero).
 *
 * col_weights: 0 -> every ky column weight 1            (true rfft2 / adjoint of rfft2)
 *              1 -> interior ky columns weight 2         (true irfft2 / adjoint of irfft2)
 *   dpot_rfft2 (x, w=0) = rfft2(x)           dpot_rfft2 (g, w=1) = d irfft2 / d spectrum  applied to g
 *   dpot_irfft2(S, w=1) = irfft2(S, s=(h,w))  dpot_irfft2(G, w=0) = d rfft2 / d x applied to G
 * dpot_irfft2 optionally adds `res` ([B,h,w,E]) - the "+ x_orig" of models/dpot.py:106.
 * ------------------------------------------------------------------------------------------------ */
int dpot_rfft2(const float* x, float* spec, int B, int h, int w, int E, int nb, int mx, int my,
               int col_weights, dpot_stream_t stream);
int dpot_irfft2(const float* spec, const float* res, float* y, int B, int h, int w, int E, int nb, int mx,
                int my, int col_weights, dpot_stream_t stream);
/* rfft2(GroupNorm(x)) with the statistics given (mean / rstd [B, G] from dpot_groupnorm_fwd with y == NULL): norm1 of
 * Block.forward (models/dpot.py:167-168) folded into the load of the transform, GroupNorm1(x) is never written.  Register-FFT
 * grids only (8, 16, 32, 64 square): dpot_rfft2_norm_supported. */
int dpot_rfft2_norm_supported(int h, int w, int E);
int dpot_rfft2_norm(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int G,
                    float* spec, int B, int h, int w, int E, int nb, int mx, int my, int col_weights,
                    dpot_stream_t stream);
/* irfft2(spec) + GroupNorm(res): the AFNO residual x_orig = norm1(x) (models/dpot.py:55,106) re-derived on the load */
int dpot_irfft2_norm(const float* spec, const float* res, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, int G, float* y, int B, int h, int w, int E, int nb, int mx, int my,
                     int col_weights, dpot_stream_t stream);

/* AFNO weight packing: w[2,nb,bs,bs], b[2,nb,bs] -> Wbig[nb,2bs,2bs] = [[Wr,Wi],[-Wi,Wr]], bbig[nb,2,bs]
 * and the adjoint (gradients back to the reference layout).  models/dpot.py:45-48,72-94 */
int dpot_afno_pack(const float* w, const float* b, float* wbig, float* bbig, int nb, int bs,
                   dpot_stream_t stream);
/* njobs packs in one launch (all AFNO layers of a model): HOST arrays of njobs device pointers each */
int dpot_afno_pack_multi(const float* const* w, const float* const* b, float* const* wbig, float* const* bbig,
                         int njobs, int nb, int bs, dpot_stream_t stream);
int dpot_afno_unpack_grad(const float* dwbig, const float* dbbig, float* dw, float* db, int nb, int bs,
                          dpot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm(G, E) on channels-last x[B, T, E] (T = h*w tokens), eps inside rsqrt.  models/dpot.py:142,152
 * fwd: y = (x - mean[b,g]) * rstd[b,g] * gamma[c] + beta[c]; mean/rstd [B,G] are saved for backward.
 * bwd: dx = rstd * (gamma*dy - mean_g(gamma*dy) - xhat * mean_g(gamma*dy*xhat)) (+ add), and
 *      dgamma/dbeta via per-sample partials part[2,B,E] that are then reduced over B in a fixed order.
 * ------------------------------------------------------------------------------------------------ */
/* workspace: dpot_groupnorm_ws_elems(B,T,E,G) floats (0: none needed, NULL is fine).  It is used when there are few,
 * large (sample, group) slabs (DPOT-L at 256^2, B = 4: 32 slabs of 768 KiB): the slabs are cut into token chunks so
 * that >= 512 workgroups run, and the group statistics are merged through the workspace (two launches per call).
 * Passing NULL always selects the one-workgroup-per-slab kernels. */
int64_t dpot_groupnorm_ws_elems(int B, int T, int E, int G);
int dpot_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                       float* rstd, float* workspace, int B, int T, int E, int G, float eps, dpot_stream_t stream);
int dpot_groupnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                       const float* gamma, const float* add, float* dx, float* dgamma, float* dbeta,
                       float* part /* [2,B,E] */, float* workspace, int B, int T, int E, int G, dpot_stream_t stream);
/* round 5: the same backward (partials left in `part`) that ALSO writes dx - the gradient entering the previous block's
 * channel-MLP backward - as that backward's bf16 operands: dx_rows_bf16 / dx_trans_bf16 = the two 1-plane packs
 * dpot_bf16_pack_both(dx) would write (dpot_bf16_packed_elems(B*T, E, 1) elements each) and dx_colsum [B * rows, E] = column
 * sums of dx over `rows` token ranges per sample (bias gradient partials; reduce over the first dimension, e.g. as a
 * column-sum job of dpot_block_finalize).  rows = dpot_groupnorm_bwd_packs_rows(B, T, E, G): 1 where one workgroup holds a
 * (sample, group) slab (128 channels per group, T <= 256: DPOT-S / -M at 128^2), the number of token chunks where the slab is
 * chunked (DPOT-L: 192 channels per group, 1024 tokens; needs `workspace` of dpot_groupnorm_ws_elems), 0 = not supported
 * (T % 32 and E % 32 must be 0). */
int dpot_groupnorm_bwd_packs_rows(int B, int T, int E, int G);
int dpot_groupnorm_bwd_packs_supported(int T, int E, int G);   /* the one-workgroup-per-slab form alone */
int dpot_groupnorm_bwd_packs(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                             const float* add, float* dx, float* part /* [2,B,E] */, void* dx_rows_bf16,
                             void* dx_trans_bf16, float* dx_colsum /* [B*rows,E] */, float* workspace, int B, int T, int E,
                             int G, dpot_stream_t stream);
/* dgamma == dbeta == NULL above leaves the per-sample partials in `part`; this reduces up to 4 such partial sets
 * (HOST arrays of njobs pointers; e.g. the two GroupNorm layers of a block) in ONE launch. */
int dpot_groupnorm_param_grads(const float* const* parts, float* const* dgammas, float* const* dbetas, int njobs,
                               int B, int E, dpot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm fused with the neighbouring DFT of the AFNO mixer (csrc/gn_dft.hip; models/dpot.py:165-175 runs
 * norm1 -> AFNO2D(rfft2 .. irfft2 + x_orig) -> norm2 as separate modules).  16 x 16 latent grid, E/G in {64, 128}
 * (dpot_gn_dft_supported); layouts as dpot_groupnorm_* / dpot_rfft2 / dpot_irfft2.
 *   gn_rfft2       spec = rfft2_ortho(GroupNorm(x; gamma, beta)), kept modes; mean / rstd [B,G] out.  GroupNorm(x)
 *                  itself is not written: consumers re-derive it from x and the statistics.
 *   irfft2_gn      y1 = irfft2_ortho(spec; col_weights) + GroupNorm1(x) (from x, mean1, rstd1, gamma1, beta1);
 *                  xn2 = GroupNorm2(y1); mean2 / rstd2 out.
 *   gn_bwd_rfft2   dx = GroupNorm backward of dy at input xin (statistics given), part [2,B,E] = per-sample partials
 *                  of (dgamma, dbeta) as dpot_groupnorm_bwd leaves them; spec = rfft2_ortho(dx; col_weights).
 *   irfft2_gn_bwd  d = irfft2_ortho(spec; col_weights) + res, dx = GroupNorm backward of d at input xin (+ add), part.
 * ------------------------------------------------------------------------------------------------ */
int dpot_gn_dft_supported(int h, int w, int E, int G);
int dpot_gn_rfft2(const float* x, const float* gamma, const float* beta, float* spec, float* mean, float* rstd, int B,
                  int h, int w, int E, int G, int nb, int mx, int my, float eps, dpot_stream_t stream);
int dpot_irfft2_gn(const float* spec, const float* x, const float* mean1, const float* rstd1, const float* gamma1,
                   const float* beta1, const float* gamma2, const float* beta2, float* y1, float* xn2, float* mean2,
                   float* rstd2, int B, int h, int w, int E, int G, int nb, int mx, int my, int col_weights, float eps,
                   dpot_stream_t stream);
int dpot_gn_bwd_rfft2(const float* dy, const float* xin, const float* mean, const float* rstd, const float* gamma,
                      float* dx, float* part, float* spec, int B, int h, int w, int E, int G, int nb, int mx, int my,
                      int col_weights, dpot_stream_t stream);
int dpot_irfft2_gn_bwd(const float* spec, const float* res, const float* xin, const float* mean, const float* rstd,
                       const float* gamma, const float* add, float* dx, float* part, int B, int h, int w, int E, int G,
                       int nb, int mx, int my, int col_weights, dpot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * One AFNO layer's forward in ONE launch (csrc/afno_fused.hip; models/dpot.py:59-102, :165-175):
 *   [GroupNorm1] -> rfft2_ortho -> block-diagonal complex 2-layer MLP -> irfft2_ortho -> + x_orig -> [GroupNorm2]
 * = dpot_gn_rfft2 + dpot_afno_mlp2(layout 1, mode 0) + dpot_irfft2_gn without the spectrum / mixer output ever leaving
 * the chip.  16 x 16 latent grid, E / nb == 128, every mode kept (mx = 16, my = 9), E / G in {64, 128}
 * (dpot_afno_fused_supported; G = 0 asks about the norm-free form).  Wa / Wb: the layout-1 `fwd` packs and ba / bb the
 * bbig rows of dpot_afno_pack_all.  Outputs, each in the layout of the three-launch path: S = the spectrum and pre = the
 * layer-1 pre-activation [B*144, 2E] (saved for the backward; both may be NULL: inference), y1 = irfft2(..) + GN1(x),
 * xn2 = GN2(y1) (either may be NULL), statistics [B, G].  gamma1 == NULL: no norm1 (x_orig = x: the reference's AFNO2D
 * module alone); gamma2 == NULL: no norm2 (y1 only).  xn2_rows_bf16 / xn2_trans_bf16 (optional, with gamma2): GroupNorm2(y1) as
 * the two 1-plane bf16 operand packs of the channel MLP - exactly what dpot_bf16_pack_both_norm(y1, mean2, rstd2, ...) writes
 * (dpot_bf16_packed_elems(B*256, E, 1) elements each), so that pack launch and its re-read of y1 disappear.
 * ------------------------------------------------------------------------------------------------ */
int dpot_afno_fused_supported(int h, int w, int E, int G, int nb, int mx, int my);
int dpot_afno_fused_fwd(const float* x, const float* gamma1, const float* beta1, const float* Wa, const float* ba,
                        const float* Wb, const float* bb, const float* gamma2, const float* beta2, float* S, float* pre,
                        float* y1, float* xn2, float* mean1, float* rstd1, float* mean2, float* rstd2, void* xn2_rows_bf16,
                        void* xn2_trans_bf16, int B, int h, int w, int E, int G, int nb, int mx, int my, int act, float eps,
                        dpot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * data movement / small ops
 * ------------------------------------------------------------------------------------------------ */
/* x[B,X,Y,T,C] -> A[(b,px,py,t), (c,i,j)], c in [0,C+3): channels C..C+2 are the (x,y,t) unit grid
 * (models/dpot.py:350-360,374-375); gx/gy/gt are the grid coordinate tables (X, Y, T floats). */
int dpot_patchify(const float* x, const float* gx, const float* gy, const float* gt, float* A, int B, int X,
                  int Y, int T, int C, int P, dpot_stream_t stream);
/* adjoint w.r.t. x: dA[(b,px,py,t), (c,i,j)] -> dx[B,X,Y,T,C]  (grid columns ignored) */
int dpot_unpatchify(const float* dA, float* dx, int B, int X, int Y, int T, int C, int P,
                    dpot_stream_t stream);

/* ---- implicit-GEMM patch embedding (csrc/embed.hip): the P x P / stride-P conv of PatchEmbed.proj[0]
 * (models/dpot.py:198-202) gathered from x[B, X, Y, T, C] itself - no patch matrix.  Fast path for C = 4, P = 8, T <= 10,
 * hid <= 48, patch-grid width % 4 == 0 (dpot_embed_supported); other shapes use dpot_patchify + dpot_gemm_f32.
 *   wfrag: the DATA-channel weights w0[:, 0:C] in MFMA fragment order (dpot_embed_pack_w0, dpot_embed_wfrag_elems floats)
 *   btab [tok*T, hidp]: bias + the contribution of the three unit-grid channels (batch independent, one small GEMM per
 *                       optimiser step on the host side)
 *   fwd:   hpre[(b,px,py,t), n] = sum_{c<C,i,j} x[b,px*P+i,py*P+j,t,c] w0[n,c,i,j] + btab[(px,py,t), n];  hh = act(hpre)
 *   wgrad: dw0[n*ldw + c*P*P + i*P + j] = sum_rows dhpre[row, n] * x[...]   for n < hid, c < C   (deterministic) */
int dpot_embed_supported(int C, int P, int T, int hid, int w);
int dpot_embed_wfrag_elems(void);
int dpot_embed_pack_w0(const float* w0, int hid, float* wfrag, dpot_stream_t stream);
int dpot_embed_fwd(const float* x, const float* wfrag, const float* btab, float* hpre, float* hh, int B, int X, int Y,
                   int T, int hidp, int act, dpot_stream_t stream);
int dpot_embed_wgrad_ws_elems(int B, int X, int Y);
int dpot_embed_wgrad(const float* x, const float* dhpre, float* workspace, float* dw0, int ldw, int hid, int B, int X,
                     int Y, int T, int hidp, dpot_stream_t stream);
/* z[(b,px,py,i,j), Cc] <-> out[b, px*P+i, py*P+j, Cc]   (ConvTranspose2d k=s=P pixel order) */
int dpot_pixel_shuffle(const float* z, float* out, int B, int h, int w, int P, int Cc, int inverse,
                       dpot_stream_t stream);
/* dst[dR,dC] = src[0:dR,0:dC] zero-filled outside src[sR,sC]  (pad or crop a dense 2-D matrix) */
int dpot_copy2d_pad(const float* src, int sR, int sC, float* dst, int dR, int dC, dpot_stream_t stream);
/* dst[C,R] = src[R,C]^T   (batched: nbatch consecutive matrices) */
int dpot_transpose2d(const float* src, float* dst, int nbatch, int R, int C, dpot_stream_t stream);
/* out[n] = sum_m X[m*ld + n], m in [0,M): two-stage deterministic column sum; part = [parts, N] scratch
 * with parts = dpot_colsum_parts(M) */
int dpot_colsum_parts(int M);
int dpot_colsum(const float* X, int M, int N, int ld, float* out, float* part, dpot_stream_t stream);
/* the same sums, delivered to up to 8 destinations: columns [seg_start[s], seg_start[s]+seg_len[s]) -> seg_dst[s]
 * (HOST arrays of nseg entries; seg_dst holds device pointers).  part: dpot_colsum_parts(M) * N floats. */
int dpot_colsum_scatter(const float* X, int M, int N, int ld, float* part, int nseg, const int* seg_start,
                        const int* seg_len, float* const* seg_dst, dpot_stream_t stream);
/* out[r, n] = sum_{b,t} X[((b*R + r)*T + t)*N + n]   (pos_embed gradient: sum over batch and time) */
int dpot_group_rowsum(const float* X, float* out, int B, int R, int T, int N, dpot_stream_t stream);
/* y[b,e] = mean_t x[b,t,e]  and its adjoint dx[b,t,e] = dy[b,e]/T (+ add[b,t,e]) */
int dpot_token_mean(const float* x, float* y, int B, int T, int E, dpot_stream_t stream);
int dpot_token_mean_bwd(const float* dy, const float* add, float* dx, int B, int T, int E,
                        dpot_stream_t stream);
/* few-row Linear (reference: models/dpot.py:333-336, cls_head on the token mean; M = batch rows):
 * y[M, N] = act(x[M, K] W[N, K]^T + bias), pre (optional) receives the pre-activation.  One wave per output column,
 * fp32 FMA chains in a fixed order.  Supported: M <= 128, K a multiple of 512. */
int dpot_small_linear_supported(int M, int N, int K);
int dpot_small_linear(const float* x, int ldx, const float* W, int ldw, const float* bias, float* y, float* pre, int ldy,
                      int M, int N, int K, int act, dpot_stream_t stream);
/* y[r, n] = x[r, n] + v[n]   (row-broadcast add; pos_embed + conv bias folded ahead of the TimeAggregator);
 * x == NULL: y = v tiled R times (the ConvTranspose bias repeated per output pixel of a patch) */
int dpot_bias_add(const float* x, const float* v, float* y, int R, int N, dpot_stream_t stream);
/* y[b,t,e] = x[b,t,e] * scale[b,e] + shift[b,e]  (AdaIN, models/dpot.py:386-387) */
int dpot_scale_shift(const float* x, const float* scale, const float* shift, float* y, int B, int T, int E,
                     dpot_stream_t stream);
/* its adjoint: dx = dy * scale, dscale[b,e] = sum_t dy * x, dshift[b,e] = sum_t dy */
int dpot_scale_shift_bwd(const float* dy, const float* x, const float* scale, float* dx, float* dscale,
                         float* dshift, int B, int T, int E, dpot_stream_t stream);

/* TimeAggregator 'exp_mlp' (models/dpot.py:229-232): ws[t,i,j] = w[t,i,j] * cos(tt[t] * gamma[i]) and the
 * adjoint: dw = dws * cos(.), dgamma[i] = sum_{t,j} dws[t,i,j] * w[t,i,j] * (-sin(tt[t]*gamma[i])) * tt[t] */
int dpot_timeagg_scale_w(const float* w, const float* gamma, const float* tt, float* ws, int T, int E,
                         dpot_stream_t stream);
int dpot_timeagg_scale_w_bwd(const float* dws, const float* w, const float* gamma, const float* tt, float* dw,
                             float* dgamma, int T, int E, dpot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * fused per-pixel tail of the out layer (models/dpot.py:316-321 after the ConvTranspose), out_layer_dim == 32:
 *   out[b, px*P+i, py*P+j, :] = W4 act(W2 act(upre[p, :]) + b2) + b4,   p = ((b*h+px)*w+py)*P*P + i*P + j
 *   upre: [B*h*w*P*P, 32] pixel-major ConvTranspose output (bias included, pre-activation); co <= 32;
 *   w4 / b4 must be ZERO-PADDED to [32,32] / [32] by the caller (dpot_copy2d_pad): weight loads are unconditional
 * bwd: dupre = d loss / d upre; partials[rows, cols] (rows = dpot_out_tail_partial_rows, cols =
 *   dpot_out_tail_partial_cols = 2144) holds one row per wave: dW2[32*32] | dW4[32*32, rows >= co zero] | db2[32] |
 *   colsum(dupre)[32] | db4[32]; the caller reduces it over rows with dpot_colsum.
 * ------------------------------------------------------------------------------------------------ */
int dpot_out_tail_partial_rows(int B, int h, int w, int P);
int dpot_out_tail_partial_cols(void);
int dpot_out_tail_fwd(const float* upre, const float* w2, const float* b2, const float* w4, const float* b4,
                      float* out, int B, int h, int w, int P, int co, int act, dpot_stream_t stream);
int dpot_out_tail_bwd(const float* upre, const float* dout, const float* w2, const float* b2, const float* w4,
                      float* dupre, float* partials, int B, int h, int w, int P, int co, int act,
                      dpot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * loss / optimiser
 * ------------------------------------------------------------------------------------------------ */
/* masked relative L2 summed over the batch (SimpleLpLoss(size_average=False), utils/criterion.py:38-59).
 * x,y: [B, S, C] (S = X*Y*T), mask: [B, Sm, C] with S % Sm == 0 broadcast over the time axis (or NULL).
 * stats: (1 + dpot_rel_l2_chunks(S, C)) * B * C * 4 floats; the first [B, C, 4] block holds the final
 * {sum d^2, sum y^2, sum mask, the channel's loss term}, the rest is per-chunk scratch.  loss: 1 float.  bwd: dx = gloss[0] * dloss/dx. */
int dpot_rel_l2_chunks(int S, int C);
int dpot_rel_l2_fwd(const float* x, const float* y, const float* mask, float* stats, float* loss, int B,
                    int S, int C, int Tt, dpot_stream_t stream);
int dpot_rel_l2_bwd(const float* x, const float* y, const float* mask, const float* stats,
                    const float* gloss, float* dx, int B, int S, int C, int Tt, dpot_stream_t stream);

/* out[0] (+)= sum g[i]^2 over n floats (fixed-order two-stage reduction; part = 1024 floats scratch) */
int dpot_sumsq(const float* g, int64_t n, float* out, float* part, int accumulate, dpot_stream_t stream);
/* One Adam step over a flat fp32 buffer (utils/optimizer.py:26-52) with the clip of
 * train_temporal.py:228 folded in:  g' = g * grad_scale * min(1, max_norm / (sqrt(sumsq[0])*grad_scale + 1e-6));
 * g' += wd*p; m = b1*m + (1-b1)*g'; v = b2*v + (1-b2)*g'^2; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).
 * hyper (DEVICE, 8 floats) = {lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, max_norm};
 * sumsq may be NULL (no clipping). */
int dpot_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                   const float* sumsq, float grad_scale, dpot_stream_t stream);

/* Host side of an optimiser step (utils/optimizer.py:139-141: state['step'] += 1, bias corrections) as ONE one-thread
 * launch whose arguments travel by value: step[0] += advance (DEVICE int64), hyper = {lr, beta1, beta2, eps,
 * weight_decay, 1-beta1^step, 1-beta2^step, max_norm (0 = no clip)}; the betas are doubles so that the bias
 * corrections are formed from the un-rounded values, as Python does.  No host buffer is read when the kernel runs,
 * so the host may enqueue any number of steps ahead (hipGraph replays included). */
int dpot_adam_stage(float* hyper, int64_t* step, float lr, double beta1, double beta2, float eps,
                    float weight_decay, float max_norm, int advance, dpot_stream_t stream);

/* xx_out = xx + noise_scale * ||xx||_2(over X,Y,T per (b,c)) * eps   (train_temporal.py:205)
 * xx, eps: [B, S, C]; norms: B*C*(1 + dpot_noise_chunks(S, C)) floats - [B, C] norms followed by the chunk partials */
int dpot_noise_chunks(int S, int C);
int dpot_noise_inject(const float* xx, const float* eps, float* out, float* norms, float noise_scale, int B,
                      int S, int C, dpot_stream_t stream);
/* the same with eps ~ N(0,1) drawn inside the kernel (Philox4x32-10 + Box-Muller; no eps tensor in HBM).
 * rng_state (DEVICE, 2 x uint64) = {seed, offset}; every call advances offset by one on the device, so a captured
 * hipGraph draws fresh noise on every replay.  Needs S*C % 4 == 0 and 16-byte aligned xx / out. */
int dpot_noise_inject_rng(const float* xx, float* out, float* norms, uint64_t* rng_state, float noise_scale, int B,
                          int S, int C, dpot_stream_t stream);

/* The AFNO mixer's 2-layer block-diagonal complex MLP (models/dpot.py:72-94) as ONE launch (csrc/afno_mlp.hip):
 *   mode 0 (forward):        pre = X Wa + ba;  mid = act(pre);          Y = mid Wb + bb
 *   mode 1 (backward data):  mid = (X Wa) * act'(aux);                  Y = mid Wb         (ba = bb = NULL)
 *                            pre (optional) = act(aux): the activated layer-1 output of the FORWARD, re-derived here
 *                            for the layer-2 weight gradient, so that the forward need not store it
 * X [M, ldx], outputs / aux [M, ldo]: block k owns columns k*N..(k+1)*N, N = 2*bs = [re | im].  Wa / Wb: the real
 * N x N matrices of the packed complex weights in FRAGMENT-BLOCK-MAJOR order, as written by dpot_afno_block_weights
 * from dpot_afno_pack's Wbig: its `fwd` output for mode 0 (multiply by Wbig), its `bwd` output for mode 1 (multiply
 * by Wbig^T).  pre / mid may be NULL (inference).  Supported when 2*bs is 64, 128, 192 or 256
 * (dpot_afno_mlp2_supported); everything 16-byte aligned.
 * layout = 1 (bs in {64, 96, 128}, dpot_afno_mlp3_supported): the THREE-product form of the complex multiplication
 *   P1 = Sr Wr, P2 = Si Wi, P3 = (Sr+Si)(Wr+Wi) -> re = P1 - P2, im = P3 - P1 - P2   (25 % fewer MFMAs);
 * Wa / Wb are then the (Wr, Wi) fragment packs written by dpot_afno_pack_all(layout = 1): `fwd` for mode 0, `bwd`
 * (= Wr^T, -Wi^T) for mode 1.  Same outputs up to fp32 rounding (tests: same tolerance as layout 0). */
int dpot_afno_mlp2_supported(int nb, int bs);
int dpot_afno_mlp3_supported(int nb, int bs);
int dpot_afno_mlp2(const float* X, const float* Wa, const float* ba, const float* Wb, const float* bb,
                   const float* aux, float* pre, float* mid, float* Y, int M, int nb, int bs, int ldx, int ldo,
                   int act, int mode, int layout, dpot_stream_t stream);
/* The same two-layer mixer MLP on the BF16 matrix cores at fp32 accuracy (csrc/afno_mlp6.hip, "bf16x6": every operand split
 * into three bf16 planes, the six plane products of weight >= 2^-16 accumulated in fp32 on v_mfma_f32_32x32x16_bf16; results
 * agree with dpot_afno_mlp2 to fp32 rounding - same tests, same tolerance).  Arguments, modes and outputs exactly as
 * dpot_afno_mlp2; Wa6 / Wb6 are the packs dpot_afno_pack6 writes ([nb] matrices of dpot_afno_pack6_elems(1, bs) bf16 each):
 * mode 0: (fwd6 of the first-layer weight, fwd6 of the second-layer weight); mode 1: (bwd6 of the SECOND-layer weight, bwd6 of
 * the FIRST-layer weight).  bs in {96, 128} (dpot_afno_mlp6_supported).  This is what DPOTNet.gemm_precision 'auto' / 'bf16x6'
 * selects for the mixer (models/dpot.py:72-94). */
int dpot_afno_mlp6_supported(int nb, int bs);
int64_t dpot_afno_pack6_elems(int nb, int bs);                    /* bf16 elements of one pack of nb matrices */
/* wbig [nitems][nb][N][N] (W[k][n], what dpot_afno_pack_all leaves; items alternate first-layer weight, second-layer weight)
 * -> fwd6 / bwd6 [nitems][nb][...] (either may be NULL): the forward operand (W) and the backward-data operand (W^T) of each
 * matrix in the fragment order of the layer it is used in, three bf16 planes */
int dpot_afno_pack6(const float* wbig, void* fwd6, void* bwd6, int nitems, int nb, int bs, dpot_stream_t stream);
int dpot_afno_mlp6(const float* X, const void* Wa6, const float* ba, const void* Wb6, const float* bb, const float* aux,
                   float* pre, float* mid, float* Y, int M, int nb, int bs, int ldx, int ldo, int act, int mode,
                   dpot_stream_t stream);
/* wbig [nmat][N][N] (row-major W[k][n]) -> [nmat][N/16][N/16][256] blocks of (16 n x 16 k), chunk l of a block =
 * (n = l&15, k = 4*(l>>4)..+3): fwd holds W (for X W), bwd holds W^T (for X W^T).  Either output may be NULL. */
int dpot_afno_block_weights(const float* wbig, float* fwd, float* bwd, int nmat, int N, dpot_stream_t stream);

/* one AFNO layer to pack (DEVICE table entry, 48 bytes): parameters w [2,nb,bs,bs], b [2,nb,bs] -> wbig [nb,N,N]
 * (W[k][n], may be NULL), bbig [nb,N], fwd / bwd fragment-block-major W / W^T for dpot_afno_mlp2 (may be NULL) */
typedef struct dpot_afno_pack_job {
  const float* w;
  const float* b;
  float* wbig;
  float* bbig;
  float* fwd;
  float* bwd;
} dpot_afno_pack_job;
/* dpot_afno_pack + dpot_afno_block_weights for every layer of a model in ONE launch; the table lives in device memory */
int dpot_afno_pack_all(const dpot_afno_pack_job* jobs_dev, int njobs, int nb, int bs, int layout, dpot_stream_t stream);

/* backward of the noise injection for AR steps whose input depends on earlier predictions:
 * dx = g + noise_scale * xx / norms[b,c] * sum_(X,Y,T)(g * eps).  eps: the tensor the forward used, or NULL with
 * rng_state = a copy of the generator state {seed, offset} the forward drew from.  norms: [B, C] written by the
 * forward.  part: B * C * dpot_noise_chunks(S, C) floats of scratch. */
int dpot_noise_inject_bwd(const float* xx, const float* eps, const uint64_t* rng_state, const float* g,
                          const float* norms, float* dx, float* part, float noise_scale, int B, int S, int C,
                          dpot_stream_t stream);

/* auto-regressive window slide (train_temporal.py:219  xx = cat(xx[..., T_bundle:, :], im)):
 * xx [rows, T, C], im [rows, Tb, C] -> out [rows, T, C] (rows = B*X*Y); bwd: dxx (first Tb steps zero) and dim
 * (either may be NULL) from dout. */
int dpot_window_slide(const float* xx, const float* im, float* out, int64_t rows, int T, int Tb, int C,
                      dpot_stream_t stream);
int dpot_window_slide_bwd(const float* dout, float* dxx, float* dim, int64_t rows, int T, int Tb, int C,
                          dpot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * panel GEMM with a pre-packed static weight (csrc/gemm_panel.hip): forward and data gradient of the channel-MLP
 * ------------------------------------------------------------------------------------------------ */
/* one weight to pack: the logical matrix Wt [rows, K] (row n, k) = trans ? src[k*ld + n] : src[n*ld + k]
 * -> dst [rows/16][K/16][256] fragment-block-major (chunk l of a block = (row l&15, k 4*(l>>4)..+3)).  32 bytes. */
typedef struct dpot_pack_job {
  const float* src;
  float* dst;
  int32_t rows, K, ld, trans;
} dpot_pack_job;
/* packs every weight of the DEVICE table jobs_dev[0..njobs) in one launch; max_elems = the largest rows*K */
int dpot_panel_pack_weights(const dpot_pack_job* jobs_dev, int njobs, int max_elems, dpot_stream_t stream);
/* C[M,N] = epilogue(A[M,K] @ Wt^T) with Wt packed by dpot_panel_pack_weights (rows = N).  epi_mode / act as for
 * dpot_gemm_f32: LINEAR, ACT (pre-activation optionally saved to `pre`), DACT (times act'(aux)); bias [N], res [M,ldres]
 * and pre may be NULL.  Needs K % 32 == 0 and N % 64 == 0 (dpot_gemm_panel_supported), 16-byte aligned operands. */
int dpot_gemm_panel_supported(int M, int N, int K);
int dpot_gemm_panel(const float* A, int lda, const float* Wpacked, const float* bias, const float* aux, int ldaux,
                    const float* res, int ldres, float* pre, int ldpre, float* C, int ldc, int M, int N, int K,
                    int act, int epi_mode, dpot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * bf16 panel GEMM (csrc/gemm_bf16p.hip) on v_mfma_f32_32x32x16_bf16, fp32 accumulation, both operands PRE-PACKED in
 * fragment-block-major order: [ceil(rows/32)][K/16][planes][64 chunks][8 bf16], chunk l = (row l&31, k 8*(l>>5)..+7);
 * rows past the matrix are zero.
 *   planes = 1: operands rounded to bf16 - REDUCED precision, opt-in (BASELINE configs[2] "bf16 channel-MLP on MFMA")
 *   planes = 3: x = x1 + x2 + x3 (three bf16 planes), six plane products accumulated ("bf16x6"): fp32-ACCURATE
 *               (~2^-24 per product, like a native fp32 FMA chain) at up to 2.67x the fp32 matrix-core roof
 * ------------------------------------------------------------------------------------------------ */
int64_t dpot_bf16_packed_elems(int rows, int K, int planes);      /* bf16 elements of a packed [rows, K] operand */
/* activations: src fp32 [rows, K] row-major (ld) - or, trans != 0, its transpose stored [K, rows] (weight gradients:
 * rows = features, k = tokens) - -> dst packed bf16 (one HBM pass); K % 16 == 0 */
int dpot_bf16_pack_rows(const float* src, int ld, int rows, int K, int trans, int planes, void* dst,
                        dpot_stream_t stream);
/* one pass that produces BOTH 1-plane packed forms of an activation (row form = GEMM A operand, transposed form =
 * weight-gradient operand; either may be NULL) and, if colsum_part != NULL, partial column sums [rows/64, K] (the bias
 * gradient = their sum over the first index, fixed order: dpot_colsum).  rows % 64 == 0, K % 256 == 0. */
int dpot_bf16_pack_both_supported(int rows, int K);
int dpot_bf16_pack_both(const float* src, int ld, int rows, int K, void* dst_rows, void* dst_trans, float* colsum_part,
                        dpot_stream_t stream);
/* the same pass over GroupNorm(src): the packs of (src - mean) * rstd * gamma + beta, statistics [rows / rows_per_sample, G]
 * given (norm2 of Block.forward, models/dpot.py:173, folded into the pack of the channel MLP's input: the normalised tensor
 * is never written in fp32).  rows_per_sample % 64 == 0, (K / G) % 4 == 0. */
int dpot_bf16_pack_both_norm(const float* src, int ld, int rows, int K, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, int rows_per_sample, int G, void* dst_rows,
                             void* dst_trans, dpot_stream_t stream);
/* static weights: a DEVICE table of dpot_pack_job entries whose dst is the packed bf16 buffer, all weights in one launch */
int dpot_bf16_pack_jobs(const dpot_pack_job* jobs_dev, int njobs, int max_elems, int planes, dpot_stream_t stream);
/* Adam that ALSO emits the 1-plane bf16 packs of the channel-MLP weights (utils/optimizer.py:26-52 + the pack pass that
 * would otherwise re-read every weight it has just written).  p / g / m / v / hyper / sumsq / grad_scale as for
 * dpot_adam_step (include: the clip of train_temporal.py:228).  jobs_dev: DEVICE table, one entry per weight [R, K] stored
 * row-major at element offset `off` of the flat buffers (R % 64 == 0, K % 256 == 0: dpot_adam_pack_supported); dst_rows =
 * packed [R rows, K] (what dpot_bf16_pack_jobs writes for (src, R, K, K, trans = 0)), dst_trans = packed [K rows, R] (its
 * (src, K, R, K, trans = 1) form); either may be NULL.  tile0 = index of the job's first 64 x 256 tile in the launch;
 * tile_job_dev[ntiles] maps a tile to its job.  ranges_dev[nranges]: the remaining stretches of the flat buffers (each
 * <= max_range_len elements; plain Adam).  Two launches, no allocation, capturable. */
typedef struct dpot_adam_pack_job {
  int64_t off;
  void* dst_rows;
  void* dst_trans;
  int32_t R, K, tile0, pad_;
} dpot_adam_pack_job;
typedef struct dpot_adam_range {
  int64_t start, len;
} dpot_adam_range;
int dpot_adam_pack_supported(int rows, int K);
int dpot_adam_step_packs(float* p, const float* g, float* m, float* v, const float* hyper, const float* sumsq,
                         float grad_scale, const dpot_adam_pack_job* jobs_dev, const int32_t* tile_job_dev, int ntiles,
                         const dpot_adam_range* ranges_dev, int nranges, int max_range_len, dpot_stream_t stream);
/* C[M,N] (fp32) = epilogue(A @ Wt^T), A = packed [M, K], Wt = packed [N, K] (same `planes`); epilogue as
 * dpot_gemm_panel.  Needs N % 256 == 0 and K % 32 == 0 (dpot_gemm_bf16p_supported). */
int dpot_gemm_bf16p_supported(int M, int N, int K);
/* which kernel dpot_gemm_bf16p selects for a shape (for reports: bench.py names the kernel it times): 0 = LDS-DMA kernel
 * (8 compute + 4 loader waves), 1 = two-workgroup kernel, 2 = B-direct with eight 128 x 32 waves, 3 = B-direct with four
 * 128 x 64 waves (two workgroups per CU), 4 = bf16x6; + 8 when it runs on 128 x 192 tiles; -1: unsupported shape */
int dpot_gemm_bf16p_kernel_kind(int M, int N, int K, int splitk, int planes, int packed_outputs);
int dpot_gemm_bf16p(const void* Apacked, const void* Wpacked, const float* bias, const float* aux, int ldaux,
                    const float* res, int ldres, float* pre, int ldpre, float* C, int ldc, int M, int N, int K, int act,
                    int epi_mode, int planes, int splitk, float* workspace, void* out_rows, void* out_trans,
                    float* colsum_part, void* dact_out, const void* dact_in, dpot_stream_t stream);
/* two independent products of that kind (plain bf16 operands, common K, linear epilogue, no split-K) in ONE launch:
 * the fc1 / fc2 weight gradients of a block, which alone have too few tiles for 256 CUs and would each go through
 * split-K partials + a reduce launch.  _wanted: 1 when each alone would be split and together they fill the chip. */
int dpot_gemm_bf16p_pair_wanted(int M0, int N0, int M1, int N1, int K);
/* common split-K factor of the pair launch: 0 = do not pair, 1 = no split, s > 1 = s splits (workspace of
 * s * (M0*N0 + M1*N1) floats; the partial sums are reduced in a fixed order by two further launches) */
int dpot_gemm_bf16p_pair_splitk(int M0, int N0, int M1, int N1, int K);
int dpot_gemm_bf16p_pair(const void* A0, const void* W0, float* C0, int ldc0, int M0, int N0, const void* A1,
                         const void* W1, float* C1, int ldc1, int M1, int N1, int K, int splitk, float* workspace,
                         dpot_stream_t stream);
/* out_rows / out_trans / colsum_part (all optional, planes == 1, splitk <= 1, M % 32 == 0): the epilogue also emits the
 * 1-plane packs of the FINAL output (row form [M, N]; transposed form = rows N, k M) and partial column sums
 * [M/32, N] - the next GEMMs of a chain then need no pack pass over this output; C may be NULL in that case.
 * dact_out (EPI_ACT launches): act'(pre-activation) as bf16, M * N elements in the kernel's FRAGMENT order
 * ([M/32][N/32][2][64 lanes][8]: the accumulator layout of a 32x32 tile, opaque to the caller) - what the backward
 * of Block.mlp (models/dpot.py:157-161) multiplies by; it replaces the fp32 `pre` save at half the bytes.
 * dact_in (EPI_DACT launches): that buffer (same M, N), used instead of act'(aux).  Same restrictions as the packed
 * outputs; not together with `pre` or `res`. */
/* split-K factor the library recommends for a shape (weight gradients: few output tiles, K = tokens); splitk > 1 needs
 * a workspace of splitk*M*N floats, summed in a fixed order by a second launch (deterministic) */
int dpot_gemm_bf16p_splitk(int M, int N, int K);

/* ------------------------------------------------------------------------------------------------
 * input pipeline, device side (utils/griddataset.py:88-101 pad_data, :125-174 __getitem__)
 * ------------------------------------------------------------------------------------------------ */
/* one raw trajectory of a dataset, already in device memory: data [H, W, T, C] fp32, window start t0 (32 bytes) */
typedef struct dpot_sample_desc {
  const float* data;
  int32_t H, W, T, C;
  int32_t t0;
  int32_t reserved;
} dpot_sample_desc;
/* For every sample b: bilinear resize of the frames t0 .. t0+t_in+t_ar-1 to res x res (F.interpolate(mode='bilinear'),
 * align_corners=False semantics), channels C..n_channels-1 filled with ones, written as
 * xx[b] = [res, res, t_in, n_channels] and yy[b] = [res, res, t_ar, n_channels] (yy may be NULL when t_ar == 0).
 * `samples_dev` is a DEVICE array of nsamples descriptors (upload it with the raw samples, one H2D copy per batch);
 * the caller validates it (C <= n_channels, t0 + t_in + t_ar <= T) - a malformed entry is skipped by the kernel.
 * down_h, down_w >= 1: the strided sub-sampling x[::down_h, ::down_w] the reference applies to the resized fields of
 * some datasets (utils/griddataset.py:170-172): xx[b] / yy[b] then have ceil(res / down) points per axis.
 * Test-mode windows (griddataset.py:159-163) are the same call with t0 = 0 and t_ar = min(t_test, T - t_in). */
int dpot_resize_pad_window(const dpot_sample_desc* samples_dev, int nsamples, float* xx, float* yy, int res, int t_in,
                           int t_ar, int n_channels, int down_h, int down_w, dpot_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DPOT_HIP_H */
