// common.h - shared device helpers for libdpot_hip.so (gfx950 only)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/dpot_hip.h"

namespace dpot {

// ---- error reporting -------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// DPOT_TUNE="key=val,key=val": the ONE environment variable behind every fallback-path selector of the library (read once;
// keys documented in DESIGN.md section 0 / dpot_amd/ops.py TUNE_KEYS).  Returns dflt when the key is absent.
int tune(const char* key, int dflt);

#define DPOT_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::dpot::set_error(__VA_ARGS__);      \
      return DPOT_EINVAL;                  \
    }                                      \
  } while (0)

static inline hipStream_t as_stream(dpot_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- activations (models/dpot.py:19) ---------------------------------------------------------------
// Exact-erf GELU is the DPOT default and sits in every GEMM epilogue and in the fused tail; libm's erff costs ~150
// VALU instructions per element (two divergent branches), which made GELU ~8% of the whole training step.
// The normal tail is evaluated branch-free, without a division, as  Phi(-a) = 2^-R(a),  a = min(|x|, 6.5):
// R(a) = -log2 Phi(-a) is smooth (1 at a = 0, ~0.72 a^2 for large a) and a degree-8 polynomial (weighted minimax fit,
// scripts/fit_gelu_tail.py) reproduces Phi(-a) to 1.4e-8; in fp32 against float64 over [-12, 12]:
// |Phi err| <= 8.3e-8, |gelu err| <= 2.5e-7 (half an ulp at |x| = 12), |gelu' err| <= 1.5e-7 - three orders below the
// 1e-4 parity tolerance.  With the tail in hand
//     gelu(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|),      Phi(x) = x >= 0 ? 1 - Phi(-|x|) : Phi(-|x|)
// (no cancellation in the left tail; beyond |x| = 6.5 the clamped tail, 4e-11, is below fp32 resolution of the result).
// Cost: 11 VALU instructions + one v_exp_f32 (quarter rate) = 15 issue slots - the previous Abramowitz-Stegun 7.1.26
// form needed a reciprocal AND an exponential plus a compare / select and was 3x less accurate.  This matters: the
// per-pixel tail kernels (csrc/tail.hip) are bound by VALU ISSUE, 33.5 M GELUs per launch at DPOT-Tiny B=32.
// The clamp and the max use gfx950's v_minimum3_f32 / v_maximum3_f32 (IEEE-754-2019 minimum / maximum): unlike
// v_min / v_max they PROPAGATE NaN, so a diverged run still surfaces as NaN loss (and they take the |x| modifier).
// (Packed v_pk_fma_f32 would halve the polynomial's slots, but packed fp32 is an anti-lever beside MFMAs -
// MI355X_MICROARCH "price of one filler" - and the compiler scalarises most of it there anyway; measured: no gain.)
__device__ __forceinline__ float normal_tail(float ax) {   // Phi(-ax), 0 <= ax <= 6.5
  float r = fmaf(ax, 2.275872930e-06f, -3.296802970e-05f);
  r = fmaf(r, ax, 1.572788460e-04f);
  r = fmaf(r, ax, 2.024787827e-04f);
  r = fmaf(r, ax, -7.142781746e-03f);
  r = fmaf(r, ax, 5.254643410e-02f);
  r = fmaf(r, ax, 4.591930509e-01f);
  r = fmaf(r, ax, 1.151106954e+00f);
  r = fmaf(r, ax, 9.999999404e-01f);
  return __builtin_amdgcn_exp2f(-r);
}
__device__ __forceinline__ float gelu_clamp_abs(float x) { return __builtin_elementwise_minimum(__builtin_fabsf(x), 6.5f); }
__device__ __forceinline__ float gelu_fwd(float x) {
  const float ax = gelu_clamp_abs(x);
  return fmaf(-ax, normal_tail(ax), __builtin_elementwise_maximum(x, 0.0f));
}
// cdf = Phi(x), gauss = exp(-x^2 / 2) (the Gaussian of the derivative)
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& gauss) {
  const float e = normal_tail(gelu_clamp_abs(x));
  gauss = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
  cdf = x >= 0.f ? 1.0f - e : e;
}
__device__ __forceinline__ float gelu_bwd(float x) {
  float cdf, g;
  gelu_parts(x, cdf, g);
  return fmaf(x * 0.39894228040143267794f, g, cdf);
}
// value and derivative together (one tail evaluation)
__device__ __forceinline__ void gelu_val_der(float x, float& val, float& der) {
  const float ax = gelu_clamp_abs(x);
  const float e = normal_tail(ax);
  const float phi = __builtin_amdgcn_exp2f(fmaf(x * x, -0.72134752044448170368f, -1.32574806473615975284f));
  val = fmaf(-ax, e, __builtin_elementwise_maximum(x, 0.0f));
  der = fmaf(x, phi, x >= 0.f ? 1.0f - e : e);
}

// (Round 4 also measured a two-elements-at-a-time form on the packed fp32 pipe - v_pk_fma_f32 for the polynomial and the FMAs
// around it, bit-identical results - in the bf16 GEMM's packed-output epilogue, where no MFMA competes: no gain,
// profiles/r04_bf16p_gelu_packed_rejected.txt.  That epilogue is not VALU-bound.)
__device__ __forceinline__ float act_fwd(int act, float x) {
  switch (act) {
    case DPOT_ACT_GELU: return gelu_fwd(x);
    case DPOT_ACT_TANH: return tanhf(x);
    case DPOT_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
    case DPOT_ACT_RELU: return x > 0.f ? x : 0.f;
    case DPOT_ACT_LEAKY_RELU: return x > 0.f ? x : 0.1f * x;
    case DPOT_ACT_SOFTPLUS: return x > 20.f ? x : log1pf(expf(x));
    case DPOT_ACT_ELU: return x > 0.f ? x : expm1f(x);
    case DPOT_ACT_SILU: return x / (1.0f + expf(-x));
    default: return x;
  }
}

// derivative of the activation as a function of the PRE-activation x
__device__ __forceinline__ float act_bwd(int act, float x) {
  switch (act) {
    case DPOT_ACT_GELU: return gelu_bwd(x);
    case DPOT_ACT_TANH: {
      const float t = tanhf(x);
      return 1.0f - t * t;
    }
    case DPOT_ACT_SIGMOID: {
      const float s = 1.0f / (1.0f + expf(-x));
      return s * (1.0f - s);
    }
    case DPOT_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case DPOT_ACT_LEAKY_RELU: return x > 0.f ? 1.f : 0.1f;
    case DPOT_ACT_SOFTPLUS: return x > 20.f ? 1.f : 1.0f / (1.0f + expf(-x));
    case DPOT_ACT_ELU: return x > 0.f ? 1.f : expf(x);
    case DPOT_ACT_SILU: {
      const float s = 1.0f / (1.0f + expf(-x));
      return s * (1.0f + x * (1.0f - s));
    }
    default: return 1.f;
  }
}

// ---- reductions (wave = 64 lanes on gfx950) ---------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `sh` needs >= 16 floats; result broadcast to all
__device__ __forceinline__ float block_sum(float v, float* sh) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();  // protect sh reuse across consecutive calls
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];  // fixed order -> deterministic
  return r;
}
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

// two block-wide sums in one pass (one pair of barriers); `sh` needs >= 32 doubles
__device__ __forceinline__ void block_sum2_d(double& a, double& b, double* sh) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  __syncthreads();
  if (lane == 0) {
    sh[wid] = a;
    sh[16 + wid] = b;
  }
  __syncthreads();
  double ra = 0.0, rb = 0.0;
  for (int i = 0; i < nw; ++i) {
    ra += sh[i];
    rb += sh[16 + i];
  }
  a = ra;
  b = rb;
}

// ---- Adam update shared by adam_kernel (loss_opt.hip) and adam_pack_kernel (gemm_bf16p.hip) ---------------------------
// utils/optimizer.py:26-52: L2 weight decay folded into the gradient, bias-corrected step; the clip coefficient of
// train_temporal.py:228 multiplies the gradient.  Every product / sum is written as the explicit fmaf / mul the compiler
// would contract it to, so the two kernels produce bit-identical parameters whatever their surrounding code.
struct AdamCoef {
  float b1, b2, omb1, omb2, eps, wd, step, sb2, gs;
};
__device__ __forceinline__ AdamCoef adam_coef(const float* __restrict__ hyper, const float* __restrict__ sumsq,
                                              float grad_scale) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
  const float bc1 = hyper[5], bc2 = hyper[6], max_norm = hyper[7];
  float gs = grad_scale;
  if (sumsq) {
    const float total_norm = sqrtf(sumsq[0]) * grad_scale;
    float coef = max_norm / (total_norm + 1e-6f);
    if (coef > 1.f) coef = 1.f;
    gs *= coef;
  }
  return AdamCoef{b1, b2, 1.f - b1, 1.f - b2, eps, wd, lr / bc1, sqrtf(bc2), gs};
}
__device__ __forceinline__ void adam_update(const AdamCoef& c, float& pv, float gv, float& mv, float& vv) {
  gv *= c.gs;
  if (c.wd != 0.f) gv = fmaf(c.wd, pv, gv);
  mv = fmaf(c.b1, mv, c.omb1 * gv);
  vv = fmaf(c.b2, vv, (c.omb2 * gv) * gv);
  const float denom = sqrtf(vv) / c.sb2 + c.eps;
  pv = pv - c.step * (mv / denom);
}

}  // namespace dpot
