// common.h - shared device helpers for libdpot_hip.so (gfx950 only)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/dpot_hip.h"

namespace dpot {

// ---- error reporting -------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

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
// Phi(x) = 0.5 (1 + erf(x/sqrt2)) is evaluated branch-free from Abramowitz-Stegun 7.1.26,
//   1 - erf(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2),  t = 1/(1 + p z),  |err| <= 1.5e-7,
// on |x| with the reflection Phi(-x) = 1 - Phi(x) (no cancellation in the left tail); measured in fp32 against
// float64 erf over [-12, 12]: |Phi err| <= 3.0e-7, |gelu err| <= 6.5e-7 (1.6e-7 relative), |gelu' err| <= 3.3e-7 -
// two orders below the 1e-4 parity tolerance.  exp(-z^2) = exp(-x^2/2) is also the Gaussian of the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& gauss) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  // v_rcp_f32 / v_exp_f32 directly (1 ulp each): __frcp_rn expands to the correctly-rounded division sequence
  // (v_div_scale x2, v_div_fmas, v_div_fixup + Newton steps: ~10 of the ~24 instructions of a GELU) and __expf to a
  // range-checked form; their last-bit differences are two orders below the 3e-7 error of the 7.1.26 polynomial itself
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(t, poly, 1.421413741f);
  poly = fmaf(t, poly, -0.284496736f);
  poly = fmaf(t, poly, 0.254829592f);
  poly *= t;
  gauss = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);        // exp(-z^2)
  const float half = 0.5f * poly * gauss;
  cdf = x >= 0.f ? 1.0f - half : half;
}
__device__ __forceinline__ float gelu_fwd(float x) {
  float cdf, g;
  gelu_parts(x, cdf, g);
  return x * cdf;
}
__device__ __forceinline__ float gelu_bwd(float x) {
  float cdf, g;
  gelu_parts(x, cdf, g);
  return fmaf(x * 0.39894228040143267794f, g, cdf);
}

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

}  // namespace dpot
