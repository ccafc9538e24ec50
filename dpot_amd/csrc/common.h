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
__device__ __forceinline__ float act_fwd(int act, float x) {
  switch (act) {
    case DPOT_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
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
    case DPOT_ACT_GELU: {
      const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
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
