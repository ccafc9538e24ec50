// norm.hip - GroupNorm(G, E) forward/backward on channels-last fields x[B, T, E]  (models/dpot.py:142,152).
//
// One 1024-thread workgroup per (sample, group): the slab is T x (E/G) floats (Tiny: 256 x 64 = 64 KiB); it is read
// from HBM once, the later sweeps hit L2.  HBM-bound: algorithmic traffic = read x + write y (fwd).
// Thread layout: tj = lane over channel QUADS inside the group (16-byte accesses, contiguous in memory),
// tt = lane over tokens.  VW = 4 when E/G is a multiple of 4 (and pointers are 16-B aligned), else 1.
#include "common.h"

namespace dpot {

constexpr int GN_THREADS = 1024;

struct GnLayout {
  int TJ, TT;  // TJ * TT == GN_THREADS
};
__host__ __device__ inline GnLayout gn_layout(int cgv) {  // cgv = channel vectors per group
  int tj = 1;
  while (tj * 2 <= cgv && tj * 2 <= 64) tj *= 2;
  GnLayout l;
  l.TJ = tj;
  l.TT = GN_THREADS / tj;
  return l;
}

template <int VW>
struct VecT;
template <>
struct VecT<4> {
  typedef float4 type;
};
template <>
struct VecT<1> {
  typedef float type;
};
template <int VW>
__device__ __forceinline__ void vload(const float* p, float (&v)[VW]) {
  if constexpr (VW == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = p[0];
  }
}
template <int VW>
__device__ __forceinline__ void vstore(float* p, const float (&v)[VW]) {
  if constexpr (VW == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    p[0] = v[0];
  }
}

template <int VW>
__global__ __launch_bounds__(GN_THREADS) void groupnorm_fwd_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float* __restrict__ y,
                                                                   float* __restrict__ mean, float* __restrict__ rstd,
                                                                   int T, int E, int G, float eps) {
  __shared__ double shd[16];
  const int g = blockIdx.x, b = blockIdx.y;
  const int cg = E / G, cgv = cg / VW;
  const GnLayout L = gn_layout(cgv);
  const int tj = threadIdx.x % L.TJ, tt = threadIdx.x / L.TJ;
  const float* xs = x + (long long)b * T * E + g * cg;
  float* ys = y + (long long)b * T * E + g * cg;
  const double n = (double)T * cg;

  float s = 0.f;
  for (int t = tt; t < T; t += L.TT)
    for (int j = tj; j < cgv; j += L.TJ) {
      float v[VW];
      vload<VW>(xs + (long long)t * E + j * VW, v);
#pragma unroll
      for (int k = 0; k < VW; ++k) s += v[k];
    }
  const float mu = (float)(block_sum_d((double)s, shd) / n);
  float q = 0.f;
  for (int t = tt; t < T; t += L.TT)
    for (int j = tj; j < cgv; j += L.TJ) {
      float v[VW];
      vload<VW>(xs + (long long)t * E + j * VW, v);
#pragma unroll
      for (int k = 0; k < VW; ++k) {
        const float d = v[k] - mu;
        q = fmaf(d, d, q);
      }
    }
  const float var = (float)(block_sum_d((double)q, shd) / n);
  const float rs = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean[b * G + g] = mu;
    rstd[b * G + g] = rs;
  }
  for (int j = tj; j < cgv; j += L.TJ) {
    float ga[VW], be[VW];
    vload<VW>(gamma + g * cg + j * VW, ga);
    vload<VW>(beta + g * cg + j * VW, be);
#pragma unroll
    for (int k = 0; k < VW; ++k) ga[k] *= rs;
    for (int t = tt; t < T; t += L.TT) {
      const long long o = (long long)t * E + j * VW;
      float v[VW];
      vload<VW>(xs + o, v);
#pragma unroll
      for (int k = 0; k < VW; ++k) v[k] = fmaf(v[k] - mu, ga[k], be[k]);
      vstore<VW>(ys + o, v);
    }
  }
}

// part[0,b,c] = sum_t dy*xhat (dgamma partial), part[1,b,c] = sum_t dy (dbeta partial)
template <int VW>
__global__ __launch_bounds__(GN_THREADS) void groupnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ add, float* __restrict__ dx,
                                                                   float* __restrict__ part, int B, int T, int E, int G) {
  __shared__ float red[2][GN_THREADS][VW];
  __shared__ double shd[16];
  const int g = blockIdx.x, b = blockIdx.y;
  const int cg = E / G, cgv = cg / VW;
  const GnLayout L = gn_layout(cgv);
  const int tj = threadIdx.x % L.TJ, tt = threadIdx.x / L.TJ;
  const long long off = (long long)b * T * E + g * cg;
  const float* xs = x + off;
  const float* dys = dy + off;
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  const double n = (double)T * cg;

  double s1 = 0.0, s2 = 0.0;  // sum gamma*dy, sum gamma*dy*xhat (only thread rows tt == 0 contribute)
  for (int j0 = 0; j0 < cgv; j0 += L.TJ) {
    const int j = j0 + tj;
    float a_dy[VW], a_dyx[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) a_dy[k] = a_dyx[k] = 0.f;
    if (j < cgv) {
      for (int t = tt; t < T; t += L.TT) {
        const long long o = (long long)t * E + j * VW;
        float d[VW], v[VW];
        vload<VW>(dys + o, d);
        vload<VW>(xs + o, v);
#pragma unroll
        for (int k = 0; k < VW; ++k) {
          a_dy[k] += d[k];
          a_dyx[k] = fmaf(d[k], (v[k] - mu) * rs, a_dyx[k]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      red[0][threadIdx.x][k] = a_dyx[k];
      red[1][threadIdx.x][k] = a_dy[k];
    }
    __syncthreads();
    // fixed-order reduction over the token lanes, done by TT/.. threads: lane (tj, k) x 1
    if (tt < VW && j < cgv) {
      const int k = tt;  // one thread per (channel vector, component)
      float sg = 0.f, sb = 0.f;
      for (int r = 0; r < L.TT; ++r) {
        sg += red[0][r * L.TJ + tj][k];
        sb += red[1][r * L.TJ + tj][k];
      }
      const int c = g * cg + j * VW + k;
      part[((long long)0 * B + b) * E + c] = sg;
      part[((long long)1 * B + b) * E + c] = sb;
      const float ga = gamma[c];
      s1 += (double)ga * sb;
      s2 += (double)ga * sg;
    }
  }
  const float m1 = (float)(block_sum_d(s1, shd) / n);
  const float m2 = (float)(block_sum_d(s2, shd) / n);
  float* dxs = dx + off;
  const float* adds = add ? add + off : nullptr;
  for (int j = tj; j < cgv; j += L.TJ) {
    float ga[VW];
    vload<VW>(gamma + g * cg + j * VW, ga);
    for (int t = tt; t < T; t += L.TT) {
      const long long o = (long long)t * E + j * VW;
      float d[VW], v[VW], a[VW];
      vload<VW>(dys + o, d);
      vload<VW>(xs + o, v);
      if (adds) vload<VW>(adds + o, a);
#pragma unroll
      for (int k = 0; k < VW; ++k) {
        const float xh = (v[k] - mu) * rs;
        float r = rs * (ga[k] * d[k] - m1 - xh * m2);
        if (adds) r += a[k];
        v[k] = r;
      }
      vstore<VW>(dxs + o, v);
    }
  }
}

__global__ void groupnorm_param_grad_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                            float* __restrict__ dbeta, int B, int E) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= E) return;
  float sg = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) {
    sg += part[((long long)0 * B + b) * E + c];
    sb += part[((long long)1 * B + b) * E + c];
  }
  dgamma[c] = sg;
  dbeta[c] = sb;
}

}  // namespace dpot

using namespace dpot;

extern "C" int dpot_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                  float* rstd, int B, int T, int E, int G, float eps, dpot_stream_t stream) {
  DPOT_REQUIRE(x && gamma && beta && y && mean && rstd, "groupnorm_fwd: null pointer");
  DPOT_REQUIRE(B > 0 && T > 0 && E > 0 && G > 0 && E % G == 0 && B <= 65535, "groupnorm_fwd: bad shape");
  const bool vec = ((E / G) % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta);
  if (vec)
    hipLaunchKernelGGL(groupnorm_fwd_kernel<4>, dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), x, gamma, beta, y,
                       mean, rstd, T, E, G, eps);
  else
    hipLaunchKernelGGL(groupnorm_fwd_kernel<1>, dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), x, gamma, beta, y,
                       mean, rstd, T, E, G, eps);
  return check_launch("groupnorm_fwd_kernel");
}

extern "C" int dpot_groupnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                                  const float* gamma, const float* add, float* dx, float* dgamma, float* dbeta,
                                  float* part, int B, int T, int E, int G, dpot_stream_t stream) {
  DPOT_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && part, "groupnorm_bwd: null pointer");
  DPOT_REQUIRE(B > 0 && T > 0 && E > 0 && G > 0 && E % G == 0 && B <= 65535, "groupnorm_bwd: bad shape");
  const bool vec = ((E / G) % 4 == 0) && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(gamma) &&
                   (add == nullptr || aligned16(add));
  if (vec)
    hipLaunchKernelGGL(groupnorm_bwd_kernel<4>, dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), dy, x, mean, rstd,
                       gamma, add, dx, part, B, T, E, G);
  else
    hipLaunchKernelGGL(groupnorm_bwd_kernel<1>, dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), dy, x, mean, rstd,
                       gamma, add, dx, part, B, T, E, G);
  int rc = check_launch("groupnorm_bwd_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(groupnorm_param_grad_kernel, dim3(cdiv(E, 256)), dim3(256), 0, as_stream(stream),
                     (const float*)part, dgamma, dbeta, B, E);
  return check_launch("groupnorm_param_grad_kernel");
}
