// norm.hip - GroupNorm(G, E) forward/backward on channels-last fields x[B, T, E]  (models/dpot.py:142,152).
//
// One workgroup per (sample, group): the slab is T x (E/G) floats (Tiny: 256 x 64 = 64 KiB) - it is read
// from HBM once and the 2nd/3rd sweeps hit L2.  HBM-bound: algorithmic traffic = read x + write y.
// Thread layout: tj = channel lane inside the group (contiguous in memory), tt = token lane.
#include "common.h"

namespace dpot {

struct GnLayout {
  int TJ, TT;  // TJ * TT == 256
};
__host__ __device__ inline GnLayout gn_layout(int cg) {
  int tj = 1;
  while (tj * 2 <= cg && tj * 2 <= 64) tj *= 2;
  GnLayout l;
  l.TJ = tj;
  l.TT = 256 / tj;
  return l;
}

__global__ __launch_bounds__(256) void groupnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int T,
                                                            int E, int G, float eps) {
  __shared__ double shd[16];
  const int g = blockIdx.x, b = blockIdx.y;
  const int cg = E / G;
  const GnLayout L = gn_layout(cg);
  const int tj = threadIdx.x % L.TJ, tt = threadIdx.x / L.TJ;
  const float* xs = x + (long long)b * T * E + g * cg;
  float* ys = y + (long long)b * T * E + g * cg;
  const double n = (double)T * cg;

  float s = 0.f;
  for (int t = tt; t < T; t += L.TT)
    for (int j = tj; j < cg; j += L.TJ) s += xs[(long long)t * E + j];
  const float mu = (float)(block_sum_d((double)s, shd) / n);
  float q = 0.f;
  for (int t = tt; t < T; t += L.TT)
    for (int j = tj; j < cg; j += L.TJ) {
      const float d = xs[(long long)t * E + j] - mu;
      q = fmaf(d, d, q);
    }
  const float var = (float)(block_sum_d((double)q, shd) / n);
  const float rs = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean[b * G + g] = mu;
    rstd[b * G + g] = rs;
  }
  for (int j = tj; j < cg; j += L.TJ) {
    const float ga = gamma[g * cg + j] * rs, be = beta[g * cg + j];
    for (int t = tt; t < T; t += L.TT) {
      const long long o = (long long)t * E + j;
      ys[o] = fmaf(xs[o] - mu, ga, be);
    }
  }
}

// part[0,b,c] = sum_t dy*xhat (dgamma partial), part[1,b,c] = sum_t dy (dbeta partial)
__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ add,
                                                            float* __restrict__ dx, float* __restrict__ part, int B, int T,
                                                            int E, int G) {
  __shared__ float red[2][256];
  __shared__ double shd[16];
  const int g = blockIdx.x, b = blockIdx.y;
  const int cg = E / G;
  const GnLayout L = gn_layout(cg);
  const int tj = threadIdx.x % L.TJ, tt = threadIdx.x / L.TJ;
  const long long off = (long long)b * T * E + g * cg;
  const float* xs = x + off;
  const float* dys = dy + off;
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  const double n = (double)T * cg;

  double s1 = 0.0, s2 = 0.0;  // sum gamma*dy, sum gamma*dy*xhat (only thread rows tt == 0 contribute)
  for (int j0 = 0; j0 < cg; j0 += L.TJ) {
    const int j = j0 + tj;
    float a_dy = 0.f, a_dyx = 0.f;
    if (j < cg) {
      for (int t = tt; t < T; t += L.TT) {
        const long long o = (long long)t * E + j;
        const float d = dys[o];
        a_dy += d;
        a_dyx = fmaf(d, (xs[o] - mu) * rs, a_dyx);
      }
    }
    __syncthreads();
    red[0][threadIdx.x] = a_dyx;
    red[1][threadIdx.x] = a_dy;
    __syncthreads();
    if (tt == 0 && j < cg) {
      float sg = 0.f, sb = 0.f;
      for (int r = 0; r < L.TT; ++r) {  // fixed order
        sg += red[0][r * L.TJ + tj];
        sb += red[1][r * L.TJ + tj];
      }
      part[((long long)0 * B + b) * E + g * cg + j] = sg;
      part[((long long)1 * B + b) * E + g * cg + j] = sb;
      const float ga = gamma[g * cg + j];
      s1 += (double)ga * sb;
      s2 += (double)ga * sg;
    }
  }
  const float m1 = (float)(block_sum_d(s1, shd) / n);
  const float m2 = (float)(block_sum_d(s2, shd) / n);
  float* dxs = dx + off;
  const float* adds = add ? add + off : nullptr;
  for (int j = tj; j < cg; j += L.TJ) {
    const float ga = gamma[g * cg + j];
    for (int t = tt; t < T; t += L.TT) {
      const long long o = (long long)t * E + j;
      const float xh = (xs[o] - mu) * rs;
      float v = rs * (ga * dys[o] - m1 - xh * m2);
      if (adds) v += adds[o];
      dxs[o] = v;
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
  hipLaunchKernelGGL(groupnorm_fwd_kernel, dim3(G, B), dim3(256), 0, as_stream(stream), x, gamma, beta, y, mean, rstd,
                     T, E, G, eps);
  return check_launch("groupnorm_fwd_kernel");
}

extern "C" int dpot_groupnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                                  const float* gamma, const float* add, float* dx, float* dgamma, float* dbeta,
                                  float* part, int B, int T, int E, int G, dpot_stream_t stream) {
  DPOT_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && part, "groupnorm_bwd: null pointer");
  DPOT_REQUIRE(B > 0 && T > 0 && E > 0 && G > 0 && E % G == 0 && B <= 65535, "groupnorm_bwd: bad shape");
  hipLaunchKernelGGL(groupnorm_bwd_kernel, dim3(G, B), dim3(256), 0, as_stream(stream), dy, x, mean, rstd, gamma, add,
                     dx, part, B, T, E, G);
  int rc = check_launch("groupnorm_bwd_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(groupnorm_param_grad_kernel, dim3(cdiv(E, 256)), dim3(256), 0, as_stream(stream),
                     (const float*)part, dgamma, dbeta, B, E);
  return check_launch("groupnorm_param_grad_kernel");
}
