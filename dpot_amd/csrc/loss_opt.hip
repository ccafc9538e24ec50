// loss_opt.hip - masked relative-L2 loss (utils/criterion.py:38-59), global gradient norm + clip
// (train_temporal.py:228), fused flat-buffer Adam (utils/optimizer.py:26-52), noise injection
// (train_temporal.py:205).  All HBM-bound streaming kernels; no host synchronisation anywhere: the loss,
// the gradient norm and the learning-rate schedule live in device memory so the whole step can be captured
// in one hipGraph.
#include "common.h"

namespace dpot {

struct ChanLayout {
  int CP, TS;  // CP = pow2 >= C channel lanes, TS = row lanes; CP*TS == blockDim
};
__device__ __forceinline__ ChanLayout chan_layout(int C, int nthreads) {
  int cp = 1;
  while (cp < C) cp <<= 1;
  ChanLayout l;
  l.CP = cp;
  l.TS = nthreads / cp;
  return l;
}

// grid (B, NCH): one 1024-thread block per (sample, chunk of the S axis):
// part[b, chunk, c, 0..2] = partial {sum ((x-y)m)^2, sum (y m)^2, sum m}; combined in fixed order by the final kernel
__global__ __launch_bounds__(1024) void rel_l2_stats_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ mask, float* __restrict__ part,
                                                            int S, int C, int Tt) {
  __shared__ double red[3][1024];
  const int b = blockIdx.x;
  const int nch = gridDim.y, chunk = blockIdx.y;
  const int rows = (S + nch - 1) / nch;
  const int s_beg = chunk * rows;
  const int s_end = min(S, s_beg + rows);
  const ChanLayout L = chan_layout(C, 1024);
  const int tc = threadIdx.x % L.CP, ts = threadIdx.x / L.CP;
  double d2 = 0.0, y2 = 0.0, ms = 0.0;
  if (tc < C) {
    const float* xb = x + (long long)b * S * C;
    const float* yb = y + (long long)b * S * C;
    const float* mb = mask ? mask + (long long)b * (S / Tt) * C : nullptr;
    for (int s = s_beg + ts; s < s_end; s += L.TS) {
      const float m = mb ? mb[(long long)(s / Tt) * C + tc] : 1.f;
      const float yv = yb[(long long)s * C + tc] * m;
      const float d = xb[(long long)s * C + tc] * m - yv;
      d2 += (double)d * d;
      y2 += (double)yv * yv;
      if ((s % Tt) == 0) ms += m;
    }
  }
  red[0][threadIdx.x] = d2;
  red[1][threadIdx.x] = y2;
  red[2][threadIdx.x] = ms;
  __syncthreads();
  if (ts == 0 && tc < C) {
    double a = 0.0, bq = 0.0, cq = 0.0;
    for (int r = 0; r < L.TS; ++r) {
      a += red[0][r * L.CP + tc];
      bq += red[1][r * L.CP + tc];
      cq += red[2][r * L.CP + tc];
    }
    float* st = part + (((long long)b * nch + chunk) * C + tc) * 4;
    st[0] = (float)a;
    st[1] = (float)bq;
    st[2] = (float)cq;
    st[3] = 0.f;
  }
}

__global__ void rel_l2_final_kernel(float* __restrict__ stats, const float* __restrict__ part, int nchunk,
                                    float* __restrict__ loss, int B, int C, int has_mask) {
  __shared__ double shd[16];
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    double s = 0.0;
    int nch = 0;
    for (int c = 0; c < C; ++c) {
      float* st = stats + ((long long)b * C + c) * 4;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      for (int k = 0; k < nchunk; ++k) {
        const float* pp = part + (((long long)b * nchunk + k) * C + c) * 4;
        a0 += (double)pp[0];
        a1 += (double)pp[1];
        a2 += (double)pp[2];
      }
      st[0] = (float)a0;
      st[1] = (float)a1;
      st[2] = (float)a2;
      st[3] = 0.f;
      s += (double)(sqrtf(st[0]) / (sqrtf(st[1]) + 1e-8f));
      if (st[2] != 0.f) ++nch;
    }
    if (!has_mask) nch = C;
    acc += s / (double)nch;
  }
  acc = block_sum_d(acc, shd);
  if (threadIdx.x == 0) loss[0] = (float)acc;
}

__global__ void rel_l2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                  const float* __restrict__ mask, const float* __restrict__ stats,
                                  const float* __restrict__ gloss, float* __restrict__ dx, int S, int C, int Tt,
                                  long long total) {
  const float g = gloss[0];
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long long r = idx / C;
    const int s = (int)(r % S);
    const long long b = r / S;
    const float* st = stats + (b * C + c) * 4;
    int nch = C;
    if (mask) {
      nch = 0;
      for (int cc = 0; cc < C; ++cc) nch += (stats[(b * C + cc) * 4 + 2] != 0.f) ? 1 : 0;
    }
    const float m = mask ? mask[(b * (S / Tt) + s / Tt) * C + c] : 1.f;
    const float dn = sqrtf(st[0]);
    const float yn = sqrtf(st[1]) + 1e-8f;
    float v = 0.f;
    if (dn > 0.f) v = g * ((x[idx] - y[idx]) * m * m) / (dn * yn * (float)nch);
    dx[idx] = v;
  }
}

// ---- sum of squares (two-stage, fixed order) ------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_part_kernel(const float* __restrict__ g, long long n,
                                                         float* __restrict__ part) {
  __shared__ double shd[16];
  double s = 0.0;
  const long long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = g4[i];
    s += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0)
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += (double)g[i] * g[i];
  s = block_sum_d(s, shd);
  if (threadIdx.x == 0) part[blockIdx.x] = (float)s;
}
__global__ void sumsq_final_kernel(const float* __restrict__ part, int parts, float* __restrict__ out,
                                   int accumulate) {
  __shared__ double shd[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < parts; i += blockDim.x) s += (double)part[i];
  s = block_sum_d(s, shd);
  if (threadIdx.x == 0) out[0] = (float)(s + (accumulate ? (double)out[0] : 0.0));
}

// ---- fused Adam over a flat buffer ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   const float* __restrict__ hyper, const float* __restrict__ sumsq,
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
  const float step = lr / bc1;
  const float sb2 = sqrtf(bc2);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float pv = p[i];
    float gv = g[i] * gs;
    if (wd != 0.f) gv = fmaf(wd, pv, gv);
    const float mv = b1 * m[i] + (1.f - b1) * gv;
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / sb2 + eps;
    p[i] = pv - step * (mv / denom);
  }
}

// ---- noise injection ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void chan_norm_kernel(const float* __restrict__ x, float* __restrict__ norms, int S,
                                                         int C) {
  __shared__ double red[1024];
  const int b = blockIdx.x;
  const ChanLayout L = chan_layout(C, 1024);
  const int tc = threadIdx.x % L.CP, ts = threadIdx.x / L.CP;
  double s2 = 0.0;
  if (tc < C) {
    const float* xb = x + (long long)b * S * C;
    for (int s = ts; s < S; s += L.TS) {
      const float v = xb[(long long)s * C + tc];
      s2 += (double)v * v;
    }
  }
  red[threadIdx.x] = s2;
  __syncthreads();
  if (ts == 0 && tc < C) {
    double a = 0.0;
    for (int r = 0; r < L.TS; ++r) a += red[r * L.CP + tc];
    norms[(long long)b * C + tc] = sqrtf((float)a);
  }
}
__global__ void noise_axpy_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                  const float* __restrict__ norms, float* __restrict__ out, float noise_scale, int S,
                                  int C, long long total) {
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long long b = idx / ((long long)S * C);
    out[idx] = fmaf(noise_scale * norms[b * C + c], eps[idx], x[idx]);
  }
}

static inline unsigned grid_for(long long n, int cap = 4096) {
  long long g = (n + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace dpot

using namespace dpot;

extern "C" int dpot_rel_l2_chunks(int S, int C) {
  long long n = ((long long)S * C + 16383) / 16384;
  if (n > 32) n = 32;
  if (n < 1) n = 1;
  return (int)n;
}

extern "C" int dpot_rel_l2_fwd(const float* x, const float* y, const float* mask, float* stats, float* loss, int B,
                               int S, int C, int Tt, dpot_stream_t stream) {
  DPOT_REQUIRE(x && y && stats && loss, "rel_l2_fwd: null pointer");
  DPOT_REQUIRE(B > 0 && S > 0 && C > 0 && C <= 1024 && Tt > 0 && S % Tt == 0, "rel_l2_fwd: bad shape");
  const int nch = dpot_rel_l2_chunks(S, C);
  float* part = stats + (size_t)B * C * 4;
  hipLaunchKernelGGL(rel_l2_stats_kernel, dim3(B, nch), dim3(1024), 0, as_stream(stream), x, y, mask, part, S, C, Tt);
  int rc = check_launch("rel_l2_stats_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(rel_l2_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), stats, (const float*)part, nch,
                     loss, B, C, mask ? 1 : 0);
  return check_launch("rel_l2_final_kernel");
}

extern "C" int dpot_rel_l2_bwd(const float* x, const float* y, const float* mask, const float* stats,
                               const float* gloss, float* dx, int B, int S, int C, int Tt, dpot_stream_t stream) {
  DPOT_REQUIRE(x && y && stats && gloss && dx, "rel_l2_bwd: null pointer");
  DPOT_REQUIRE(B > 0 && S > 0 && C > 0 && Tt > 0 && S % Tt == 0, "rel_l2_bwd: bad shape");
  const long long total = (long long)B * S * C;
  hipLaunchKernelGGL(rel_l2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, y, mask, stats,
                     gloss, dx, S, C, Tt, total);
  return check_launch("rel_l2_bwd_kernel");
}

extern "C" int dpot_sumsq(const float* g, int64_t n, float* out, float* part, int accumulate, dpot_stream_t stream) {
  DPOT_REQUIRE(g && out && part && n > 0, "sumsq: bad argument");
  DPOT_REQUIRE(aligned16(g), "sumsq: buffer must be 16-byte aligned");
  int parts = (int)grid_for(n >> 2, 1024);
  hipLaunchKernelGGL(sumsq_part_kernel, dim3(parts), dim3(256), 0, as_stream(stream), g, (long long)n, part);
  int rc = check_launch("sumsq_part_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const float*)part, parts, out,
                     accumulate);
  return check_launch("sumsq_final_kernel");
}

extern "C" int dpot_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                              const float* sumsq, float grad_scale, dpot_stream_t stream) {
  DPOT_REQUIRE(p && g && m && v && hyper && n > 0, "adam_step: bad argument");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, as_stream(stream), p, g, m, v, (long long)n,
                     hyper, sumsq, grad_scale);
  return check_launch("adam_kernel");
}

extern "C" int dpot_noise_inject(const float* xx, const float* eps, float* out, float* norms, float noise_scale,
                                 int B, int S, int C, dpot_stream_t stream) {
  DPOT_REQUIRE(xx && eps && out && norms && B > 0 && S > 0 && C > 0 && C <= 1024, "noise_inject: bad argument");
  hipLaunchKernelGGL(chan_norm_kernel, dim3(B), dim3(1024), 0, as_stream(stream), xx, norms, S, C);
  int rc = check_launch("chan_norm_kernel");
  if (rc) return rc;
  const long long total = (long long)B * S * C;
  hipLaunchKernelGGL(noise_axpy_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), xx, eps,
                     (const float*)norms, out, noise_scale, S, C, total);
  return check_launch("noise_axpy_kernel");
}
