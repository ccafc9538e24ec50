// misc.hip - data movement and small reductions around the GEMMs (all HBM-bound):
//   patchify / unpatchify (strided 8x8 conv <-> GEMM rows), pixel shuffle (ConvTranspose k=s=P), padded copies,
//   transposes, column sums (bias gradients), pos_embed gradient, token mean (cls head), TimeAggregator scaling.
#include "common.h"

namespace dpot {

// ---------------------------------------------------------------------------------------------------------
// patchify: one workgroup per patch site (b,px,py).  The P input rows of a site are P contiguous runs of
// P*T*C floats -> staged through LDS with coalesced reads, then written as T consecutive GEMM rows of
// K0 = (C+3)*P*P floats (coalesced).  Columns C..C+2 are the unit grid (x, y, t).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, const float* __restrict__ gx,
                                                       const float* __restrict__ gy, const float* __restrict__ gt,
                                                       float* __restrict__ A, int X, int Y, int T, int C, int P) {
  extern __shared__ float sm[];  // [P][P*T*C + 1]: row i of the site, contiguous as in memory (+1: rows on distinct banks)
  const int w = Y / P, h = X / P;
  const int site = blockIdx.x;
  const int py = site % w, px = (site / w) % h, b = site / (w * h);
  const int TC = T * C, run = P * TC, rp = run + 1;
  for (int i = 0; i < P; ++i) {
    const float* src = x + (((long long)b * X + px * P + i) * Y + py * P) * TC;
    for (int r = threadIdx.x; r < run; r += 256) sm[i * rp + r] = src[r];
  }
  __syncthreads();
  const int PP = P * P, K0 = (C + 3) * PP;
  float* out = A + (long long)site * T * K0;
  // one (c,i,j) decomposition per column k, reused for the T rows (integer division is ~30 VALU instructions)
  for (int k = threadIdx.x; k < K0; k += 256) {
    const int c = k / PP, rem = k - c * PP;
    const int i = rem / P, j = rem - i * P;
    if (c < C) {
      const float* s0 = sm + i * rp + j * TC + c;
      for (int t = 0; t < T; ++t) out[(long long)t * K0 + k] = s0[t * C];
    } else if (c == C) {
      const float v = gx[px * P + i];
      for (int t = 0; t < T; ++t) out[(long long)t * K0 + k] = v;
    } else if (c == C + 1) {
      const float v = gy[py * P + j];
      for (int t = 0; t < T; ++t) out[(long long)t * K0 + k] = v;
    } else {
      for (int t = 0; t < T; ++t) out[(long long)t * K0 + k] = gt[t];
    }
  }
}

__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ dA, float* __restrict__ dx, int X,
                                                         int Y, int T, int C, int P) {
  extern __shared__ float sm[];  // [P][P*T*C + 1]
  const int w = Y / P, h = X / P;
  const int site = blockIdx.x;
  const int py = site % w, px = (site / w) % h, b = site / (w * h);
  const int TC = T * C, run = P * TC, rp = run + 1;
  const int PP = P * P, K0 = (C + 3) * PP, KC = C * PP;
  const float* in = dA + (long long)site * T * K0;
  for (int k = threadIdx.x; k < KC; k += 256) {
    const int c = k / PP, rem = k - c * PP;
    const int i = rem / P, j = rem - i * P;
    float* d0 = sm + i * rp + j * TC + c;
    for (int t = 0; t < T; ++t) d0[t * C] = in[(long long)t * K0 + k];
  }
  __syncthreads();
  for (int i = 0; i < P; ++i) {
    float* dst = dx + (((long long)b * X + px * P + i) * Y + py * P) * TC;
    for (int r = threadIdx.x; r < run; r += 256) dst[r] = sm[i * rp + r];
  }
}

// z[(b,px,py,i,j), Cc] <-> out[b, px*P+i, py*P+j, Cc]
__global__ void pixel_shuffle_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w, int P,
                                     int Cc, int inverse, long long total) {
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    // idx enumerates the image layout [b, X, Y, Cc]
    const int c = (int)(idx % Cc);
    long long r = idx / Cc;
    const int Yy = (int)(r % (w * P));
    r /= (w * P);
    const int Xx = (int)(r % (h * P));
    const long long b = r / (h * P);
    const int px = Xx / P, i = Xx % P, py = Yy / P, j = Yy % P;
    const long long zi = (((((b * h + px) * w + py) * P + i) * P + j)) * Cc + c;
    if (inverse) dst[zi] = src[idx];
    else dst[idx] = src[zi];
  }
}

__global__ void copy2d_pad_kernel(const float* __restrict__ src, int sR, int sC, float* __restrict__ dst, int dR,
                                  int dC) {
  const long long total = (long long)dR * dC;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int r = (int)(idx / dC), c = (int)(idx % dC);
    dst[idx] = (r < sR && c < sC) ? src[(long long)r * sC + c] : 0.f;
  }
}

// Small weight-only LAYOUT jobs in ONE launch (round 4): dst[i0][i1][i2] (contiguous, d0 x d1 x d2) =
//   (i0 < v0 && i1 < v1 && i2 < v2 ? src[i0 s0 + i1 s1 + i2 s2] : 0) + (add ? add[i2] : 0)
// covers the zero-padded copies, small transposes, bias broadcasts and "+ bias" passes that DPOTNet derives from its
// parameters every optimiser step (padded patch-conv weights, pos_embed^T + conv bias, de-embed bias per pixel, padded tail
// weights: eight launches of 2-18 us - ~4 us each inside the replayed graph).  The table lives in device memory (a by-value
// table indexed by blockIdx.y is copied to scratch by every thread); the destinations are persistent buffers.
__global__ __launch_bounds__(256) void layout_jobs_kernel(const dpot_layout_job* __restrict__ jobs) {
  const dpot_layout_job j = jobs[blockIdx.y];
  const unsigned total = (unsigned)j.d0 * (unsigned)j.d1 * (unsigned)j.d2;
  const unsigned d12 = (unsigned)j.d1 * (unsigned)j.d2;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned i0 = idx / d12, r = idx - i0 * d12;
    const unsigned i1 = r / (unsigned)j.d2, i2 = r - i1 * (unsigned)j.d2;
    float v = 0.f;
    if ((int)i0 < j.v0 && (int)i1 < j.v1 && (int)i2 < j.v2) v = j.src[(long long)i0 * j.s0 + (long long)i1 * j.s1 + (long long)i2 * j.s2];
    if (j.add) v += j.add[i2];
    j.dst[idx] = v;
  }
}

// dst[C,R] = src[R,C]^T per batch; 32x32 tiles through LDS (+1 padding: conflict-free)
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int R,
                                                          int C) {
  __shared__ float tile[32][33];
  const long long boff = (long long)blockIdx.z * R * C;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    if (r < R && c < C) tile[k][tx] = src[boff + (long long)r * C + c];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (r < R && c < C) dst[boff + (long long)c * R + r] = tile[tx][k];
  }
}

// column sums: grid (cdiv(N,CW), parts); block = CW column lanes x (256/CW) row lanes, rows strided by the lanes,
// 4 independent accumulators per thread (memory-level parallelism), fixed-order combine -> deterministic.
// Stage 2 runs the same kernel over the [parts, N] partials with parts = 1.
template <int CW>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int M, int N, int ld,
                                                     float* __restrict__ out, int rows_per_part) {
  constexpr int RL = 256 / CW;
  __shared__ float red[RL][CW];
  const int tc = threadIdx.x % CW, tr = threadIdx.x / CW;
  const int n = blockIdx.x * CW + tc;
  const int m0 = blockIdx.y * rows_per_part;
  int m1 = m0 + rows_per_part;
  if (m1 > M) m1 = M;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (n < N) {
    const float* p = X + n;
    int m = m0 + tr;
    for (; m + 3 * RL < m1; m += 4 * RL) {
      s0 += p[(long long)m * ld];
      s1 += p[(long long)(m + RL) * ld];
      s2 += p[(long long)(m + 2 * RL) * ld];
      s3 += p[(long long)(m + 3 * RL) * ld];
    }
    for (; m < m1; m += RL) s0 += p[(long long)m * ld];
  }
  red[tr][tc] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (tr == 0 && n < N) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RL; ++r) s += red[r][tc];
    out[(long long)blockIdx.y * N + n] = s;
  }
}

// out[r, n] = sum_{b,t} X[((b*R + r)*T + t)*N + n]
// block = 64 column lanes x 4 lanes over the (b, t) pairs; fixed-order combine
__global__ __launch_bounds__(256) void group_rowsum_kernel(const float* __restrict__ X, float* __restrict__ out, int B,
                                                           int R, int T, int N) {
  __shared__ float red[4][64];
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tc;
  const int r = blockIdx.y;
  float s0 = 0.f, s1 = 0.f;
  if (n < N) {
    const int BT = B * T;
    int q = tr;
    for (; q + 4 < BT; q += 8) {
      const int b0 = q / T, t0 = q % T, b1 = (q + 4) / T, t1 = (q + 4) % T;
      s0 += X[(((long long)b0 * R + r) * T + t0) * N + n];
      s1 += X[(((long long)b1 * R + r) * T + t1) * N + n];
    }
    for (; q < BT; q += 4) s0 += X[(((long long)(q / T) * R + r) * T + (q % T)) * N + n];
  }
  red[tr][tc] = s0 + s1;
  __syncthreads();
  if (tr == 0 && n < N) out[(long long)r * N + n] = (red[0][tc] + red[1][tc]) + (red[2][tc] + red[3][tc]);
}

__global__ __launch_bounds__(256) void token_mean_kernel(const float* __restrict__ x, float* __restrict__ y, int T,
                                                         int E) {
  __shared__ float red[4][64];
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + tc;
  const int b = blockIdx.y;
  float s0 = 0.f, s1 = 0.f;
  if (e < E) {
    const float* p = x + (long long)b * T * E + e;
    int t = tr;
    for (; t + 28 < T; t += 32) {          // eight loads in flight (two were latency bound: 16.8 MB in 11.6 us)
      const float a0 = p[(long long)t * E], a1 = p[(long long)(t + 4) * E], a2 = p[(long long)(t + 8) * E];
      const float a3 = p[(long long)(t + 12) * E], a4 = p[(long long)(t + 16) * E], a5 = p[(long long)(t + 20) * E];
      const float a6 = p[(long long)(t + 24) * E], a7 = p[(long long)(t + 28) * E];
      s0 += (a0 + a1) + (a2 + a3);
      s1 += (a4 + a5) + (a6 + a7);
    }
    for (; t < T; t += 4) s0 += p[(long long)t * E];
  }
  red[tr][tc] = s0 + s1;
  __syncthreads();
  if (tr == 0 && e < E)
    y[(long long)b * E + e] = ((red[0][tc] + red[1][tc]) + (red[2][tc] + red[3][tc])) / (float)T;
}
__global__ void token_mean_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ add,
                                      float* __restrict__ dx, int T, int E, long long total) {
  const float inv = 1.0f / (float)T;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int e = (int)(idx % E);
    const long long b = idx / ((long long)T * E);
    float v = dy[b * E + e] * inv;
    if (add) v += add[idx];
    dx[idx] = v;
  }
}

__global__ void bias_add_kernel(const float* __restrict__ x, const float* __restrict__ v, float* __restrict__ y, int N,
                                long long total) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    y[i] = (x ? x[i] : 0.f) + v[i % N];
}

__global__ void scale_shift_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                   const float* __restrict__ shift, float* __restrict__ y, int T, int E,
                                   long long total) {
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int e = (int)(idx % E);
    const long long b = idx / ((long long)T * E);
    y[idx] = fmaf(x[idx], scale[b * E + e], shift[b * E + e]);
  }
}

// TimeAggregator exp_mlp: one block per (t, i) row of w[T,E,E]
__global__ __launch_bounds__(256) void timeagg_scale_w_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                              const float* __restrict__ tt, float* __restrict__ ws,
                                                              int E) {
  const int i = blockIdx.x, t = blockIdx.y;
  const float cs = cosf(tt[t] * gamma[i]);
  const long long off = ((long long)t * E + i) * E;
  for (int j = threadIdx.x; j < E; j += 256) ws[off + j] = w[off + j] * cs;
}
// one block per i: dw rows for every t, and dgamma[i]
__global__ __launch_bounds__(256) void timeagg_scale_w_bwd_kernel(const float* __restrict__ dws, const float* __restrict__ w,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ tt, float* __restrict__ dw,
                                                                  float* __restrict__ dgamma, int T, int E) {
  __shared__ float sh[16];
  const int i = blockIdx.x;
  const float ga = gamma[i];
  // dgamma[i] = sum_t (-sin(t g) t) sum_j dws w  - one accumulator per thread over (t, j) and ONE block reduction
  // (a reduction per t put T barrier phases in series: 20 us for 10 MB)
  float acc = 0.f;
  for (int t = 0; t < T; ++t) {
    const float tv = tt[t];
    const float arg = tv * ga;
    const float cs = cosf(arg), coef = -sinf(arg) * tv;
    const long long off = ((long long)t * E + i) * E;
    float s = 0.f;
    for (int j = threadIdx.x; j < E; j += 256) {
      const float d = dws[off + j];
      dw[off + j] = d * cs;
      s = fmaf(d, w[off + j], s);
    }
    acc = fmaf(s, coef, acc);
  }
  const float dg = block_sum(acc, sh);
  if (dgamma && threadIdx.x == 0) dgamma[i] = dg;
}

// adjoint of y = x * scale[b,e] + shift[b,e]: dx = dy * scale; dscale[b,e] = sum_t dy*x; dshift[b,e] = sum_t dy
// grid (ceil(E/64), B), block 256 = 64 channels x 4 token lanes; fixed-order combination -> deterministic
__global__ __launch_bounds__(256) void scale_shift_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ scale, float* __restrict__ dx,
                                                              float* __restrict__ dscale, float* __restrict__ dshift,
                                                              int T, int E) {
  __shared__ float s1[256], s2[256];
  const int b = blockIdx.y, e = blockIdx.x * 64 + (threadIdx.x & 63), tl = threadIdx.x >> 6;
  float a1 = 0.f, a2 = 0.f;
  if (e < E) {
    const float sc = scale[(long long)b * E + e];
    for (int t = tl; t < T; t += 4) {
      const long long idx = ((long long)b * T + t) * E + e;
      const float g = dy[idx];
      dx[idx] = g * sc;
      a1 = fmaf(g, x[idx], a1);
      a2 += g;
    }
  }
  s1[threadIdx.x] = a1;
  s2[threadIdx.x] = a2;
  __syncthreads();
  if (tl == 0 && e < E) {
    const int c = threadIdx.x;
    dscale[(long long)b * E + e] = (s1[c] + s1[c + 64]) + (s1[c + 128] + s1[c + 192]);
    dshift[(long long)b * E + e] = (s2[c] + s2[c + 64]) + (s2[c + 128] + s2[c + 192]);
  }
}

static inline unsigned grid_for(long long n, int cap = 8192) {
  long long g = (n + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace dpot

using namespace dpot;

extern "C" int dpot_patchify(const float* x, const float* gx, const float* gy, const float* gt, float* A, int B,
                             int X, int Y, int T, int C, int P, dpot_stream_t stream) {
  DPOT_REQUIRE(x && gx && gy && gt && A, "patchify: null pointer");
  DPOT_REQUIRE(B > 0 && P > 0 && X % P == 0 && Y % P == 0 && T > 0 && C > 0, "patchify: bad shape");
  const size_t lds = sizeof(float) * (size_t)P * ((size_t)P * T * C + 1);
  DPOT_REQUIRE(lds <= 64 * 1024, "patchify: patch slab of %zu bytes exceeds 64 KiB", lds);
  const long long sites = (long long)B * (X / P) * (Y / P);
  DPOT_REQUIRE(sites < (1ll << 31), "patchify: too many sites");
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)sites), dim3(256), lds, as_stream(stream), x, gx, gy, gt, A, X, Y,
                     T, C, P);
  return check_launch("patchify_kernel");
}

extern "C" int dpot_unpatchify(const float* dA, float* dx, int B, int X, int Y, int T, int C, int P,
                               dpot_stream_t stream) {
  DPOT_REQUIRE(dA && dx, "unpatchify: null pointer");
  DPOT_REQUIRE(B > 0 && P > 0 && X % P == 0 && Y % P == 0 && T > 0 && C > 0, "unpatchify: bad shape");
  const size_t lds = sizeof(float) * (size_t)P * ((size_t)P * T * C + 1);
  DPOT_REQUIRE(lds <= 64 * 1024, "unpatchify: patch slab of %zu bytes exceeds 64 KiB", lds);
  const long long sites = (long long)B * (X / P) * (Y / P);
  hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)sites), dim3(256), lds, as_stream(stream), dA, dx, X, Y, T, C,
                     P);
  return check_launch("unpatchify_kernel");
}

extern "C" int dpot_pixel_shuffle(const float* z, float* out, int B, int h, int w, int P, int Cc, int inverse,
                                  dpot_stream_t stream) {
  DPOT_REQUIRE(z && out && B > 0 && h > 0 && w > 0 && P > 0 && Cc > 0, "pixel_shuffle: bad argument");
  const long long total = (long long)B * h * P * w * P * Cc;
  hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), z, out, h, w, P, Cc,
                     inverse, total);
  return check_launch("pixel_shuffle_kernel");
}

extern "C" int dpot_copy2d_pad(const float* src, int sR, int sC, float* dst, int dR, int dC, dpot_stream_t stream) {
  DPOT_REQUIRE(src && dst && sR > 0 && sC > 0 && dR > 0 && dC > 0, "copy2d_pad: bad argument");
  hipLaunchKernelGGL(copy2d_pad_kernel, dim3(grid_for((long long)dR * dC)), dim3(256), 0, as_stream(stream), src, sR,
                     sC, dst, dR, dC);
  return check_launch("copy2d_pad_kernel");
}

extern "C" int dpot_layout_jobs(const dpot_layout_job* jobs_dev, int njobs, int64_t max_elems, dpot_stream_t stream) {
  DPOT_REQUIRE(jobs_dev && njobs > 0 && njobs <= 65535 && max_elems > 0 && max_elems < (1ll << 31), "layout_jobs: bad argument");
  long long g = (max_elems + 255) / 256;
  if (g > 256) g = 256;
  hipLaunchKernelGGL(layout_jobs_kernel, dim3((unsigned)g, njobs), dim3(256), 0, as_stream(stream), jobs_dev);
  return check_launch("layout_jobs_kernel");
}

extern "C" int dpot_transpose2d(const float* src, float* dst, int nbatch, int R, int C, dpot_stream_t stream) {
  DPOT_REQUIRE(src && dst && nbatch > 0 && R > 0 && C > 0 && nbatch <= 65535 && cdiv(R, 32) <= 65535,
               "transpose2d: bad argument");
  hipLaunchKernelGGL(transpose2d_kernel, dim3(cdiv(C, 32), cdiv(R, 32), nbatch), dim3(256), 0, as_stream(stream), src,
                     dst, R, C);
  return check_launch("transpose2d_kernel");
}

extern "C" int dpot_colsum_parts(int M) {
  int parts = cdiv(M, 256);
  if (parts > 1024) parts = 1024;
  if (parts < 1) parts = 1;
  return parts;
}

static void launch_colsum(const float* X, int M, int N, int ld, float* out, int parts, hipStream_t s) {
  const int rpp = cdiv(M, parts);
  if (N <= 16) {
    hipLaunchKernelGGL(colsum_kernel<16>, dim3(cdiv(N, 16), parts), dim3(256), 0, s, X, M, N, ld, out, rpp);
  } else if (N <= 32) {
    hipLaunchKernelGGL(colsum_kernel<32>, dim3(cdiv(N, 32), parts), dim3(256), 0, s, X, M, N, ld, out, rpp);
  } else {
    hipLaunchKernelGGL(colsum_kernel<64>, dim3(cdiv(N, 64), parts), dim3(256), 0, s, X, M, N, ld, out, rpp);
  }
}

extern "C" int dpot_colsum(const float* X, int M, int N, int ld, float* out, float* part, dpot_stream_t stream) {
  DPOT_REQUIRE(X && out && part && M > 0 && N > 0 && ld >= N, "colsum: bad argument");
  // aim for ~1024 workgroups in stage 1 whatever the aspect ratio
  int parts = dpot_colsum_parts(M);
  const int colblocks = cdiv(N, N <= 16 ? 16 : N <= 32 ? 32 : 64);
  const int want = cdiv(1024, colblocks);
  if (parts > want) parts = want;
  if (parts <= 1) {
    launch_colsum(X, M, N, ld, out, 1, as_stream(stream));
    return check_launch("colsum_kernel");
  }
  launch_colsum(X, M, N, ld, part, parts, as_stream(stream));
  int rc = check_launch("colsum_kernel");
  if (rc) return rc;
  launch_colsum(part, parts, N, N, out, 1, as_stream(stream));
  return check_launch("colsum_kernel(stage 2)");
}

// stage 2 of a column sum whose result goes to several destinations (the partial-row matrix of the fused out-layer tail
// holds five parameter gradients side by side; their slots in the flat gradient buffer are not adjacent)
struct ScatterSegs {
  int n;
  int start[8], len[8];
  float* dst[8];
};
__global__ __launch_bounds__(256) void colsum_scatter_kernel(const float* __restrict__ part, int parts, int N,
                                                             const ScatterSegs segs) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  float a = 0.f;
  for (int r = 0; r < parts; ++r) a += part[(long long)r * N + j];      // fixed order
  for (int s = 0; s < segs.n; ++s)
    if (j >= segs.start[s] && j < segs.start[s] + segs.len[s]) segs.dst[s][j - segs.start[s]] = a;
}

extern "C" int dpot_colsum_scatter(const float* X, int M, int N, int ld, float* part, int nseg, const int* seg_start,
                                   const int* seg_len, float* const* seg_dst, dpot_stream_t stream) {
  DPOT_REQUIRE(X && part && M > 0 && N > 0 && ld >= N && nseg > 0 && nseg <= 8 && seg_start && seg_len && seg_dst,
               "colsum_scatter: bad argument");
  ScatterSegs segs;
  segs.n = nseg;
  for (int s = 0; s < nseg; ++s) {
    DPOT_REQUIRE(seg_start[s] >= 0 && seg_len[s] > 0 && seg_start[s] + seg_len[s] <= N && seg_dst[s],
                 "colsum_scatter: bad segment %d", s);
    segs.start[s] = seg_start[s]; segs.len[s] = seg_len[s]; segs.dst[s] = seg_dst[s];
  }
  int parts = dpot_colsum_parts(M);
  const int want = cdiv(1024, cdiv(N, N <= 16 ? 16 : N <= 32 ? 32 : 64));
  if (parts > want) parts = want;
  if (parts < 1) parts = 1;
  launch_colsum(X, M, N, ld, part, parts, as_stream(stream));
  int rc = check_launch("colsum_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_scatter_kernel, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream), (const float*)part,
                     parts, N, segs);
  return check_launch("colsum_scatter_kernel");
}

extern "C" int dpot_group_rowsum(const float* X, float* out, int B, int R, int T, int N, dpot_stream_t stream) {
  DPOT_REQUIRE(X && out && B > 0 && R > 0 && T > 0 && N > 0 && R <= 65535, "group_rowsum: bad argument");
  hipLaunchKernelGGL(group_rowsum_kernel, dim3(cdiv(N, 64), R), dim3(256), 0, as_stream(stream), X, out, B, R, T, N);
  return check_launch("group_rowsum_kernel");
}

namespace dpot {
// ---- few-row Linear: y[M, N] = act(x[M, K] W[N, K]^T + b), M <= a few dozen rows (models/dpot.py:333-336, the cls_head
// on the token mean: M = batch).  A GEMM tile grid has 8 output tiles here and the split-K + reduce pair that spreads
// it costs 12 us per layer, all latency; as a mat-vec family the layer is one 1-MiB pass over W.  A workgroup stages
// 32 rows x 512 k of x in LDS (64 KiB, one round of 16 float4 loads per thread); each of its 8 waves owns one output
// column: its W row chunk sits in registers (2 float4 per lane), the 32 row products are FMA chains over conflict-free
// ds_read_b128, reduced across the lanes by a halving butterfly (32 shuffles for 32 values; lane l ends with row
// l >> 1).  fp32 throughout, fixed order.
constexpr int SL_ROWS = 32, SL_KCH = 512, SL_WAVES = 8;
// one butterfly stage (compile-time HALF: a run-time loop bound would turn the register array into select chains)
template <int HALF>
__device__ __forceinline__ void sl_stage(float (&part)[SL_ROWS], int lane) {
  const bool hi = (lane & (2 * HALF)) != 0;
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const float send = hi ? part[i] : part[HALF + i];
    const float keep = hi ? part[HALF + i] : part[i];
    part[i] = keep + __shfl_xor(send, 2 * HALF, 64);
  }
}
__global__ __launch_bounds__(64 * SL_WAVES) void small_linear_kernel(const float* __restrict__ x, int ldx,
                                                                     const float* __restrict__ W, int ldw,
                                                                     const float* __restrict__ bias,
                                                                     float* __restrict__ y, float* __restrict__ pre,
                                                                     int ldy, int M, int N, int K, int act) {
  extern __shared__ __attribute__((aligned(16))) float xs[];      // [32][512]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_raw = blockIdx.x * SL_WAVES + wave;
  const int n = n_raw < N ? n_raw : N - 1;                          // clamped (barriers below); masked at the store
  const float bn = bias ? bias[n] : 0.f;
  for (int m0 = 0; m0 < M; m0 += SL_ROWS) {
    float part[SL_ROWS];
#pragma unroll
    for (int i = 0; i < SL_ROWS; ++i) part[i] = 0.f;
    for (int k0 = 0; k0 < K; k0 += SL_KCH) {
      const float4 w0 = *reinterpret_cast<const float4*>(W + (long long)n * ldw + k0 + 4 * lane);
      const float4 w1 = *reinterpret_cast<const float4*>(W + (long long)n * ldw + k0 + 256 + 4 * lane);
      __syncthreads();                                              // the previous chunk has been consumed
      float4 st[SL_ROWS * SL_KCH / 4 / (64 * SL_WAVES)];
#pragma unroll
      for (int j = 0; j < SL_ROWS * SL_KCH / 4 / (64 * SL_WAVES); ++j) {
        const int q = tid + j * 64 * SL_WAVES;                      // float4 index: row = q / 128, quad = q % 128
        const int m = m0 + (q >> 7) < M ? m0 + (q >> 7) : M - 1;
        st[j] = *reinterpret_cast<const float4*>(x + (long long)m * ldx + k0 + 4 * (q & 127));
      }
#pragma unroll
      for (int j = 0; j < SL_ROWS * SL_KCH / 4 / (64 * SL_WAVES); ++j)
        *reinterpret_cast<float4*>(xs + 4 * (tid + j * 64 * SL_WAVES)) = st[j];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < SL_ROWS; ++i) {
        const float4 a = *reinterpret_cast<const float4*>(xs + i * SL_KCH + 4 * lane);
        const float4 b = *reinterpret_cast<const float4*>(xs + i * SL_KCH + 256 + 4 * lane);
        float t = part[i];
        t = fmaf(a.x, w0.x, t); t = fmaf(a.y, w0.y, t); t = fmaf(a.z, w0.z, t); t = fmaf(a.w, w0.w, t);
        t = fmaf(b.x, w1.x, t); t = fmaf(b.y, w1.y, t); t = fmaf(b.z, w1.z, t); t = fmaf(b.w, w1.w, t);
        part[i] = t;
      }
    }
    // halving butterfly: after the stage with stride s a lane keeps the half of its values selected by (lane & s)
    sl_stage<16>(part, lane); sl_stage<8>(part, lane); sl_stage<4>(part, lane); sl_stage<2>(part, lane);
    sl_stage<1>(part, lane);
    const float v = part[0] + __shfl_xor(part[0], 1, 64) + bn;
    const int m = m0 + (lane >> 1);
    if ((lane & 1) == 0 && m < M && n_raw < N) {
      if (pre) pre[(long long)m * ldy + n] = v;
      y[(long long)m * ldy + n] = act_fwd(act, v);
    }
  }
}

}  // namespace dpot

extern "C" int dpot_token_mean(const float* x, float* y, int B, int T, int E, dpot_stream_t stream) {
  DPOT_REQUIRE(x && y && B > 0 && T > 0 && E > 0 && B <= 65535, "token_mean: bad argument");
  hipLaunchKernelGGL(token_mean_kernel, dim3(cdiv(E, 64), B), dim3(256), 0, as_stream(stream), x, y, T, E);
  return check_launch("token_mean_kernel");
}

extern "C" int dpot_token_mean_bwd(const float* dy, const float* add, float* dx, int B, int T, int E,
                                   dpot_stream_t stream) {
  DPOT_REQUIRE(dy && dx && B > 0 && T > 0 && E > 0, "token_mean_bwd: bad argument");
  const long long total = (long long)B * T * E;
  hipLaunchKernelGGL(token_mean_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dy, add, dx, T, E,
                     total);
  return check_launch("token_mean_bwd_kernel");
}

extern "C" int dpot_bias_add(const float* x, const float* v, float* y, int R, int N, dpot_stream_t stream) {
  DPOT_REQUIRE(v && y && R > 0 && N > 0, "bias_add: bad argument");   // x == NULL: y = v tiled R times
  const long long total = (long long)R * N;
  hipLaunchKernelGGL(bias_add_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, v, y, N, total);
  return check_launch("bias_add_kernel");
}

extern "C" int dpot_scale_shift(const float* x, const float* scale, const float* shift, float* y, int B, int T, int E,
                                dpot_stream_t stream) {
  DPOT_REQUIRE(x && scale && shift && y && B > 0 && T > 0 && E > 0, "scale_shift: bad argument");
  const long long total = (long long)B * T * E;
  hipLaunchKernelGGL(scale_shift_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, scale, shift, y, T,
                     E, total);
  return check_launch("scale_shift_kernel");
}

extern "C" int dpot_scale_shift_bwd(const float* dy, const float* x, const float* scale, float* dx, float* dscale,
                                    float* dshift, int B, int T, int E, dpot_stream_t stream) {
  DPOT_REQUIRE(dy && x && scale && dx && dscale && dshift && B > 0 && B <= 65535 && T > 0 && E > 0,
               "scale_shift_bwd: bad argument");
  hipLaunchKernelGGL(scale_shift_bwd_kernel, dim3((unsigned)cdiv(E, 64), B), dim3(256), 0, as_stream(stream), dy, x,
                     scale, dx, dscale, dshift, T, E);
  return check_launch("scale_shift_bwd_kernel");
}

extern "C" int dpot_timeagg_scale_w(const float* w, const float* gamma, const float* tt, float* ws, int T, int E,
                                    dpot_stream_t stream) {
  DPOT_REQUIRE(w && gamma && tt && ws && T > 0 && E > 0 && T <= 65535, "timeagg_scale_w: bad argument");
  hipLaunchKernelGGL(timeagg_scale_w_kernel, dim3(E, T), dim3(256), 0, as_stream(stream), w, gamma, tt, ws, E);
  return check_launch("timeagg_scale_w_kernel");
}

extern "C" int dpot_timeagg_scale_w_bwd(const float* dws, const float* w, const float* gamma, const float* tt,
                                        float* dw, float* dgamma, int T, int E, dpot_stream_t stream) {
  DPOT_REQUIRE(dws && w && gamma && tt && dw && T > 0 && E > 0, "timeagg_scale_w_bwd: bad argument");
  hipLaunchKernelGGL(timeagg_scale_w_bwd_kernel, dim3(E), dim3(256), 0, as_stream(stream), dws, w, gamma, tt, dw,
                     dgamma, T, E);
  return check_launch("timeagg_scale_w_bwd_kernel");
}

extern "C" int dpot_small_linear_supported(int M, int N, int K) {
  return M > 0 && M <= 128 && N > 0 && K >= SL_KCH && K % SL_KCH == 0 ? 1 : 0;
}

extern "C" int dpot_small_linear(const float* x, int ldx, const float* W, int ldw, const float* bias, float* y, float* pre,
                                 int ldy, int M, int N, int K, int act, dpot_stream_t stream) {
  DPOT_REQUIRE(x && W && y, "small_linear: null operand");
  DPOT_REQUIRE(dpot_small_linear_supported(M, N, K), "small_linear: unsupported shape M=%d N=%d K=%d", M, N, K);
  DPOT_REQUIRE(ldx >= K && ldw >= K && ldy >= N && ldx % 4 == 0 && ldw % 4 == 0 && aligned16(x) && aligned16(W),
               "small_linear: bad leading dimension / alignment");
  const size_t lds = sizeof(float) * SL_ROWS * SL_KCH;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(small_linear_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  DPOT_REQUIRE(attr == hipSuccess, "small_linear: cannot reserve %zu bytes of LDS", lds);
  hipLaunchKernelGGL(small_linear_kernel, dim3((unsigned)cdiv(N, SL_WAVES)), dim3(64 * SL_WAVES), lds, as_stream(stream),
                     x, ldx, W, ldw, bias, y, pre, ldy, M, N, K, act);
  return check_launch("small_linear_kernel");
}
