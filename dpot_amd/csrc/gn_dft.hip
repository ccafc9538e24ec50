// gn_dft.hip - GroupNorm fused with the neighbouring 2-D DFT of the AFNO mixer ("cheaper half"):
//
//   forward   K1  GroupNorm1 statistics + rfft2          x            -> S (spectrum of GN1(x)), mean1, rstd1
//             K2  irfft2 + x_orig + GroupNorm2           O2, x        -> y1 = irfft2(O2) + GN1(x),  xn2 = GN2(y1), stats2
//   backward  K3  GroupNorm2 backward + rfft2 (adjoint)  dxn2, y1     -> dy1, dO2, parameter-gradient partials of norm2
//             K4  irfft2 (adjoint) + dy1 + GroupNorm1 backward (+ skip)  dS, dy1, x, dout -> dx, partials of norm1
//
// models/dpot.py:165-175 runs  GroupNorm -> AFNO2D (rfft2 .. irfft2 + x_orig) -> GroupNorm  as separate modules; round 2
// ran GroupNorm and the DFTs as separate kernels (5 launches around the mixer, forward).  A (sample, group) workgroup
// holds all 16 x 16 tokens of its E/8 channels, which is everything a GroupNorm needs AND everything a 2-D DFT of those
// channels needs, so each pair is one kernel: one launch and one HBM pass over the field less per pair.
//   * GN1(x) is never materialised: the DFT is linear, so S = a_c * FFT(x) + (DC term of the per-channel constant),
//     a_c = rstd * gamma_c - the statistics are reduced WHILE the row FFTs run, and the residual "x_orig" of the mixer is
//     re-derived from x in K2 (one fused multiply-add per element).
//   * statistics: every thread owns whole 16-token rows; it forms the row's mean and centred sum of squares in
//     registers, and the rows are merged with Chan's formula in double precision (no E[x^2] - E[x]^2 cancellation).
//   * the half-complex intermediate between the two 1-D passes lives in LDS ([ky][x][re|im][channel]: channel on the
//     lanes, conflict free both ways); the backward kernels alias their reduction scratch onto it.
// Covers the 16 x 16 latent grid (128^2 / patch 8: DPOT-Tiny / -Small / -Medium) with 64 or 128 channels per group;
// anything else (DPOT-Large at 256^2: 32 x 32 tokens x 192 channels = 768 KiB per slab) keeps the separate kernels.
#include "common.h"
#define DPOT_DFT_NO_KERNELS
#include "dft_fast.h"

namespace dpot {

constexpr int GD_H = 16, GD_W = 16, GD_WF = 9, GD_T = 1024;

// three block-wide sums (double), fixed order, result broadcast; sh >= 48 doubles.  Contains two barriers.
__device__ __forceinline__ void block_sum3_d(double& a, double& b, double& c, double* sh) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  c = wave_sum_d(c);
  __syncthreads();
  if (lane == 0) {
    sh[wid] = a;
    sh[16 + wid] = b;
    sh[32 + wid] = c;
  }
  __syncthreads();
  double ra = 0.0, rb = 0.0, rc = 0.0;
  for (int i = 0; i < nw; ++i) {
    ra += sh[i];
    rb += sh[16 + i];
    rc += sh[32 + i];
  }
  a = ra;
  b = rb;
  c = rc;
}

// mean and centred sum of squares of one 16-token row
__device__ __forceinline__ void row_stats(const float (&v)[GD_W], float& lm, float& q) {
  float s = 0.f;
#pragma unroll
  for (int y = 0; y < GD_W; ++y) s += v[y];
  lm = s * (1.0f / GD_W);
  q = 0.f;
#pragma unroll
  for (int y = 0; y < GD_W; ++y) {
    const float d = v[y] - lm;
    q = fmaf(d, d, q);
  }
}

// forward row pass: real 16-point FFT of v, the WF one-sided outputs go to Z[(ky*H + xr)*2 + {0,1}][c]
template <int CG>
__device__ __forceinline__ void row_fft_store(float (&v)[GD_W], float* __restrict__ Z, int xr, int c) {
  float vi[GD_W];
#pragma unroll
  for (int y = 0; y < GD_W; ++y) vi[y] = 0.f;
  fft_regs<GD_W, -1>(v, vi);
  fft_sfor<0, GD_WF>([&](auto KY) __attribute__((always_inline)) {
    constexpr int ky = decltype(KY)::value;
    Z[((ky * GD_H + xr) * 2 + 0) * CG + c] = v[brev<GD_W>(ky)];
    Z[((ky * GD_H + xr) * 2 + 1) * CG + c] = vi[brev<GD_W>(ky)];
  });
}

// forward column pass of item (ky, c): complex 16-point FFT over x, kept modes written to spec (planar per channel block)
//   out = z * (ca * wgt) [+ dc at (kx, ky) = (0, 0), real part]
template <int CG>
__device__ __forceinline__ void col_fft_out(const float* __restrict__ Z, float* __restrict__ spec, int b, int chn, int c,
                                            int ky, int nb, int bs, int mx, int my, float mul, float dc) {
  float zr[GD_H], zi[GD_H];
#pragma unroll
  for (int xr = 0; xr < GD_H; ++xr) {
    zr[xr] = Z[((ky * GD_H + xr) * 2 + 0) * CG + c];
    zi[xr] = Z[((ky * GD_H + xr) * 2 + 1) * CG + c];
  }
  const int blk = chn / bs, ci = chn - blk * bs;
  float* __restrict__ outb = spec + (long long)b * mx * my * nb * 2 * bs;      // uniform; the rest fits 32 bits
  const int o0 = (ky * nb + blk) * 2 * bs + ci, kxstride = my * nb * 2 * bs;
  fft_regs<GD_H, -1>(zr, zi);
  fft_sfor<0, GD_H>([&](auto KX) __attribute__((always_inline)) {
    constexpr int kx = decltype(KX)::value;
    if (kx < mx) {
      float re = zr[brev<GD_H>(kx)] * mul;
      if (kx == 0) re += dc;
      outb[o0 + kx * kxstride] = re;
      outb[o0 + kx * kxstride + bs] = zi[brev<GD_H>(kx)] * mul;
    }
  });
}

// inverse column pass of item (ky, c): U[(xr*WF + ky)*2 + {0,1}][c] = wgt * sum_kx S[kx, ky] e^{+2 pi i kx xr / H}
template <int CG>
__device__ __forceinline__ void col_ifft_store(const float* __restrict__ spec, float* __restrict__ U, int b, int chn,
                                               int c, int ky, int nb, int bs, int mx, int my, float wgt) {
  const int blk = chn / bs, ci = chn - blk * bs;
  const float* __restrict__ in = spec + (long long)b * mx * my * nb * 2 * bs;
  const int o0 = (ky * nb + blk) * 2 * bs + ci, kxstride = my * nb * 2 * bs;
  float sr[GD_H], si[GD_H];
#pragma unroll
  for (int kx = 0; kx < GD_H; ++kx) {
    const int kc = kx < mx ? kx : mx - 1;                 // clamped address, selected value: no divergent loads
    const float a = in[o0 + kc * kxstride], bq = in[o0 + kc * kxstride + bs];
    sr[kx] = kx < mx ? a : 0.f;
    si[kx] = kx < mx ? bq : 0.f;
  }
  fft_regs<GD_H, 1>(sr, si);
  fft_sfor<0, GD_H>([&](auto XR) __attribute__((always_inline)) {
    constexpr int xr = decltype(XR)::value;
    U[((xr * GD_WF + ky) * 2 + 0) * CG + c] = sr[brev<GD_H>(xr)] * wgt;
    U[((xr * GD_WF + ky) * 2 + 1) * CG + c] = si[brev<GD_H>(xr)] * wgt;
  });
}

// inverse row pass of item (xr, c): out[yy] = Re(sum_ky U[xr, ky] e^{+2 pi i ky yy / W}) (natural order, unscaled)
template <int CG>
__device__ __forceinline__ void row_ifft_load(const float* __restrict__ U, int xr, int c, int my, float (&out)[GD_W]) {
  float ur[GD_W], ui[GD_W];
#pragma unroll
  for (int ky = 0; ky < GD_W; ++ky) {
    if (ky < GD_WF) {
      const int kc = ky < my ? ky : my - 1;
      const float a = U[((xr * GD_WF + kc) * 2 + 0) * CG + c], bq = U[((xr * GD_WF + kc) * 2 + 1) * CG + c];
      ur[ky] = ky < my ? a : 0.f;
      ui[ky] = ky < my ? bq : 0.f;
    } else {
      ur[ky] = 0.f;
      ui[ky] = 0.f;
    }
  }
  fft_regs<GD_W, 1>(ur, ui);
  fft_sfor<0, GD_W>([&](auto YY) __attribute__((always_inline)) {
    constexpr int yy = decltype(YY)::value;
    out[yy] = ur[brev<GD_W>(yy)];
  });
}

// ---------------------------------------------------------------------------------------------------------------------
// K1: S = rfft2(GroupNorm(x)) (ortho, kept modes), statistics out.  grid (G, B), 1024 threads, LDS = WF*H*2*CG floats
// ---------------------------------------------------------------------------------------------------------------------
template <int CG>
__global__ __launch_bounds__(GD_T) void gn_rfft2_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ spec,
                                                        float* __restrict__ mean, float* __restrict__ rstd, int E, int G,
                                                        int nb, int mx, int my, float eps, float scale) {
  constexpr int IT = GD_H * CG / GD_T, XS = GD_T / CG;     // row items per thread, row stride between them
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Z = sm;
  __shared__ double shd[48];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int c = tid % CG, xr0 = tid / CG;
  const float* __restrict__ xb = x + (long long)b * GD_H * GD_W * E + g * CG;   // uniform base, 32-bit offsets below
  double s_m = 0.0, s_mm = 0.0, s_q = 0.0;
  const float piv = xb[c];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int xr = xr0 + i * XS;
    float v[GD_W];
#pragma unroll
    for (int y = 0; y < GD_W; ++y) v[y] = xb[(xr * GD_W + y) * E + c];
    float lm, q;
    row_stats(v, lm, q);
    s_m += (double)lm;
    s_mm += (double)lm * (double)lm;
    s_q += (double)q;
    // transform x - p_c (p_c = the channel's first token): the FFT's round-off then scales with the spread of the
    // channel, not with |mean| - a group whose |mean| >> std would otherwise lose log10(|mean| / std) digits in every bin
    // against GroupNorm-then-rfft2.  p_c comes back through the DC term below.
#pragma unroll
    for (int y = 0; y < GD_W; ++y) v[y] -= piv;
    row_fft_store<CG>(v, Z, xr, c);
  }
  block_sum3_d(s_m, s_mm, s_q, shd);                        // (its barriers also publish Z)
  constexpr double NR = (double)(GD_H * CG), NE = (double)(GD_H * GD_W * CG);
  const double mu_d = s_m / NR;
  const double m2 = s_q + (double)GD_W * (s_mm - s_m * s_m / NR);
  const float mu = (float)mu_d;
  const float rs = 1.0f / sqrtf((float)(m2 / NE) + eps);
  if (tid == 0) {
    mean[b * G + g] = mu;
    rstd[b * G + g] = rs;
  }
  const int bs = E / nb;
  for (int it = tid; it < my * CG; it += GD_T) {
    const int c2 = it % CG, ky = it / CG;
    const int ch2 = g * CG + c2;
    const float a = rs * gamma[ch2];
    // GN(x) = a (x - p) + (beta + a (p - mu)): the constant's spectrum is its DC bin; p - mu is formed in double
    const float pm = (float)((double)xb[c2] - mu_d);
    const float dc = ky == 0 ? fmaf(pm, a, beta[ch2]) * (float)(GD_H * GD_W) * scale : 0.f;
    col_fft_out<CG>(Z, spec, b, ch2, c2, ky, nb, bs, mx, my, a * scale, dc);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// K2: y1 = irfft2(spec) + (a1_c x + b1_c)   (the mixer's "+ x_orig", x_orig = GroupNorm1(x) re-derived from x),
//     xn2 = GroupNorm2(y1), statistics out.  colw: Hermitian column weights of the inverse (forward pass: 1)
// ---------------------------------------------------------------------------------------------------------------------
template <int CG>
__global__ __launch_bounds__(GD_T) void irfft2_gn_kernel(const float* __restrict__ spec, const float* __restrict__ x,
                                                         const float* __restrict__ mean1, const float* __restrict__ rstd1,
                                                         const float* __restrict__ gamma1, const float* __restrict__ beta1,
                                                         const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                         float* __restrict__ y1, float* __restrict__ xn2,
                                                         float* __restrict__ mean2, float* __restrict__ rstd2, int E, int G,
                                                         int nb, int mx, int my, int colw, float eps, float scale) {
  constexpr int IT = GD_H * CG / GD_T, XS = GD_T / CG;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* U = sm;                                            // [H][WF][2][CG]
  __shared__ double shd[48];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int bs = E / nb;
  const long long base = (long long)b * GD_H * GD_W * E + g * CG;
  for (int it = tid; it < my * CG; it += GD_T) {
    const int c2 = it % CG, ky = it / CG;
    col_ifft_store<CG>(spec, U, b, g * CG + c2, c2, ky, nb, bs, mx, my, colw_f(colw, ky, GD_W));
  }
  __syncthreads();
  const int c = tid % CG, xr0 = tid / CG;
  const int chn = g * CG + c;
  const float* __restrict__ xs = x + base;
  float* __restrict__ y1s = y1 + base;
  float* __restrict__ xn2s = xn2 + base;
  const float a1 = rstd1[b * G + g] * gamma1[chn];
  const float b1 = beta1[chn] - mean1[b * G + g] * a1;
  float yv[IT][GD_W];
  double s_m = 0.0, s_mm = 0.0, s_q = 0.0;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int xr = xr0 + i * XS;
    row_ifft_load<CG>(U, xr, c, my, yv[i]);
    float q[GD_W];                                          // (loaded AFTER the FFT: 16 fewer live registers inside it)
#pragma unroll
    for (int yy = 0; yy < GD_W; ++yy) q[yy] = xs[(xr * GD_W + yy) * E + c];
#pragma unroll
    for (int yy = 0; yy < GD_W; ++yy) yv[i][yy] = fmaf(yv[i][yy], scale, fmaf(q[yy], a1, b1));
    float lm, qq;
    row_stats(yv[i], lm, qq);
    s_m += (double)lm;
    s_mm += (double)lm * (double)lm;
    s_q += (double)qq;
  }
  block_sum3_d(s_m, s_mm, s_q, shd);
  constexpr double NR = (double)(GD_H * CG), NE = (double)(GD_H * GD_W * CG);
  const double m2 = s_q + (double)GD_W * (s_mm - s_m * s_m / NR);
  const float mu = (float)(s_m / NR);
  const float rs = 1.0f / sqrtf((float)(m2 / NE) + eps);
  if (tid == 0) {
    mean2[b * G + g] = mu;
    rstd2[b * G + g] = rs;
  }
  const float a2 = rs * gamma2[chn], b2 = beta2[chn];
  int co = c;                                               // opaque copy: no 64-bit addresses kept alive across the
  asm volatile("" : "+v"(co));                              // reduction (see irfft2_gn_bwd_kernel)
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int xr = xr0 + i * XS;
#pragma unroll
    for (int yy = 0; yy < GD_W; ++yy) {
      const int o = (xr * GD_W + yy) * E + co;
      y1s[o] = yv[i][yy];
      if (xn2) xn2s[o] = fmaf(yv[i][yy] - mu, a2, b2);       // (NULL: the bf16 channel MLP packs GN2(y1) itself, below)
    }
  }
}

// per-channel sums over the 16 rows of a slab: every row item deposits (sum d*xhat, sum d) of its row in red[2][H][CG],
// thread c < CG adds the 16 rows in a fixed order, writes the parameter-gradient partials and contributes gamma-weighted
// terms to the two group means m1 = mean(gamma d), m2 = mean(gamma d xhat) of the GroupNorm backward
template <int CG, int IT>
__device__ __forceinline__ void gn_bwd_means(float* __restrict__ red, const float (&a_dyx)[IT], const float (&a_dy)[IT],
                                             int xr0, int c, int chn, const float* __restrict__ gamma,
                                             float* __restrict__ part, int B, int b, int E, double* shd, float& m1,
                                             float& m2) {
  constexpr int XS = GD_T / CG;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int xr = xr0 + i * XS;
    red[(0 * GD_H + xr) * CG + c] = a_dyx[i];
    red[(1 * GD_H + xr) * CG + c] = a_dy[i];
  }
  __syncthreads();
  double s1 = 0.0, s2 = 0.0;
  if (threadIdx.x < CG) {                                   // here c == threadIdx.x, chn = its channel
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int r = 0; r < GD_H; ++r) {
      sg += red[(0 * GD_H + r) * CG + c];
      sb += red[(1 * GD_H + r) * CG + c];
    }
    part[((long long)0 * B + b) * E + chn] = sg;
    part[((long long)1 * B + b) * E + chn] = sb;
    const float ga = gamma[chn];
    s1 = (double)ga * sb;
    s2 = (double)ga * sg;
  }
  block_sum2_d(s1, s2, shd);                                // (barriers: red may be overwritten afterwards)
  constexpr double NE = (double)(GD_H * GD_W * CG);
  m1 = (float)(s1 / NE);
  m2 = (float)(s2 / NE);
}

// ---------------------------------------------------------------------------------------------------------------------
// K3: GroupNorm backward (dy -> dx, partials of dgamma / dbeta) + rfft2 of dx with the column weights `colw`
//     (the adjoint of the forward irfft2): dx is written (the mixer's skip path needs it again) AND transformed
// ---------------------------------------------------------------------------------------------------------------------
template <int CG>
__global__ __launch_bounds__(GD_T) void gn_bwd_rfft2_kernel(const float* __restrict__ dy, const float* __restrict__ xin,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float* __restrict__ dx,
                                                            float* __restrict__ part, float* __restrict__ spec, int B,
                                                            int E, int G, int nb, int mx, int my, int colw, float scale) {
  constexpr int IT = GD_H * CG / GD_T, XS = GD_T / CG;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Z = sm;                                            // the reduction scratch red[2][H][CG] aliases its head
  __shared__ double shd[32];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int c = tid % CG, xr0 = tid / CG;
  const int chn = g * CG + c;
  const long long base = (long long)b * GD_H * GD_W * E + g * CG;
  const float* __restrict__ dys = dy + base;
  const float* __restrict__ xis = xin + base;
  float* __restrict__ dxs = dx + base;
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  float d[IT][GD_W], xh[IT][GD_W], a_dyx[IT], a_dy[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int xr = xr0 + i * XS;
    float sd = 0.f, sx = 0.f;
#pragma unroll
    for (int y = 0; y < GD_W; ++y) {
      const int o = (xr * GD_W + y) * E + c;
      d[i][y] = dys[o];
      xh[i][y] = (xis[o] - mu) * rs;
      sd += d[i][y];
      sx = fmaf(d[i][y], xh[i][y], sx);
    }
    a_dy[i] = sd;
    a_dyx[i] = sx;
  }
  float m1, m2;
  gn_bwd_means<CG, IT>(Z, a_dyx, a_dy, xr0, c, chn, gamma, part, B, b, E, shd, m1, m2);
  const float ga = gamma[chn];
  int co = c;                                               // opaque copy (see irfft2_gn_bwd_kernel)
  asm volatile("" : "+v"(co));
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int xr = xr0 + i * XS;
#pragma unroll
    for (int y = 0; y < GD_W; ++y) {
      const float r = rs * (ga * d[i][y] - m1 - xh[i][y] * m2);
      d[i][y] = r;
      dxs[(xr * GD_W + y) * E + co] = r;
    }
    row_fft_store<CG>(d[i], Z, xr, c);
  }
  __syncthreads();
  const int bs = E / nb;
  for (int it = tid; it < my * CG; it += GD_T) {
    const int c2 = it % CG, ky = it / CG;
    col_fft_out<CG>(Z, spec, b, g * CG + c2, c2, ky, nb, bs, mx, my, scale * colw_f(colw, ky, GD_W), 0.f);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// K4: d = irfft2(spec; colw) + res  (adjoint of the forward rfft2 + the mixer's skip path), then GroupNorm backward of
//     d with the saved input x / statistics, + `add` (the Block's outer skip) -> dx, partials of dgamma / dbeta
// ---------------------------------------------------------------------------------------------------------------------
template <int CG>
__global__ __launch_bounds__(GD_T) void irfft2_gn_bwd_kernel(const float* __restrict__ spec, const float* __restrict__ res,
                                                             const float* __restrict__ xin, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ add, float* __restrict__ dx,
                                                             float* __restrict__ part, int B, int E, int G, int nb, int mx,
                                                             int my, int colw, float scale) {
  constexpr int IT = GD_H * CG / GD_T, XS = GD_T / CG;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* U = sm;
  __shared__ double shd[32];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int bs = E / nb;
  const long long base = (long long)b * GD_H * GD_W * E + g * CG;
  for (int it = tid; it < my * CG; it += GD_T) {
    const int c2 = it % CG, ky = it / CG;
    col_ifft_store<CG>(spec, U, b, g * CG + c2, c2, ky, nb, bs, mx, my, colw_f(colw, ky, GD_W));
  }
  __syncthreads();
  const int c = tid % CG, xr0 = tid / CG;
  const int chn = g * CG + c;
  const float* __restrict__ ress = res + base;
  const float* __restrict__ xis = xin + base;
  const float* __restrict__ adds = add ? add + base : nullptr;
  float* __restrict__ dxs = dx + base;
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  // d stays in registers across the reductions; x-hat too with one row item per thread (64 channels per group) - with
  // two items (128 channels) it is re-derived from x in the apply loop (L2-hot) to stay inside 128 registers
  constexpr bool KEEPX = IT == 1;
  float dk[IT][GD_W], xhk[GD_W], a_dyx[IT], a_dy[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int xr = xr0 + i * XS;
    row_ifft_load<CG>(U, xr, c, my, dk[i]);
    float sd = 0.f, sx = 0.f;
#pragma unroll
    for (int y = 0; y < GD_W; ++y) {                        // (operands loaded AFTER the FFT: fewer live registers in it)
      const int o = (xr * GD_W + y) * E + c;
      const float xh = (xis[o] - mu) * rs;
      dk[i][y] = fmaf(dk[i][y], scale, ress[o]);
      sd += dk[i][y];
      sx = fmaf(dk[i][y], xh, sx);
      if (KEEPX) xhk[y] = xh;
    }
    a_dy[i] = sd;
    a_dyx[i] = sx;
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();                                          // every thread is done reading U: its head becomes `red`
  float m1, m2;
  gn_bwd_means<CG, IT>(U, a_dyx, a_dy, xr0, c, chn, gamma, part, B, b, E, shd, m1, m2);
  const float ga = gamma[chn];
  // the apply loop's addresses hang off an OPAQUE copy of the channel index: shared with the loop above, the compiler
  // keeps 3 x 16 x IT 64-bit addresses alive across the reductions and spills them
  int co = c;
  asm volatile("" : "+v"(co));
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int xr = xr0 + i * XS;
#pragma unroll
    for (int y = 0; y < GD_W; ++y) {
      const int o = (xr * GD_W + y) * E + co;
      const float xhv = KEEPX ? xhk[y] : (xis[o] - mu) * rs;
      float r = rs * (ga * dk[i][y] - m1 - xhv * m2);
      if (adds) r += adds[o];
      dxs[o] = r;
    }
  }
}

template <int CG>
static size_t gd_lds() {
  return sizeof(float) * (size_t)GD_WF * GD_H * 2 * CG;
}
template <typename K>
static void gd_attr(K kernel, size_t lds) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

}  // namespace dpot

using namespace dpot;

extern "C" int dpot_gn_dft_supported(int h, int w, int E, int G) {
  static const int enabled = tune("gn_fuse", 1);
  if (!enabled || h != GD_H || w != GD_W || G <= 0 || E <= 0 || E % G) return 0;
  const int cg = E / G;
  return cg == 64 || cg == 128 ? 1 : 0;
}

static int gd_check(const char* what, int B, int h, int w, int E, int G, int nb, int mx, int my) {
  DPOT_REQUIRE(dpot_gn_dft_supported(h, w, E, G), "%s: needs a 16x16 latent grid and 64 or 128 channels per group", what);
  DPOT_REQUIRE(B > 0 && B <= 65535 && nb > 0 && E % nb == 0 && mx >= 1 && mx <= h && my >= 1 && my <= w / 2 + 1,
               "%s: bad argument", what);
  return DPOT_OK;
}

extern "C" int dpot_gn_rfft2(const float* x, const float* gamma, const float* beta, float* spec, float* mean, float* rstd,
                             int B, int h, int w, int E, int G, int nb, int mx, int my, float eps, dpot_stream_t stream) {
  int rc = gd_check("gn_rfft2", B, h, w, E, G, nb, mx, my);
  if (rc) return rc;
  DPOT_REQUIRE(x && gamma && beta && spec && mean && rstd, "gn_rfft2: null pointer");
  const float scale = 1.0f / 16.0f;                           // ortho: 1 / sqrt(16 * 16)
  if (E / G == 64) {
    gd_attr(gn_rfft2_kernel<64>, gd_lds<64>());
    hipLaunchKernelGGL(gn_rfft2_kernel<64>, dim3(G, B), dim3(GD_T), gd_lds<64>(), as_stream(stream), x, gamma, beta, spec,
                       mean, rstd, E, G, nb, mx, my, eps, scale);
  } else {
    gd_attr(gn_rfft2_kernel<128>, gd_lds<128>());
    hipLaunchKernelGGL(gn_rfft2_kernel<128>, dim3(G, B), dim3(GD_T), gd_lds<128>(), as_stream(stream), x, gamma, beta, spec,
                       mean, rstd, E, G, nb, mx, my, eps, scale);
  }
  return check_launch("gn_rfft2_kernel");
}

extern "C" int dpot_irfft2_gn(const float* spec, const float* x, const float* mean1, const float* rstd1,
                              const float* gamma1, const float* beta1, const float* gamma2, const float* beta2, float* y1,
                              float* xn2, float* mean2, float* rstd2, int B, int h, int w, int E, int G, int nb, int mx,
                              int my, int col_weights, float eps, dpot_stream_t stream) {
  int rc = gd_check("irfft2_gn", B, h, w, E, G, nb, mx, my);
  if (rc) return rc;
  DPOT_REQUIRE(spec && x && mean1 && rstd1 && gamma1 && beta1 && gamma2 && beta2 && y1 && mean2 && rstd2,
               "irfft2_gn: null pointer");                  // xn2 may be NULL: statistics only (dpot_bf16_pack_both_norm)
  const float scale = 1.0f / 16.0f;
  if (E / G == 64) {
    gd_attr(irfft2_gn_kernel<64>, gd_lds<64>());
    hipLaunchKernelGGL(irfft2_gn_kernel<64>, dim3(G, B), dim3(GD_T), gd_lds<64>(), as_stream(stream), spec, x, mean1, rstd1,
                       gamma1, beta1, gamma2, beta2, y1, xn2, mean2, rstd2, E, G, nb, mx, my, col_weights, eps, scale);
  } else {
    gd_attr(irfft2_gn_kernel<128>, gd_lds<128>());
    hipLaunchKernelGGL(irfft2_gn_kernel<128>, dim3(G, B), dim3(GD_T), gd_lds<128>(), as_stream(stream), spec, x, mean1,
                       rstd1, gamma1, beta1, gamma2, beta2, y1, xn2, mean2, rstd2, E, G, nb, mx, my, col_weights, eps,
                       scale);
  }
  return check_launch("irfft2_gn_kernel");
}

extern "C" int dpot_gn_bwd_rfft2(const float* dy, const float* xin, const float* mean, const float* rstd,
                                 const float* gamma, float* dx, float* part, float* spec, int B, int h, int w, int E, int G,
                                 int nb, int mx, int my, int col_weights, dpot_stream_t stream) {
  int rc = gd_check("gn_bwd_rfft2", B, h, w, E, G, nb, mx, my);
  if (rc) return rc;
  DPOT_REQUIRE(dy && xin && mean && rstd && gamma && dx && part && spec, "gn_bwd_rfft2: null pointer");
  const float scale = 1.0f / 16.0f;
  if (E / G == 64) {
    gd_attr(gn_bwd_rfft2_kernel<64>, gd_lds<64>());
    hipLaunchKernelGGL(gn_bwd_rfft2_kernel<64>, dim3(G, B), dim3(GD_T), gd_lds<64>(), as_stream(stream), dy, xin, mean, rstd,
                       gamma, dx, part, spec, B, E, G, nb, mx, my, col_weights, scale);
  } else {
    gd_attr(gn_bwd_rfft2_kernel<128>, gd_lds<128>());
    hipLaunchKernelGGL(gn_bwd_rfft2_kernel<128>, dim3(G, B), dim3(GD_T), gd_lds<128>(), as_stream(stream), dy, xin, mean,
                       rstd, gamma, dx, part, spec, B, E, G, nb, mx, my, col_weights, scale);
  }
  return check_launch("gn_bwd_rfft2_kernel");
}

extern "C" int dpot_irfft2_gn_bwd(const float* spec, const float* res, const float* xin, const float* mean,
                                  const float* rstd, const float* gamma, const float* add, float* dx, float* part, int B,
                                  int h, int w, int E, int G, int nb, int mx, int my, int col_weights,
                                  dpot_stream_t stream) {
  int rc = gd_check("irfft2_gn_bwd", B, h, w, E, G, nb, mx, my);
  if (rc) return rc;
  DPOT_REQUIRE(spec && res && xin && mean && rstd && gamma && dx && part, "irfft2_gn_bwd: null pointer");
  const float scale = 1.0f / 16.0f;
  if (E / G == 64) {
    gd_attr(irfft2_gn_bwd_kernel<64>, gd_lds<64>());
    hipLaunchKernelGGL(irfft2_gn_bwd_kernel<64>, dim3(G, B), dim3(GD_T), gd_lds<64>(), as_stream(stream), spec, res, xin,
                       mean, rstd, gamma, add, dx, part, B, E, G, nb, mx, my, col_weights, scale);
  } else {
    gd_attr(irfft2_gn_bwd_kernel<128>, gd_lds<128>());
    hipLaunchKernelGGL(irfft2_gn_bwd_kernel<128>, dim3(G, B), dim3(GD_T), gd_lds<128>(), as_stream(stream), spec, res, xin,
                       mean, rstd, gamma, add, dx, part, B, E, G, nb, mx, my, col_weights, scale);
  }
  return check_launch("irfft2_gn_bwd_kernel");
}
