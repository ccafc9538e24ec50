// dft_fast.h - register-blocked rfft2 / irfft2 for the latent grids DPOT actually uses (16x16: 128^2/patch 8;
// 32x32: 256^2/patch 8).  Same contract as the generic kernels in dft.hip, ~8x fewer instructions:
//   * pass 1 keeps one spatial row of a channel in VGPRs and evaluates the row DFT with compile-time twiddles
//     (fully unrolled -> the twiddles are immediates, products by 0/1 fold away)
//   * the half-complex intermediate crosses to the column pass through LDS, laid out so that both the writes and the
//     reads are conflict free (channel index on the lanes)
//   * HBM traffic stays at the algorithmic minimum: one read of the field, one write of the kept modes.
#pragma once
#include "common.h"

namespace dpot {

template <int N> struct Twid;
template <> struct Twid<16> {
  static __device__ __forceinline__ float c(int i) {
    constexpr float t[16] = {1.0f, 0.9238795042037964f, 0.7071067690849304f, 0.3826834261417389f, 0.0f, -0.3826834261417389f, -0.7071067690849304f, -0.9238795042037964f, -1.0f, -0.9238795042037964f, -0.7071067690849304f, -0.3826834261417389f, 0.0f, 0.3826834261417389f, 0.7071067690849304f, 0.9238795042037964f};
    return t[i];
  }
  static __device__ __forceinline__ float s(int i) {
    constexpr float t[16] = {0.0f, 0.3826834261417389f, 0.7071067690849304f, 0.9238795042037964f, 1.0f, 0.9238795042037964f, 0.7071067690849304f, 0.3826834261417389f, 0.0f, -0.3826834261417389f, -0.7071067690849304f, -0.9238795042037964f, -1.0f, -0.9238795042037964f, -0.7071067690849304f, -0.3826834261417389f};
    return t[i];
  }
};

template <> struct Twid<32> {
  static __device__ __forceinline__ float c(int i) {
    constexpr float t[32] = {1.0f, 0.9807852506637573f, 0.9238795042037964f, 0.8314695954322815f, 0.7071067690849304f, 0.5555702447891235f, 0.3826834261417389f, 0.19509032368659973f, 0.0f, -0.19509032368659973f, -0.3826834261417389f, -0.5555702447891235f, -0.7071067690849304f, -0.8314695954322815f, -0.9238795042037964f, -0.9807852506637573f, -1.0f, -0.9807852506637573f, -0.9238795042037964f, -0.8314695954322815f, -0.7071067690849304f, -0.5555702447891235f, -0.3826834261417389f, -0.19509032368659973f, 0.0f, 0.19509032368659973f, 0.3826834261417389f, 0.5555702447891235f, 0.7071067690849304f, 0.8314695954322815f, 0.9238795042037964f, 0.9807852506637573f};
    return t[i];
  }
  static __device__ __forceinline__ float s(int i) {
    constexpr float t[32] = {0.0f, 0.19509032368659973f, 0.3826834261417389f, 0.5555702447891235f, 0.7071067690849304f, 0.8314695954322815f, 0.9238795042037964f, 0.9807852506637573f, 1.0f, 0.9807852506637573f, 0.9238795042037964f, 0.8314695954322815f, 0.7071067690849304f, 0.5555702447891235f, 0.3826834261417389f, 0.19509032368659973f, 0.0f, -0.19509032368659973f, -0.3826834261417389f, -0.5555702447891235f, -0.7071067690849304f, -0.8314695954322815f, -0.9238795042037964f, -0.9807852506637573f, -1.0f, -0.9807852506637573f, -0.9238795042037964f, -0.8314695954322815f, -0.7071067690849304f, -0.5555702447891235f, -0.3826834261417389f, -0.19509032368659973f};
    return t[i];
  }
};


template <int N>
__device__ __forceinline__ void fill_tw2(float2* tw2) {
  for (int t = threadIdx.x; t < N; t += blockDim.x) {
    const double a = (double)(2 * t) / (double)N;
    tw2[t] = make_float2((float)cospi(a), (float)sinpi(a));
  }
}

__device__ __forceinline__ float colw_f(int colw, int ky, int w) {
  if (!colw || ky == 0 || ((w & 1) == 0 && ky == (w >> 1))) return 1.f;
  return 2.f;
}

// x[B,H*W,E] -> spec[B,mx,my,nb,2,bs]
template <int H, int W, int CC>
__global__ __launch_bounds__(256) void rfft2_fast_kernel(const float* __restrict__ x, float* __restrict__ spec, int E,
                                                         int nb, int mx, int my, int colw, float scale) {
  constexpr int WF = W / 2 + 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Z = sm;                                            // [WF][H][2][CC]
  float2* tw2 = reinterpret_cast<float2*>(sm + WF * H * 2 * CC);  // [H] (runtime-loop variant only)
  const int b = blockIdx.y, c0 = blockIdx.x * CC;
  const int tid = threadIdx.x;
  const int bs = E / nb;
  if constexpr (H > 16) fill_tw2<H>(tw2);
  const float* xb = x + (long long)b * H * W * E + c0;

  // pass 1: rows (real -> half complex)
  for (int it = tid; it < H * CC; it += 256) {
    const int c = it % CC, xr = it / CC;
    float v[W];
#pragma unroll
    for (int y = 0; y < W; ++y) v[y] = xb[(long long)(xr * W + y) * E + c];
#pragma unroll
    for (int ky = 0; ky < WF; ++ky) {
      float re = 0.f, im = 0.f;
#pragma unroll
      for (int y = 0; y < W; ++y) {
        const int idx = (ky * y) % W;
        re = fmaf(v[y], Twid<W>::c(idx), re);
        im = fmaf(-v[y], Twid<W>::s(idx), im);
      }
      Z[((ky * H + xr) * 2 + 0) * CC + c] = re;
      Z[((ky * H + xr) * 2 + 1) * CC + c] = im;
    }
  }
  __syncthreads();

  // pass 2: columns (complex -> complex), kept modes only
  for (int it = tid; it < my * CC; it += 256) {
    const int c = it % CC, ky = it / CC;
    float zr[H], zi[H];
#pragma unroll
    for (int xr = 0; xr < H; ++xr) {
      zr[xr] = Z[((ky * H + xr) * 2 + 0) * CC + c];
      zi[xr] = Z[((ky * H + xr) * 2 + 1) * CC + c];
    }
    const int chn = c0 + c;
    const int blk = chn / bs, ci = chn % bs;
    const float wgt = scale * colw_f(colw, ky, W);
    float* out = spec + (((long long)b * mx * my + ky) * nb + blk) * 2 * bs + ci;
    const long long kxstride = (long long)my * nb * 2 * bs;
    if constexpr (H <= 16) {
#pragma unroll
      for (int kx = 0; kx < H; ++kx) {
        if (kx < mx) {
          float re = 0.f, im = 0.f;
#pragma unroll
          for (int xr = 0; xr < H; ++xr) {
            const int idx = (kx * xr) % H;
            const float cc_ = Twid<H>::c(idx), ss_ = Twid<H>::s(idx);
            re = fmaf(zr[xr], cc_, fmaf(zi[xr], ss_, re));
            im = fmaf(zi[xr], cc_, fmaf(-zr[xr], ss_, im));
          }
          out[kx * kxstride] = re * wgt;
          out[kx * kxstride + bs] = im * wgt;
        }
      }
    } else {
#pragma unroll 1
      for (int kx = 0; kx < mx; ++kx) {
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int xr = 0; xr < H; ++xr) {
          const float2 t = tw2[(kx * xr) & (H - 1)];
          re = fmaf(zr[xr], t.x, fmaf(zi[xr], t.y, re));
          im = fmaf(zi[xr], t.x, fmaf(-zr[xr], t.y, im));
        }
        out[kx * kxstride] = re * wgt;
        out[kx * kxstride + bs] = im * wgt;
      }
    }
  }
}

// spec[B,mx,my,nb,2,bs] (+ res[B,H*W,E]) -> y[B,H*W,E]
template <int H, int W, int CC>
__global__ __launch_bounds__(256) void irfft2_fast_kernel(const float* __restrict__ spec, const float* __restrict__ res,
                                                          float* __restrict__ y, int E, int nb, int mx, int my,
                                                          int colw, float scale) {
  constexpr int WF = W / 2 + 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* U = sm;                                            // [H][WF][2][CC]
  float2* tw2 = reinterpret_cast<float2*>(sm + WF * H * 2 * CC);
  const int b = blockIdx.y, c0 = blockIdx.x * CC;
  const int tid = threadIdx.x;
  const int bs = E / nb;
  if constexpr (H > 16) {
    fill_tw2<H>(tw2);
    __syncthreads();
  }

  // pass A: columns, U[x,ky] = sum_kx S[kx,ky] e^{+2 pi i kx x / H}, times the column weight
  for (int it = tid; it < my * CC; it += 256) {
    const int c = it % CC, ky = it / CC;
    const int chn = c0 + c;
    const int blk = chn / bs, ci = chn % bs;
    const float* in = spec + (((long long)b * mx * my + ky) * nb + blk) * 2 * bs + ci;
    const long long kxstride = (long long)my * nb * 2 * bs;
    float sr[H], si[H];
#pragma unroll
    for (int kx = 0; kx < H; ++kx) {
      const int kc = kx < mx ? kx : mx - 1;                 // clamped address, selected value: no divergent loads
      const float a = in[kc * kxstride], bq = in[kc * kxstride + bs];
      sr[kx] = kx < mx ? a : 0.f;
      si[kx] = kx < mx ? bq : 0.f;
    }
    const float wgt = colw_f(colw, ky, W);
    if constexpr (H <= 16) {
#pragma unroll
      for (int xr = 0; xr < H; ++xr) {
        float ur = 0.f, ui = 0.f;
#pragma unroll
        for (int kx = 0; kx < H; ++kx) {
          const int idx = (kx * xr) % H;
          const float cc_ = Twid<H>::c(idx), ss_ = Twid<H>::s(idx);
          ur = fmaf(sr[kx], cc_, fmaf(-si[kx], ss_, ur));
          ui = fmaf(sr[kx], ss_, fmaf(si[kx], cc_, ui));
        }
        U[((xr * WF + ky) * 2 + 0) * CC + c] = ur * wgt;
        U[((xr * WF + ky) * 2 + 1) * CC + c] = ui * wgt;
      }
    } else {
#pragma unroll 1
      for (int xr = 0; xr < H; ++xr) {
        float ur = 0.f, ui = 0.f;
#pragma unroll
        for (int kx = 0; kx < H; ++kx) {
          const float2 t = tw2[(kx * xr) & (H - 1)];
          ur = fmaf(sr[kx], t.x, fmaf(-si[kx], t.y, ur));
          ui = fmaf(sr[kx], t.y, fmaf(si[kx], t.x, ui));
        }
        U[((xr * WF + ky) * 2 + 0) * CC + c] = ur * wgt;
        U[((xr * WF + ky) * 2 + 1) * CC + c] = ui * wgt;
      }
    }
  }
  __syncthreads();

  // pass B: rows (half complex -> real), + residual
  const long long base = (long long)b * H * W * E + c0;
  for (int it = tid; it < H * CC; it += 256) {
    const int c = it % CC, xr = it / CC;
    float ur[WF], ui[WF];
#pragma unroll
    for (int ky = 0; ky < WF; ++ky) {
      const int kc = ky < my ? ky : my - 1;
      const float a = U[((xr * WF + kc) * 2 + 0) * CC + c], bq = U[((xr * WF + kc) * 2 + 1) * CC + c];
      ur[ky] = ky < my ? a : 0.f;
      ui[ky] = ky < my ? bq : 0.f;
    }
    float q[W];
    if (res) {
#pragma unroll
      for (int yy = 0; yy < W; ++yy) q[yy] = res[base + (long long)(xr * W + yy) * E + c];
    }
#pragma unroll
    for (int yy = 0; yy < W; ++yy) {
      float acc = 0.f;
#pragma unroll
      for (int ky = 0; ky < WF; ++ky) {
        const int idx = (ky * yy) % W;
        acc = fmaf(ur[ky], Twid<W>::c(idx), fmaf(-ui[ky], Twid<W>::s(idx), acc));
      }
      float v = acc * scale;
      if (res) v += q[yy];
      y[base + (long long)(xr * W + yy) * E + c] = v;
    }
  }
}

template <int H, int W, int CC>
static int launch_rfft2_fast(const float* x, float* spec, int B, int E, int nb, int mx, int my, int colw, float scale,
                             hipStream_t s) {
  constexpr size_t lds = sizeof(float) * ((size_t)(W / 2 + 1) * H * 2 * CC + 2 * H);
  hipFuncSetAttribute(reinterpret_cast<const void*>(rfft2_fast_kernel<H, W, CC>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((rfft2_fast_kernel<H, W, CC>), dim3(E / CC, B), dim3(256), lds, s, x, spec, E, nb, mx, my, colw,
                     scale);
  return check_launch("rfft2_fast_kernel");
}
template <int H, int W, int CC>
static int launch_irfft2_fast(const float* spec, const float* res, float* y, int B, int E, int nb, int mx, int my,
                              int colw, float scale, hipStream_t s) {
  constexpr size_t lds = sizeof(float) * ((size_t)(W / 2 + 1) * H * 2 * CC + 2 * H);
  hipFuncSetAttribute(reinterpret_cast<const void*>(irfft2_fast_kernel<H, W, CC>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((irfft2_fast_kernel<H, W, CC>), dim3(E / CC, B), dim3(256), lds, s, spec, res, y, E, nb, mx, my,
                     colw, scale);
  return check_launch("irfft2_fast_kernel");
}

// returns 1 if a fast kernel was launched (rc in *rc), 0 if the shape has no fast path
static inline int try_rfft2_fast(const float* x, float* spec, int B, int h, int w, int E, int nb, int mx, int my,
                                 int colw, float scale, hipStream_t s, int* rc) {
  if (h == 16 && w == 16) {
    // 64-channel slabs only when that still gives every CU >= 2 workgroups (latency hiding); else 32-channel slabs
    if (E % 64 == 0 && (long long)B * (E / 64) >= 512) { *rc = launch_rfft2_fast<16, 16, 64>(x, spec, B, E, nb, mx, my, colw, scale, s); return 1; }
    if (E % 32 == 0) { *rc = launch_rfft2_fast<16, 16, 32>(x, spec, B, E, nb, mx, my, colw, scale, s); return 1; }
  } else if (h == 32 && w == 32) {
    if (E % 16 == 0) { *rc = launch_rfft2_fast<32, 32, 16>(x, spec, B, E, nb, mx, my, colw, scale, s); return 1; }
  }
  return 0;
}
static inline int try_irfft2_fast(const float* spec, const float* res, float* y, int B, int h, int w, int E, int nb,
                                  int mx, int my, int colw, float scale, hipStream_t s, int* rc) {
  if (h == 16 && w == 16) {
    if (E % 64 == 0 && (long long)B * (E / 64) >= 512) { *rc = launch_irfft2_fast<16, 16, 64>(spec, res, y, B, E, nb, mx, my, colw, scale, s); return 1; }
    if (E % 32 == 0) { *rc = launch_irfft2_fast<16, 16, 32>(spec, res, y, B, E, nb, mx, my, colw, scale, s); return 1; }
  } else if (h == 32 && w == 32) {
    if (E % 16 == 0) { *rc = launch_irfft2_fast<32, 32, 16>(spec, res, y, B, E, nb, mx, my, colw, scale, s); return 1; }
  }
  return 0;
}

}  // namespace dpot
