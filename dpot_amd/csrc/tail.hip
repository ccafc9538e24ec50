// tail.hip - the per-pixel tail of the DPOT out layer, fused (models/dpot.py:315-321 after the ConvTranspose):
//
//     u = act(Upre)            Upre[pixels, 32] : ConvTranspose2d(k=s=P) output incl. bias, pixel-major
//     v = act(W2 u + b2)       Conv2d(32 -> 32, 1x1)
//     z = W4 v + b4            Conv2d(32 -> co, 1x1), co = out_channels * out_timesteps <= 32
//     out[b, px*P+i, py*P+j, :] = z        (pixel shuffle folded into the store)
//
// As separate GEMMs this costs five passes over [pixels, 32] matrices forward (67 MB each at DPOT-Tiny B=32) and about
// ten backward; fused, the forward reads Upre once and the backward reads Upre + dOut and writes dUpre once -
// the algorithmic minimum.  One wave owns a tile of 32 pixels and keeps everything in registers:
//
//   * the 32x32 mat-vecs run on the matrix cores in the TRANSPOSED form  V'[o2,pix] = sum_o W2[o2,o] u[o,pix]
//     (v_mfma_f32_32x32x2_f32, A = weights, B = activations): the accumulator layout of one product
//     (col = pix = lane&31, row = perm(r, lane>>5)) IS the B-operand layout of the next, so the chain
//     Upre -> v -> z (and dz -> dv -> du backward) needs no cross-lane data movement at all;
//   * a tile's 4 KiB of Upre is fetched with four fully coalesced 1-KiB wave loads (software-prefetched one tile
//     ahead) and re-dealt through a per-wave LDS slab into "one pixel per lane, 16 channels perm(r,kh) per lane";
//     (loading the operand pattern directly touches 32 cache lines per instruction at 25 % use and thrashes L1);
//     dUpre leaves the same way in reverse;
//   * weight gradients (dW2 = dvpre u^T, dW4 = dz v^T, sums over pixels) are MFMA products with the PIXEL index as the
//     contraction: the operands are transposed through LDS slabs; each wave accumulates over all its tiles
//     and writes one partial row, reduced afterwards by dpot_colsum (fixed order -> deterministic).
//
// Fast path for out_layer_dim == 32 (DPOT-Ti/S/M); other widths use the generic GEMM chain (functional.HeadFn).
#include "common.h"

namespace dpot {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TD = 32;                   // out_layer_dim handled here
constexpr int TLD = 36;                  // LDS row stride (floats) of the slabs
constexpr int SLAB = TD * TLD;           // floats per slab
constexpr int TAIL_PCOLS = 2144;         // partial row: dW2[1024] dW4[1024] db2[32] db0[32] db4[32]

__device__ __forceinline__ int permk(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

// value and derivative of the activation at x (GELU shares the exponential)
__device__ __forceinline__ void act_val_der(int act, float x, float& val, float& der) {
  if (act == DPOT_ACT_GELU) {
    float cdf, g;
    gelu_parts(x, cdf, g);
    val = x * cdf;
    der = fmaf(x * 0.39894228040143267794f, g, cdf);
  } else {
    val = act_fwd(act, x);
    der = act_bwd(act, x);
  }
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct TailGeom {
  int h, w, P, co;
};
// pixel-major index p = ((b*h + px)*w + py)*P*P + i*P + j  ->  image pixel index (b, px*P+i, py*P+j)
// (32-bit arithmetic: 64-bit integer division costs ~150 VALU instructions on gfx950; pixel counts fit easily)
__device__ __forceinline__ unsigned image_pixel(const TailGeom& g, unsigned p) {
  const unsigned P = (unsigned)g.P, PP = P * P;
  const unsigned m = p / PP;
  const unsigned ij = p - m * PP;
  const unsigned i = ij / P, j = ij - i * P;
  const unsigned t = m / (unsigned)g.w;
  const unsigned py = m - t * (unsigned)g.w;
  const unsigned b = t / (unsigned)g.h;
  const unsigned px = t - b * (unsigned)g.h;
  return ((b * g.h + px) * P + i) * ((unsigned)g.w * P) + py * P + j;
}

// coalesced fetch of one tile (32 pixels x 32 channels = 1024 floats): 4 wave-wide float4 loads into g0..g3.
// (named registers + macros on purpose: a float4[4] handed to a helper by reference across the tile loop ends up
// in scratch memory)
#define TILE_FETCH(base, tile_)                                                        \
  do {                                                                                 \
    const float* t_ = (base) + (tile_) * (32 * TD) + lane * 4;                         \
    g0 = *reinterpret_cast<const float4*>(t_);                                         \
    g1 = *reinterpret_cast<const float4*>(t_ + 256);                                   \
    g2 = *reinterpret_cast<const float4*>(t_ + 512);                                   \
    g3 = *reinterpret_cast<const float4*>(t_ + 768);                                   \
  } while (0)
// re-deal through LDS: flat element e = 256k + 4*lane -> pixel e/32 = 8k + lane/8, channel 4*(lane%8);
// afterwards lane (li, kh) holds x[4k+e] = Upre[pixel li][channel perm(4k+e, kh)]
#define TILE_TO_OPERAND(XARR)                                                          \
  do {                                                                                 \
    float* w_ = &stg[(lane >> 3) * TLD + 4 * (lane & 7)];                              \
    *reinterpret_cast<float4*>(w_) = g0;                                               \
    *reinterpret_cast<float4*>(w_ + 8 * TLD) = g1;                                     \
    *reinterpret_cast<float4*>(w_ + 16 * TLD) = g2;                                    \
    *reinterpret_cast<float4*>(w_ + 24 * TLD) = g3;                                    \
    wave_lds_sync();                                                                   \
    const float* r_ = &stg[li * TLD + 4 * kh];                                         \
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) {                                 \
      const float4 t4_ = *reinterpret_cast<const float4*>(r_ + 8 * k_);                \
      XARR[4 * k_] = t4_.x; XARR[4 * k_ + 1] = t4_.y; XARR[4 * k_ + 2] = t4_.z; XARR[4 * k_ + 3] = t4_.w; \
    }                                                                                  \
    __builtin_amdgcn_wave_barrier();                                                   \
  } while (0)

template <bool GELU>
__device__ __forceinline__ float actf(int act, float x) {
  if constexpr (GELU) return gelu_fwd(x);
  else return act_fwd(act, x);
}
template <bool GELU>
__device__ __forceinline__ void actvd(int act, float x, float& val, float& der) {
  if constexpr (GELU) {
    float cdf, g;
    gelu_parts(x, cdf, g);
    val = x * cdf;
    der = fmaf(x * 0.39894228040143267794f, g, cdf);
  } else {
    val = act_fwd(act, x);
    der = act_bwd(act, x);
  }
}

// w4p: W4 zero-padded to [32][32], b4p: b4 zero-padded to [32] (so that every weight load is unconditional)
template <bool GELU>
__global__ __launch_bounds__(256) void out_tail_fwd_kernel(const float* __restrict__ upre, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, const float* __restrict__ w4p,
                                                           const float* __restrict__ b4p, float* __restrict__ out,
                                                           long long ntiles, TailGeom g, int act) {
  __shared__ __attribute__((aligned(16))) float sm[4 * SLAB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, kh = lane >> 5;
  float* stg = sm + wave * SLAB;
  float a2[16], a4[16], bb2[16], bb4[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = permk(r, kh);
    a2[r] = w2[li * TD + o];
    a4[r] = w4p[li * TD + o];
    bb2[r] = b2[o];
    bb4[r] = b4p[o];
  }
  const long long stride = (long long)gridDim.x * 4;
  long long tile = (long long)blockIdx.x * 4 + wave;
  float4 g0, g1, g2, g3;   // software prefetch: the next tile is in flight while this one is computed (clamped index)
  TILE_FETCH(upre, tile < ntiles ? tile : ntiles - 1);
  for (; tile < ntiles; tile += stride) {
    float x[16];
    TILE_TO_OPERAND(x);
    TILE_FETCH(upre, tile + stride < ntiles ? tile + stride : ntiles - 1);
    float u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r] = actf<GELU>(act, x[r]);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[r], u[r], acc, 0, 0, 0);
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = actf<GELU>(act, acc[r] + bb2[r]);
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) z = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[r], v[r], z, 0, 0, 0);
    float* dst = out + (long long)image_pixel(g, (unsigned)(tile * 32 + li)) * g.co;
    if (g.co == 4) {
      if (kh == 0)
        *reinterpret_cast<float4*>(dst) = make_float4(z[0] + bb4[0], z[1] + bb4[1], z[2] + bb4[2], z[3] + bb4[3]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = permk(r, kh);
        if (c < g.co) dst[c] = z[r] + bb4[r];
      }
    }
  }
}

template <bool GELU>
__global__ __launch_bounds__(256) void out_tail_bwd_kernel(const float* __restrict__ upre, const float* __restrict__ dout,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           const float* __restrict__ w4p, float* __restrict__ dupre,
                                                           float* __restrict__ partials, long long ntiles, TailGeom g,
                                                           int act) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, kh = lane >> 5;
  float* Xd = sm + wave * (5 * SLAB);      // dvpre [o2][pix]
  float* Xu = Xd + SLAB;                   // u     [o ][pix]
  float* Xv = Xu + SLAB;                   // v     [o2][pix]
  float* Xz = Xv + SLAB;                   // dz    [c ][pix]   (rows >= co stay zero)
  float* stg = Xz + SLAB;                  // [pix][ch] staging for the coalesced Upre load / dUpre store
  for (int idx = lane; idx < SLAB; idx += 64) Xz[idx] = 0.f;

  const int R4 = 4 * ((g.co + 7) / 8);     // MFMA k-steps that can carry a channel c < co
  float a2[16], a2t[16], a4t[16], bb2[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = permk(r, kh);
    a2[r] = w2[li * TD + o];               // A[i=o2=li][k=o]      : W2
    a2t[r] = w2[o * TD + li];              // A[i=o =li][k=o2]     : W2^T
    a4t[r] = w4p[o * TD + li];             // A[i=o2=li][k=c]      : W4^T (rows >= co are zero)
    bb2[r] = b2[o];
  }
  f32x16 gW2, gW4;
  float gb2[16], gb0[16], gb4[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) gW2[r] = gW4[r] = gb2[r] = gb0[r] = gb4[r] = 0.f;

  const long long stride = (long long)gridDim.x * 4;
  long long tile = (long long)blockIdx.x * 4 + wave;
  float4 g0, g1, g2, g3;
  TILE_FETCH(upre, tile < ntiles ? tile : ntiles - 1);
  for (; tile < ntiles; tile += stride) {
    float xs[16];
    TILE_TO_OPERAND(xs);
    TILE_FETCH(upre, tile + stride < ntiles ? tile + stride : ntiles - 1);
    // dz in B-operand layout: dz[r] = dOut[pixel][c = perm(r,kh)]
    const float* src = dout + (long long)image_pixel(g, (unsigned)(tile * 32 + li)) * g.co;
    float dz[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dz[r] = 0.f;
    if (g.co == 4) {
      const float4 t = *reinterpret_cast<const float4*>(src);   // every lane loads (no divergent load); kh=1 discards
      if (kh == 0) { dz[0] = t.x; dz[1] = t.y; dz[2] = t.z; dz[3] = t.w; }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = permk(r, kh);
        const float t = src[c < g.co ? c : 0];
        dz[r] = c < g.co ? t : 0.f;
      }
    }
    float u[16], du_act[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) actvd<GELU>(act, xs[r], u[r], du_act[r]);
    // recompute v
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[r], u[r], acc, 0, 0, 0);
    float v[16], dvpre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) actvd<GELU>(act, acc[r] + bb2[r], v[r], dvpre[r]);   // dvpre holds act'(vpre) for now
    // dv'[o2,pix] = sum_c W4[c,o2] dz[c,pix]
    f32x16 dv;
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (r < R4) dv = __builtin_amdgcn_mfma_f32_32x32x2f32(a4t[r], dz[r], dv, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) dvpre[r] *= dv[r];
    // du'[o,pix] = sum_o2 W2[o2,o] dvpre[o2,pix]
    f32x16 du;
#pragma unroll
    for (int r = 0; r < 16; ++r) du[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) du = __builtin_amdgcn_mfma_f32_32x32x2f32(a2t[r], dvpre[r], du, 0, 0, 0);
    float dup[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dup[r] = du[r] * du_act[r];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gb2[r] += dvpre[r];
      gb0[r] += dup[r];
      gb4[r] += dz[r];
    }
    // ---- LDS: dUpre back to the coalesced [pix][ch] deal; operands of the weight-gradient products transposed so
    //      that the pixel index becomes the MFMA k index
#pragma unroll
    for (int k = 0; k < 4; ++k)
      *reinterpret_cast<float4*>(&stg[li * TLD + 8 * k + 4 * kh]) =
          make_float4(dup[4 * k], dup[4 * k + 1], dup[4 * k + 2], dup[4 * k + 3]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ch = permk(r, kh);
      Xd[ch * TLD + li] = dvpre[r];
      Xu[ch * TLD + li] = u[r];
      Xv[ch * TLD + li] = v[r];
      if (r < R4) Xz[ch * TLD + li] = dz[r];
    }
    wave_lds_sync();
    {
      float* drow = dupre + tile * (32 * TD) + lane * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<float4*>(drow + 256 * k) =
            *reinterpret_cast<const float4*>(&stg[(8 * k + (lane >> 3)) * TLD + 4 * (lane & 7)]);
    }
    float xd[16], xu[16], xv[16], xz[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int off = li * TLD + 16 * kh + 4 * k;
      const float4 td = *reinterpret_cast<const float4*>(&Xd[off]);
      const float4 tu = *reinterpret_cast<const float4*>(&Xu[off]);
      const float4 tv = *reinterpret_cast<const float4*>(&Xv[off]);
      const float4 tz = *reinterpret_cast<const float4*>(&Xz[off]);
      xd[4 * k] = td.x; xd[4 * k + 1] = td.y; xd[4 * k + 2] = td.z; xd[4 * k + 3] = td.w;
      xu[4 * k] = tu.x; xu[4 * k + 1] = tu.y; xu[4 * k + 2] = tu.z; xu[4 * k + 3] = tu.w;
      xv[4 * k] = tv.x; xv[4 * k + 1] = tv.y; xv[4 * k + 2] = tv.z; xv[4 * k + 3] = tv.w;
      xz[4 * k] = tz.x; xz[4 * k + 1] = tz.y; xz[4 * k + 2] = tz.z; xz[4 * k + 3] = tz.w;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      gW2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xd[s], xu[s], gW2, 0, 0, 0);   // dW2[o2][o] += dvpre[o2,pix] u[o,pix]
      gW4 = __builtin_amdgcn_mfma_f32_32x32x2f32(xz[s], xv[s], gW4, 0, 0, 0);   // dW4[c][o2]  += dz[c,pix] v[o2,pix]
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- one partial row per wave
  float* prow = partials + ((long long)blockIdx.x * 4 + wave) * TAIL_PCOLS;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rowi = permk(r, kh);
    prow[rowi * TD + li] = gW2[r];
    prow[1024 + rowi * TD + li] = gW4[r];
    float s2 = gb2[r], s0 = gb0[r], s4 = gb4[r];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {   // reduce over the 32 pixel lanes of this half-wave
      s2 += __shfl_xor(s2, off, 64);
      s0 += __shfl_xor(s0, off, 64);
      s4 += __shfl_xor(s4, off, 64);
    }
    if (li == 0) {
      prow[2048 + rowi] = s2;
      prow[2080 + rowi] = s0;
      prow[2112 + rowi] = s4;
    }
  }
}

static int tail_grid_fwd(long long ntiles) {
  long long g = (ntiles + 3) / 4;
  if (g > 768) g = 768;   // 3 resident workgroups per CU; each wave walks several tiles (amortises the weight preload)
  return (int)(g < 1 ? 1 : g);
}
static int tail_grid_bwd(long long ntiles) {
  long long g = (ntiles + 3) / 4;
  if (g > 256) g = 256;   // 90 KiB of LDS and ~370 registers per lane: one workgroup per CU
  return (int)(g < 1 ? 1 : g);
}

}  // namespace dpot

using namespace dpot;

static int tail_check(const char* who, int B, int h, int w, int P, int co) {
  DPOT_REQUIRE(B > 0 && h > 0 && w > 0 && P > 0 && co > 0 && co <= 32, "%s: bad shape", who);
  DPOT_REQUIRE(((long long)B * h * w * P * P) % 32 == 0, "%s: pixel count must be a multiple of 32", who);
  DPOT_REQUIRE((long long)B * h * w * P * P * 32 < (1ll << 31), "%s: too many pixels for 32-bit indexing", who);
  return DPOT_OK;
}

extern "C" int dpot_out_tail_partial_rows(int B, int h, int w, int P) {
  return tail_grid_bwd((long long)B * h * w * P * P / 32) * 4;
}
extern "C" int dpot_out_tail_partial_cols(void) { return TAIL_PCOLS; }

extern "C" int dpot_out_tail_fwd(const float* upre, const float* w2, const float* b2, const float* w4, const float* b4,
                                 float* out, int B, int h, int w, int P, int co, int act, dpot_stream_t stream) {
  int rc = tail_check("out_tail_fwd", B, h, w, P, co);
  if (rc) return rc;
  DPOT_REQUIRE(upre && w2 && b2 && w4 && b4 && out && aligned16(upre) && aligned16(out), "out_tail_fwd: bad pointer");
  const long long ntiles = (long long)B * h * w * P * P / 32;
  TailGeom g{h, w, P, co};
  if (act == DPOT_ACT_GELU)
    hipLaunchKernelGGL(out_tail_fwd_kernel<true>, dim3(tail_grid_fwd(ntiles)), dim3(256), 0, as_stream(stream), upre, w2,
                       b2, w4, b4, out, ntiles, g, act);
  else
    hipLaunchKernelGGL(out_tail_fwd_kernel<false>, dim3(tail_grid_fwd(ntiles)), dim3(256), 0, as_stream(stream), upre, w2,
                       b2, w4, b4, out, ntiles, g, act);
  return check_launch("out_tail_fwd_kernel");
}

extern "C" int dpot_out_tail_bwd(const float* upre, const float* dout, const float* w2, const float* b2, const float* w4,
                                 float* dupre, float* partials, int B, int h, int w, int P, int co, int act,
                                 dpot_stream_t stream) {
  int rc = tail_check("out_tail_bwd", B, h, w, P, co);
  if (rc) return rc;
  DPOT_REQUIRE(upre && dout && w2 && b2 && w4 && dupre && partials && aligned16(upre) && aligned16(dupre) &&
                   aligned16(dout),
               "out_tail_bwd: bad pointer");
  const long long ntiles = (long long)B * h * w * P * P / 32;
  TailGeom g{h, w, P, co};
  const size_t lds = sizeof(float) * 4 * 5 * SLAB;
  const int grid = tail_grid_bwd(ntiles);
  if (act == DPOT_ACT_GELU) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(out_tail_bwd_kernel<true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(out_tail_bwd_kernel<true>, dim3(grid), dim3(256), lds, as_stream(stream), upre, dout, w2, b2, w4,
                       dupre, partials, ntiles, g, act);
  } else {
    hipFuncSetAttribute(reinterpret_cast<const void*>(out_tail_bwd_kernel<false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(out_tail_bwd_kernel<false>, dim3(grid), dim3(256), lds, as_stream(stream), upre, dout, w2, b2, w4,
                       dupre, partials, ntiles, g, act);
  }
  return check_launch("out_tail_bwd_kernel");
}
