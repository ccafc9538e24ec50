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
//     the four waves of a workgroup sum their rows through LDS and write one partial row, reduced afterwards by
//     dpot_colsum (fixed order -> deterministic).
//
// Fast path for out_layer_dim == 32 (DPOT-Ti/S/M); other widths use the generic GEMM chain (functional.HeadFn).
#include "common.h"

namespace dpot {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TD = 32;                   // out_layer_dim handled here
constexpr int TLD = 36;                  // LDS row stride (floats) of the slabs
constexpr int SLAB = TD * TLD;           // floats per slab
constexpr int TAIL_PCOLS = 2144;         // partial row: dW2[1024] dW4[1024] db2[32] db0[32] db4[32]

__device__ __forceinline__ int permk(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct TailGeom {
  int h, w, P, co;
  unsigned mPP, mP, mw, mh;   // floor(2^32 / d) + 1 for d = P*P, P, w, h (0: d == 1)
};
// n / d by the reciprocal: exact while n * d < 2^32 (host-checked; a 32-bit unsigned division costs ~30 VALU
// instructions on gfx950 and this kernel is VALU-issue bound)
__device__ __forceinline__ unsigned mdiv(unsigned n, unsigned m) { return m ? __umulhi(n, m) : n; }
// pixel-major index p = ((b*h + px)*w + py)*P*P + i*P + j  ->  image pixel index (b, px*P+i, py*P+j)
__device__ __forceinline__ unsigned image_pixel(const TailGeom& g, unsigned p) {
  const unsigned P = (unsigned)g.P, PP = P * P;
  const unsigned m = mdiv(p, g.mPP);
  const unsigned ij = p - m * PP;
  const unsigned i = mdiv(ij, g.mP), j = ij - i * P;
  const unsigned t = mdiv(m, g.mw);
  const unsigned py = m - t * (unsigned)g.w;
  const unsigned b = mdiv(t, g.mh);
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
__device__ __forceinline__ void act16(int act, const float (&x)[16], float (&u)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) u[r] = GELU ? gelu_fwd(x[r]) : act_fwd(act, x[r]);
}
template <bool GELU>
__device__ __forceinline__ void actvd16(int act, const float (&x)[16], float (&val)[16], float (&der)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if constexpr (GELU) {
      gelu_val_der(x[r], val[r], der[r]);
    } else {
      val[r] = act_fwd(act, x[r]);
      der[r] = act_bwd(act, x[r]);
    }
  }
}

// w4p: W4 zero-padded to [32][32], b4p: b4 zero-padded to [32] (so that every weight load is unconditional).
// CO4 (co <= 4, every DPOT config with out_channels * out_timesteps <= 4): the 32 -> co product runs on
// v_mfma_f32_4x4x1_16b_f32 - 16 blocks of 4x4, block = 4 consecutive lanes, A row i = lane & 3 = output channel,
// B column j = lane & 3 = pixel within the block (so B IS "one pixel per lane": v[r] as it stands), one k-step =
// one input channel perm(r, kh): 16 instructions of 8 cycles instead of 16 of 64 with 28 of the 32 output rows
// unused.  Lanes (li, kh = 0 / 1) end up with the partial sums over their half of the input channels.
template <bool GELU, bool CO4>
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
    a4[r] = w4p[(CO4 ? (lane & 3) : li) * TD + o];
    bb2[r] = b2[o];
    bb4[r] = CO4 ? 0.f : b4p[o];
  }
  const float4 b4v = *reinterpret_cast<const float4*>(b4p);
  const long long stride = (long long)gridDim.x * 4;
  long long tile = (long long)blockIdx.x * 4 + wave;
  float4 g0, g1, g2, g3;   // software prefetch: the next tile is in flight while this one is computed (clamped index)
  TILE_FETCH(upre, tile < ntiles ? tile : ntiles - 1);
  for (; tile < ntiles; tile += stride) {
    float x[16];
    TILE_TO_OPERAND(x);
    TILE_FETCH(upre, tile + stride < ntiles ? tile + stride : ntiles - 1);
    float u[16];
    act16<GELU>(act, x, u);
    f32x16 acc;                          // (two interleaved accumulators: measured, no gain)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bb2[r];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[r], u[r], acc, 0, 0, 0);
    float vp[16], v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vp[r] = acc[r];
    act16<GELU>(act, vp, v);
    float* dst = out + (long long)image_pixel(g, (unsigned)(tile * 32 + li)) * g.co;
    if constexpr (CO4) {
      f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        z0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[r], v[r], z0, 0, 0, 0);
        z1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[r + 1], v[r + 1], z1, 0, 0, 0);
      }
      float zs[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        zs[c] = z0[c] + z1[c];
        zs[c] += __shfl_xor(zs[c], 32, 64);          // the other half of the input channels
      }
      if (kh == 0) {
        if (g.co == 4) {
          *reinterpret_cast<float4*>(dst) = make_float4(zs[0] + b4v.x, zs[1] + b4v.y, zs[2] + b4v.z, zs[3] + b4v.w);
        } else {
          const float bs[4] = {b4v.x, b4v.y, b4v.z, b4v.w};
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c < g.co) dst[c] = zs[c] + bs[c];
        }
      }
    } else {
      f32x16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) z = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[r], v[r], z, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = permk(r, kh);
        if (c < g.co) dst[c] = z[r] + bb4[r];
      }
    }
  }
}

// Backward.  CO4 as in the forward: dv = W4^T dz (contraction over the <= 4 output channels: 4 groups of 4 k-steps,
// group q produces rows i + 8q + 4kh = perm(4q + i, kh) - exactly the register order of the 32x32x2 accumulators) and
// dW4 = dz v^T (contraction over pixels: A = dz[c = lane & 3][pixel 16 kh + s] from a 4-row LDS slab, B = the
// transposed v, one instruction per pixel pair) run on the 4x4x1 blocks; the freed slab and registers (dz, dW4 and db4
// accumulators shrink from 16 to 4 each; db2 is summed from the transposed operand: one register) let two workgroups
// share a CU, which is what hides the LDS / MFMA / HBM latencies of this long dependent chain.
template <bool GELU, bool CO4>
__global__ __launch_bounds__(256, CO4 ? 2 : 1) void out_tail_bwd_kernel(
    const float* __restrict__ upre, const float* __restrict__ dout, const float* __restrict__ w2,
    const float* __restrict__ b2, const float* __restrict__ w4p, float* __restrict__ dupre,
    float* __restrict__ partials, long long ntiles, TailGeom g, int act) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int XZF = CO4 ? 4 * TLD : SLAB;
  constexpr int WSL = 4 * SLAB + XZF;      // floats per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, kh = lane >> 5;
  float* Xd = sm + wave * WSL;             // dvpre [o2][pix]
  float* Xu = Xd + SLAB;                   // u     [o ][pix]
  float* Xv = Xu + SLAB;                   // v     [o2][pix]
  float* stg = Xv + SLAB;                  // [pix][ch] staging for the coalesced Upre load / dUpre store
  float* Xz = stg + SLAB;                  // dz    [c ][pix]   (!CO4: rows >= co stay zero)
  float* Kc = sm + 4 * WSL;                // CO4: per-workgroup constants (see below) - registers are the scarce resource
  if constexpr (!CO4)
    for (int idx = lane; idx < SLAB; idx += 64) Xz[idx] = 0.f;

  const int R4 = 4 * ((g.co + 7) / 8);     // !CO4: MFMA k-steps that can carry a channel c < co
  float a2[16], a2t[16], a4t[16], bb2[16]; // CO4: a4t / bb2 are re-read from Kc every tile (4 + 4 broadcast ds_read_b128)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = permk(r, kh);
    a2[r] = w2[li * TD + o];               // A[i=o2=li][k=o]      : W2
    a2t[r] = w2[o * TD + li];              // A[i=o =li][k=o2]     : W2^T
    if constexpr (!CO4) {
      a4t[r] = w4p[o * TD + li];           // A[i=o2=li][k=c] = W4^T (rows >= co are zero)
      bb2[r] = b2[o];
    }
  }
  // CO4 constants: Kc[(kh*4 + i)*16 + 4q + c] = W4[c][i + 8q + 4kh]  (A operand of dv: block row i = lane & 3, group q,
  // k-step c), Kc[128 + kh*16 + r] = b2[perm(r, kh)]
  const float* const Ka = Kc + ((kh * 4 + (lane & 3)) << 4);
  const float* const Kb = Kc + 128 + (kh << 4);
  if constexpr (CO4) {
    if (threadIdx.x < 160) {
      const int t = threadIdx.x;
      float val;
      if (t < 128) {
        const int r = t & 15, i = (t >> 4) & 3, k2 = t >> 6;
        val = w4p[(r & 3) * TD + i + 8 * (r >> 2) + 4 * k2];
      } else {
        val = b2[permk(t & 15, (t - 128) >> 4)];
      }
      Kc[t] = val;
    }
    __syncthreads();
  }
  f32x16 gW2, gW4;                         // CO4: only gW4[0..7] (two interleaved 4-register accumulators)
  float gb0[16], gb4[16], gb2[16];         // CO4: gb4[0..3], gb2[0], gb0[0..3] (channels 4 (lane & 7) + e, store deal)
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
    // dz in B-operand layout: dz[r] = dOut[pixel][c = perm(r,kh)];  CO4: dz[c], c < 4, on BOTH halves
    const float* src = dout + (long long)image_pixel(g, (unsigned)(tile * 32 + li)) * g.co;
    float dz[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dz[r] = 0.f;
    if (g.co == 4) {
      const float4 t = *reinterpret_cast<const float4*>(src);   // every lane loads (no divergent load)
      if (CO4 || kh == 0) { dz[0] = t.x; dz[1] = t.y; dz[2] = t.z; dz[3] = t.w; }
    } else if constexpr (CO4) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float t = src[c < g.co ? c : 0];
        dz[c] = c < g.co ? t : 0.f;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = permk(r, kh);
        const float t = src[c < g.co ? c : 0];
        dz[r] = c < g.co ? t : 0.f;
      }
    }
    float u[16], du_act[16];
    actvd16<GELU>(act, xs, u, du_act);
    // recompute v
    f32x16 acc;
    if constexpr (CO4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(Kb + 4 * k);
        acc[4 * k] = t.x; acc[4 * k + 1] = t.y; acc[4 * k + 2] = t.z; acc[4 * k + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bb2[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[r], u[r], acc, 0, 0, 0);
    float vp[16], v[16], dvpre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vp[r] = acc[r];
    actvd16<GELU>(act, vp, v, dvpre);  // dvpre holds act'(vpre) for now
    // dv'[o2,pix] = sum_c W4[c,o2] dz[c,pix]
    f32x16 dv;
    if constexpr (CO4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 aq = *reinterpret_cast<const float4*>(Ka + 4 * q);
        const float a4q[4] = {aq.x, aq.y, aq.z, aq.w};
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) d = __builtin_amdgcn_mfma_f32_4x4x1f32(a4q[c], dz[c], d, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) dv[4 * q + i] = d[i];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) dv[r] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (r < R4) dv = __builtin_amdgcn_mfma_f32_32x32x2f32(a4t[r], dz[r], dv, 0, 0, 0);
    }
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
      if (!CO4) {
        gb0[r] += dup[r];
        gb2[r] += dvpre[r];
        gb4[r] += dz[r];
      } else if (r < 4) {
        gb4[r] += dz[r];
      }
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
      if (!CO4 && r < R4) Xz[ch * TLD + li] = dz[r];
    }
    if constexpr (CO4) {
      if (kh == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) Xz[c * TLD + li] = dz[c];
      }
    }
    wave_lds_sync();
    {
      float* drow = dupre + tile * (32 * TD) + lane * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(&stg[(8 * k + (lane >> 3)) * TLD + 4 * (lane & 7)]);
        *reinterpret_cast<float4*>(drow + 256 * k) = t;
        if constexpr (CO4) {               // db0 from the store deal: this lane always carries channels 4 (lane & 7) + e
          gb0[0] += t.x; gb0[1] += t.y; gb0[2] += t.z; gb0[3] += t.w;
        }
      }
    }
    float xd[16], xu[16], xv[16], xz[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int off = li * TLD + 16 * kh + 4 * k;
      const int offz = CO4 ? (lane & 3) * TLD + 16 * kh + 4 * k : off;
      const float4 td = *reinterpret_cast<const float4*>(&Xd[off]);
      const float4 tu = *reinterpret_cast<const float4*>(&Xu[off]);
      const float4 tv = *reinterpret_cast<const float4*>(&Xv[off]);
      const float4 tz = *reinterpret_cast<const float4*>(&Xz[offz]);
      xd[4 * k] = td.x; xd[4 * k + 1] = td.y; xd[4 * k + 2] = td.z; xd[4 * k + 3] = td.w;
      xu[4 * k] = tu.x; xu[4 * k + 1] = tu.y; xu[4 * k + 2] = tu.z; xu[4 * k + 3] = tu.w;
      xv[4 * k] = tv.x; xv[4 * k + 1] = tv.y; xv[4 * k + 2] = tv.z; xv[4 * k + 3] = tv.w;
      xz[4 * k] = tz.x; xz[4 * k + 1] = tz.y; xz[4 * k + 2] = tz.z; xz[4 * k + 3] = tz.w;
    }
    if constexpr (CO4) {
      f32x4 w0 = {gW4[0], gW4[1], gW4[2], gW4[3]}, w1 = {gW4[4], gW4[5], gW4[6], gW4[7]};
      float sd = 0.f;
#pragma unroll
      for (int s = 0; s < 16; s += 2) {
        gW2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xd[s], xu[s], gW2, 0, 0, 0);
        w0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xz[s], xv[s], w0, 0, 0, 0);          // dW4[c][o2 = li] += dz v
        gW2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xd[s + 1], xu[s + 1], gW2, 0, 0, 0);
        w1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xz[s + 1], xv[s + 1], w1, 0, 0, 0);
        sd += xd[s] + xd[s + 1];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) { gW4[c] = w0[c]; gW4[4 + c] = w1[c]; }
      gb2[0] += sd;                        // db2[o2 = li], this lane's half of the pixels
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        gW2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xd[s], xu[s], gW2, 0, 0, 0);   // dW2[o2][o] += dvpre[o2,pix] u[o,pix]
        gW4 = __builtin_amdgcn_mfma_f32_32x32x2f32(xz[s], xv[s], gW4, 0, 0, 0);   // dW4[c][o2]  += dz[c,pix] v[o2,pix]
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- one partial row per WORKGROUP: the waves park their rows in LDS (their own slab region, dead by now), four
  //      rows are summed in a fixed order (a row per wave made the reduction pass read 17.6 MB at DPOT-Tiny B=32)
  float* prow = sm + wave * WSL;
  if constexpr (CO4) {
    float w4s[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      w4s[c] = gW4[c] + gW4[4 + c];
      w4s[c] += __shfl_xor(w4s[c], 32, 64);          // the other half of the pixels
    }
    const float s2 = gb2[0] + __shfl_xor(gb2[0], 32, 64);
    if (kh == 0) prow[2048 + li] = s2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rowi = permk(r, kh);
      prow[rowi * TD + li] = gW2[r];
      prow[1024 + rowi * TD + li] = (kh == 0 && r < 4) ? w4s[r & 3] : 0.f;   // perm(r, 0) = r for r < 4
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s0 = gb0[e];
      s0 += __shfl_xor(s0, 8, 64);
      s0 += __shfl_xor(s0, 16, 64);
      s0 += __shfl_xor(s0, 32, 64);
      if (lane < 8) prow[2080 + 4 * lane + e] = s0;
    }
    float s4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s4[c] = gb4[c];
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) s4[c] += __shfl_xor(s4[c], off, 64);
    }
    if (lane < 32) prow[2112 + lane] = lane < 4 ? s4[lane & 3] : 0.f;   // (dynamic index into 4 registers: selects)
  } else {
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
  __syncthreads();
  float* grow = partials + (long long)blockIdx.x * TAIL_PCOLS;
  for (int c = threadIdx.x; c < TAIL_PCOLS; c += 256)
    grow[c] = (sm[c] + sm[WSL + c]) + (sm[2 * WSL + c] + sm[3 * WSL + c]);
}

static unsigned tail_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / d) + 1u; }
static TailGeom tail_geom(int h, int w, int P, int co) {
  return TailGeom{h, w, P, co, tail_magic((unsigned)(P * P)), tail_magic((unsigned)P), tail_magic((unsigned)w),
                  tail_magic((unsigned)h)};
}
static int tail_grid_fwd(long long ntiles) {
  long long g = (ntiles + 3) / 4;
  if (g > 1024) g = 1024; // 4 resident workgroups per CU (<= 128 registers); each wave walks several tiles (amortises
                          // the weight preload); 16384 tiles at DPOT-Tiny B=32 = exactly 4 per wave
  return (int)(g < 1 ? 1 : g);
}
static int tail_grid_bwd(long long ntiles, bool co4) {
  long long g = (ntiles + 3) / 4;
  const long long cap = co4 ? 512 : 256;   // co <= 4: 76 KiB of LDS and <= 256 registers: two workgroups per CU
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}
static size_t tail_lds_bwd(bool co4) { return sizeof(float) * (4 * (4 * SLAB + (co4 ? 4 * TLD : SLAB)) + (co4 ? 160 : 0)); }

}  // namespace dpot

using namespace dpot;

static int tail_check(const char* who, int B, int h, int w, int P, int co) {
  DPOT_REQUIRE(B > 0 && h > 0 && w > 0 && P > 0 && co > 0 && co <= 32, "%s: bad shape", who);
  DPOT_REQUIRE(((long long)B * h * w * P * P) % 32 == 0, "%s: pixel count must be a multiple of 32", who);
  DPOT_REQUIRE((long long)B * h * w * P * P * 32 < (1ll << 31), "%s: too many pixels for 32-bit indexing", who);
  DPOT_REQUIRE((long long)B * h * w * P * P * P * P < (1ll << 32) && (long long)B * h * w * (h > w ? h : w) < (1ll << 32),
               "%s: geometry outside the range of the reciprocal divisions", who);
  return DPOT_OK;
}

// rows of the partial buffer: sized for the larger (co <= 4) grid, so that the count does not depend on co
extern "C" int dpot_out_tail_partial_rows(int B, int h, int w, int P) {
  return tail_grid_bwd((long long)B * h * w * P * P / 32, true);
}
extern "C" int dpot_out_tail_partial_cols(void) { return TAIL_PCOLS; }

template <bool GELU, bool CO4>
static int tail_launch_fwd(const float* upre, const float* w2, const float* b2, const float* w4, const float* b4,
                           float* out, long long ntiles, TailGeom g, int act, hipStream_t s) {
  hipLaunchKernelGGL((out_tail_fwd_kernel<GELU, CO4>), dim3(tail_grid_fwd(ntiles)), dim3(256), 0, s, upre, w2, b2, w4,
                     b4, out, ntiles, g, act);
  return check_launch("out_tail_fwd_kernel");
}
template <bool GELU, bool CO4>
static int tail_launch_bwd(const float* upre, const float* dout, const float* w2, const float* b2, const float* w4,
                           float* dupre, float* partials, long long ntiles, TailGeom g, int act, hipStream_t s) {
  const size_t lds = tail_lds_bwd(CO4);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(out_tail_bwd_kernel<GELU, CO4>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  DPOT_REQUIRE(attr == hipSuccess, "out_tail_bwd: cannot reserve %zu bytes of LDS", lds);
  hipLaunchKernelGGL((out_tail_bwd_kernel<GELU, CO4>), dim3(tail_grid_bwd(ntiles, CO4)), dim3(256), lds, s, upre, dout,
                     w2, b2, w4, dupre, partials, ntiles, g, act);
  return check_launch("out_tail_bwd_kernel");
}

extern "C" int dpot_out_tail_fwd(const float* upre, const float* w2, const float* b2, const float* w4, const float* b4,
                                 float* out, int B, int h, int w, int P, int co, int act, dpot_stream_t stream) {
  int rc = tail_check("out_tail_fwd", B, h, w, P, co);
  if (rc) return rc;
  DPOT_REQUIRE(upre && w2 && b2 && w4 && b4 && out && aligned16(upre) && aligned16(out) && aligned16(b4),
               "out_tail_fwd: bad pointer");
  const long long ntiles = (long long)B * h * w * P * P / 32;
  const TailGeom g = tail_geom(h, w, P, co);
  hipStream_t s = as_stream(stream);
  const bool gelu = act == DPOT_ACT_GELU;
  if (co <= 4) return gelu ? tail_launch_fwd<true, true>(upre, w2, b2, w4, b4, out, ntiles, g, act, s)
                           : tail_launch_fwd<false, true>(upre, w2, b2, w4, b4, out, ntiles, g, act, s);
  return gelu ? tail_launch_fwd<true, false>(upre, w2, b2, w4, b4, out, ntiles, g, act, s)
              : tail_launch_fwd<false, false>(upre, w2, b2, w4, b4, out, ntiles, g, act, s);
}

// partials: [dpot_out_tail_partial_rows][dpot_out_tail_partial_cols]; rows the grid does not reach are zero-filled
extern "C" int dpot_out_tail_bwd(const float* upre, const float* dout, const float* w2, const float* b2, const float* w4,
                                 float* dupre, float* partials, int B, int h, int w, int P, int co, int act,
                                 dpot_stream_t stream) {
  int rc = tail_check("out_tail_bwd", B, h, w, P, co);
  if (rc) return rc;
  DPOT_REQUIRE(upre && dout && w2 && b2 && w4 && dupre && partials && aligned16(upre) && aligned16(dupre) &&
                   aligned16(dout),
               "out_tail_bwd: bad pointer");
  const long long ntiles = (long long)B * h * w * P * P / 32;
  const TailGeom g = tail_geom(h, w, P, co);
  hipStream_t s = as_stream(stream);
  const bool gelu = act == DPOT_ACT_GELU, co4 = co <= 4;
  const int rows_all = tail_grid_bwd(ntiles, true), rows = tail_grid_bwd(ntiles, co4);
  if (rows < rows_all) {
    const hipError_t e = hipMemsetAsync(partials + (size_t)rows * TAIL_PCOLS, 0,
                                        sizeof(float) * (size_t)(rows_all - rows) * TAIL_PCOLS, s);
    DPOT_REQUIRE(e == hipSuccess, "out_tail_bwd: memset failed");
  }
  if (co4) return gelu ? tail_launch_bwd<true, true>(upre, dout, w2, b2, w4, dupre, partials, ntiles, g, act, s)
                       : tail_launch_bwd<false, true>(upre, dout, w2, b2, w4, dupre, partials, ntiles, g, act, s);
  return gelu ? tail_launch_bwd<true, false>(upre, dout, w2, b2, w4, dupre, partials, ntiles, g, act, s)
              : tail_launch_bwd<false, false>(upre, dout, w2, b2, w4, dupre, partials, ntiles, g, act, s);
}
