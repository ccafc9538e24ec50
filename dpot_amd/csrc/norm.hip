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
  __shared__ double shd[32];
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
  __shared__ double shd[32];
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
  block_sum2_d(s1, s2, shd);
  const float m1 = (float)(s1 / n);
  const float m2 = (float)(s2 / n);
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

// dgamma/dbeta = sum over samples of the per-sample partials; block = 64 channels x 4 sample lanes, fixed order
struct GnParamJobs {
  const float* part[4];
  float* dgamma[4];
  float* dbeta[4];
};
// blockIdx.y = job: several GroupNorm layers (the two of a DPOT block) share one launch - each of these reductions is
// a handful of workgroups and pure launch / dependent-load latency
__global__ __launch_bounds__(256) void groupnorm_param_grad_kernel(const GnParamJobs jobs, int B, int E) {
  __shared__ float red[2][4][64];
  const float* __restrict__ part = jobs.part[blockIdx.y];
  float* __restrict__ dgamma = jobs.dgamma[blockIdx.y];
  float* __restrict__ dbeta = jobs.dbeta[blockIdx.y];
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tc;
  float sg0 = 0.f, sg1 = 0.f, sb0 = 0.f, sb1 = 0.f;
  if (c < E) {
    int b = tr;
    for (; b + 4 < B; b += 8) {
      sg0 += part[((long long)0 * B + b) * E + c];
      sg1 += part[((long long)0 * B + b + 4) * E + c];
      sb0 += part[((long long)1 * B + b) * E + c];
      sb1 += part[((long long)1 * B + b + 4) * E + c];
    }
    for (; b < B; b += 4) {
      sg0 += part[((long long)0 * B + b) * E + c];
      sb0 += part[((long long)1 * B + b) * E + c];
    }
  }
  red[0][tr][tc] = sg0 + sg1;
  red[1][tr][tc] = sb0 + sb1;
  __syncthreads();
  if (tr == 0 && c < E) {
    dgamma[c] = (red[0][0][tc] + red[0][1][tc]) + (red[0][2][tc] + red[0][3][tc]);
    dbeta[c] = (red[1][0][tc] + red[1][1][tc]) + (red[1][2][tc] + red[1][3][tc]);
  }
}

// ---- register-resident variants: the (b, group) slab is read from HBM exactly once --------------------
// usable when the channel quads of a group map 1:1 onto the TJ lanes (cg/4 a power of two <= 64) and the
// T/TT tokens a thread owns fit in ITEMS registers-quads (DPOT-Ti/S/M at 128^2: ITEMS = 4 / 8)
template <int ITEMS>
__global__ __launch_bounds__(GN_THREADS) void groupnorm_fwd_cached_kernel(const float* __restrict__ x,
                                                                          const float* __restrict__ gamma,
                                                                          const float* __restrict__ beta,
                                                                          float* __restrict__ y, float* __restrict__ mean,
                                                                          float* __restrict__ rstd, int T, int E, int G,
                                                                          float eps) {
  __shared__ double shd[32];
  const int g = blockIdx.x, b = blockIdx.y;
  const int cg = E / G, TJ = cg / 4, TT = GN_THREADS / TJ;
  const int tj = threadIdx.x % TJ, tt = threadIdx.x / TJ;
  const float* xs = x + (long long)b * T * E + g * cg + tj * 4;
  float* ys = y + (long long)b * T * E + g * cg + tj * 4;
  const double n = (double)T * cg;
  float4 v[ITEMS];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int t = tt + i * TT;
    v[i] = *reinterpret_cast<const float4*>(xs + (long long)(t < T ? t : T - 1) * E);   // clamped, masked below
    if (t < T) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mu = (float)(block_sum_d((double)s, shd) / n);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    if (tt + i * TT < T) {
      const float d0 = v[i].x - mu, d1 = v[i].y - mu, d2 = v[i].z - mu, d3 = v[i].w - mu;
      q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); q = fmaf(d2, d2, q); q = fmaf(d3, d3, q);
    }
  }
  const float var = (float)(block_sum_d((double)q, shd) / n);
  const float rs = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean[b * G + g] = mu;
    rstd[b * G + g] = rs;
  }
  float4 ga = *reinterpret_cast<const float4*>(gamma + g * cg + tj * 4);
  const float4 be = *reinterpret_cast<const float4*>(beta + g * cg + tj * 4);
  ga.x *= rs; ga.y *= rs; ga.z *= rs; ga.w *= rs;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int t = tt + i * TT;
    if (t < T)
      *reinterpret_cast<float4*>(ys + (long long)t * E) =
          make_float4(fmaf(v[i].x - mu, ga.x, be.x), fmaf(v[i].y - mu, ga.y, be.y), fmaf(v[i].z - mu, ga.z, be.z),
                      fmaf(v[i].w - mu, ga.w, be.w));
  }
}

// PK (round 5; 128 channels per group, T % 32 == 0): the kernel ALSO writes dx - the gradient that enters the previous block's
// channel-MLP backward - as that backward's bf16 operands: the row-form pack (A operand of the fc2 data gradient), the
// transposed pack (operand of the fc2 weight gradient) and its per-sample column sums (bias gradient), i.e. what
// dpot_bf16_pack_both(dx, column sums) would produce from a second pass over dx.  An item of the thread grid = 32 tokens x 128
// channels = one row tile of the pack: staged as bf16 in LDS, written out as eight whole 1 KiB blocks per form.
struct GnBwdPacks {
  uint4* rows;          // [B*T / 32][E / 16][64 chunks][8 bf16]
  uint4* trans;         // [E / 32][B*T / 16][64 chunks][8 bf16]
  float* colsum;        // [B, E]: sum over the sample's tokens
};
__device__ __forceinline__ unsigned gn_pack2(float lo, float hi) {     // v_cvt_pk_bf16_f32, round to nearest even
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

template <int ITEMS, bool PK>
__global__ __launch_bounds__(GN_THREADS) void groupnorm_bwd_cached_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ add,
    float* __restrict__ dx, float* __restrict__ part, int B, int T, int E, int G, const GnBwdPacks pk) {
  __shared__ float red[2][GN_THREADS][4];
  __shared__ double shd[32];
  __shared__ __attribute__((aligned(16))) unsigned short tile[PK ? 32 * 128 : 8];   // [32 tokens][128 channels] bf16
  const int g = blockIdx.x, b = blockIdx.y;
  const int cg = E / G, TJ = cg / 4, TT = GN_THREADS / TJ;
  const int tj = threadIdx.x % TJ, tt = threadIdx.x / TJ;
  const long long off = (long long)b * T * E + g * cg + tj * 4;
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  const double n = (double)T * cg;
  float4 d[ITEMS], xh[ITEMS];
  float a_dy[4] = {0.f, 0.f, 0.f, 0.f}, a_dyx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int t = tt + i * TT;
    const long long o = off + (long long)(t < T ? t : T - 1) * E;
    d[i] = *reinterpret_cast<const float4*>(dy + o);
    const float4 xv = *reinterpret_cast<const float4*>(x + o);
    xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
    if (t < T) {
      a_dy[0] += d[i].x; a_dy[1] += d[i].y; a_dy[2] += d[i].z; a_dy[3] += d[i].w;
      a_dyx[0] = fmaf(d[i].x, xh[i].x, a_dyx[0]); a_dyx[1] = fmaf(d[i].y, xh[i].y, a_dyx[1]);
      a_dyx[2] = fmaf(d[i].z, xh[i].z, a_dyx[2]); a_dyx[3] = fmaf(d[i].w, xh[i].w, a_dyx[3]);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[0][threadIdx.x][k] = a_dyx[k];
    red[1][threadIdx.x][k] = a_dy[k];
  }
  __syncthreads();
  double s1 = 0.0, s2 = 0.0;
  if (tt < 4) {   // thread (tj, k = tt) reduces channel 4*tj + k over the token lanes, fixed order
    const int k = tt;
    float sg = 0.f, sb = 0.f;
    for (int r = 0; r < TT; ++r) {
      sg += red[0][r * TJ + tj][k];
      sb += red[1][r * TJ + tj][k];
    }
    const int c = g * cg + tj * 4 + k;
    part[((long long)0 * B + b) * E + c] = sg;
    part[((long long)1 * B + b) * E + c] = sb;
    const float ga = gamma[c];
    s1 = (double)ga * sb;
    s2 = (double)ga * sg;
  }
  block_sum2_d(s1, s2, shd);
  const float m1 = (float)(s1 / n);
  const float m2 = (float)(s2 / n);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + g * cg + tj * 4);
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int t = tt + i * TT;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < T) {
      const long long o = off + (long long)t * E;
      r = make_float4(rs * (ga.x * d[i].x - m1 - xh[i].x * m2), rs * (ga.y * d[i].y - m1 - xh[i].y * m2),
                      rs * (ga.z * d[i].z - m1 - xh[i].z * m2), rs * (ga.w * d[i].w - m1 - xh[i].w * m2));
      if (add) {
        const float4 a = *reinterpret_cast<const float4*>(add + o);
        r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
      }
      *reinterpret_cast<float4*>(dx + o) = r;
    }
    if constexpr (PK) {
      // host-checked: cg == 128 (TJ = TT = 32: item i = tokens 32 i .. 32 i + 31 of the sample = row tile b * T / 32 + i), T % 32 == 0
      if (32 * i < T) {
        cs[0] += r.x; cs[1] += r.y; cs[2] += r.z; cs[3] += r.w;
        *reinterpret_cast<uint2*>(&tile[tt * 128 + 4 * tj]) = make_uint2(gn_pack2(r.x, r.y), gn_pack2(r.z, r.w));
        __syncthreads();
        const int tid = threadIdx.x;
        if (tid < 512) {
          // row form: block fb (16 features) of this row tile, chunk l = (token l & 31, features 8 (l >> 5) .. + 7)
          const int fb = tid >> 6, l = tid & 63;
          const uint4 v = *reinterpret_cast<const uint4*>(&tile[(l & 31) * 128 + 16 * fb + 8 * (l >> 5)]);
          pk.rows[((long long)(b * (T >> 5) + i) * (E >> 4) + g * 8 + fb) * 64 + l] = v;
        } else {
          // transposed form: block (feature tile ft, token block tb), chunk l = (feature l & 31, tokens 8 (l >> 5) .. + 7)
          const int c2 = tid - 512, bt = c2 >> 6, l = c2 & 63, ft = bt >> 1, tb = bt & 1;
          const unsigned short* src = &tile[(16 * tb + 8 * (l >> 5)) * 128 + 32 * ft + (l & 31)];
          unsigned w[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = (unsigned)src[(2 * j) * 128] | ((unsigned)src[(2 * j + 1) * 128] << 16);
          pk.trans[((long long)(g * 4 + ft) * (((long long)B * T) >> 4) + b * (T >> 4) + 2 * i + tb) * 64 + l] =
              make_uint4(w[0], w[1], w[2], w[3]);
        }
        __syncthreads();
      }
    }
  }
  if constexpr (PK) {
    // column sums of dx over the sample's tokens: the token lanes of a channel quad through `red`, fixed order
#pragma unroll
    for (int k = 0; k < 4; ++k) red[0][threadIdx.x][k] = cs[k];
    __syncthreads();
    if (tt < 4) {
      float sacc = 0.f;
      for (int r = 0; r < TT; ++r) sacc += red[0][r * TJ + tj][tt];
      pk.colsum[(long long)b * E + g * cg + tj * 4 + tt] = sacc;
    }
  }
}

// ITEMS (4 or 8) if the cached kernels apply, else 0
static int gn_cached_items(int T, int E, int G) {
  const int cg = E / G;
  if (cg % 4) return 0;
  const int tj = cg / 4;
  if (tj < 1 || tj > 64 || (tj & (tj - 1))) return 0;
  const int tt = GN_THREADS / tj;
  if (T <= 4 * tt) return 4;
  if (T <= 8 * tt) return 8;
  return 0;
}


// ---- chunked variants: large slabs on few (sample, group) pairs -------------------------------------------------------
// DPOT-L at 256^2, B = 4: a (b, group) slab is 1024 tokens x 192 channels = 768 KiB and there are only B*G = 32 of them:
// one workgroup per slab keeps 32 of 256 CUs busy (measured 69 us forward / 92 us backward for 25 MB tensors).  Here a
// slab is cut into token chunks, grid (chunks, G, B) >= 512 workgroups, and the group statistics go through a tiny
// workspace: pass A writes per-chunk partials, pass B merges them in its prologue (every workgroup of the slab redoes the
// merge - `chunks` values, fixed order) and applies.  x is read twice (the second read mostly from L2 / MALL).
struct GnChunks {
  int TC, chunks;      // tokens per chunk, chunks per slab
};

// forward, pass A: per-chunk (mean, M2 = sum (x - mean)^2) in ONE sweep (round 5): sums of (x - p) and (x - p)^2 around a pivot p
// close to the chunk mean (M2 = S2 - S1^2 / n then loses nothing measurable; large-mean and outlier cases in
// tests/test_gpu_ops.py::test_groupnorm_chunked_*), block sums and the final arithmetic in double.  (Rounds 2-4 swept the chunk twice - mean, then
// centred squares; at DPOT-L batch 16 the 32 chunks an XCD works on are 6.3 MB, more than its L2: the second sweep was not free -
// 30 us per launch for a 100 MB tensor.)
__global__ __launch_bounds__(GN_THREADS) void gn_chunk_stats_kernel(const float* __restrict__ x, float* __restrict__ ws,
                                                                    int T, int E, int G, GnChunks c) {
  __shared__ double shd[32];
  const int ch = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int cg = E / G, q4 = cg >> 2;
  const int t0 = ch * c.TC;
  const int nt = T - t0 < c.TC ? T - t0 : c.TC;
  const unsigned nq = (unsigned)nt * (unsigned)q4;
  const float* xs = x + ((long long)b * T + t0) * E + g * cg;
  // pivot: the mean of GN_THREADS float4s sampled with a STRIDE over the whole chunk (one per thread, index tid * nq / GN_THREADS)
  // - within sigma / 32 of the chunk mean for ordinary data, one outlier among them moves it by 1/1024 of its size only, and a
  // region that differs from the rest of the chunk (a constant / masked border at its start: round 5 sampled the FIRST
  // GN_THREADS float4s only, ~21 tokens, and would have lost (offset / sigma)^2 x 1e-7 of M2 there - ADVICE r5) is sampled in
  // proportion.  The sample and the first GN_PF float4s of every thread are loaded BEFORE the pivot's block reduction (they fly
  // while it runs: the reduction's two barriers otherwise delay the first load of every workgroup by ~5 us, most of what the
  // single sweep saves)
  constexpr int GN_PF = 6;
  float4 pre[GN_PF];
#pragma unroll
  for (int k = 0; k < GN_PF; ++k) {
    const unsigned i = threadIdx.x + (unsigned)k * GN_THREADS;
    const unsigned ic = i < nq ? i : 0u;
    const unsigned t = ic / (unsigned)q4, j = ic - t * (unsigned)q4;
    pre[k] = *reinterpret_cast<const float4*>(xs + (long long)t * E + 4 * j);
  }
  float p0;
  {
    const unsigned cnt = nq < (unsigned)GN_THREADS ? nq : (unsigned)GN_THREADS;
    const unsigned is = nq < (unsigned)GN_THREADS ? (threadIdx.x < nq ? threadIdx.x : 0u)
                                                  : (unsigned)(((unsigned long long)threadIdx.x * nq) / GN_THREADS);
    const unsigned ts = is / (unsigned)q4, js = is - ts * (unsigned)q4;
    const float4 smp = *reinterpret_cast<const float4*>(xs + (long long)ts * E + 4 * js);
    const double loc = threadIdx.x < cnt ? (double)((smp.x + smp.y) + (smp.z + smp.w)) : 0.0;
    p0 = (float)(block_sum_d(loc, shd) / (4.0 * cnt));
  }
  float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
  auto take = [&](const float4 v) __attribute__((always_inline)) {
    const float d0 = v.x - p0, d1 = v.y - p0, d2 = v.z - p0, d3 = v.w - p0;
    s0 += d0 + d1;
    s1 += d2 + d3;
    q0 = fmaf(d0, d0, q0); q0 = fmaf(d1, d1, q0);
    q1 = fmaf(d2, d2, q1); q1 = fmaf(d3, d3, q1);
  };
#pragma unroll
  for (int k = 0; k < GN_PF; ++k)
    if (threadIdx.x + (unsigned)k * GN_THREADS < nq) take(pre[k]);
  for (unsigned i = threadIdx.x + (unsigned)GN_PF * GN_THREADS; i < nq; i += GN_THREADS) {
    const unsigned t = i / (unsigned)q4, j = i - t * (unsigned)q4;
    take(*reinterpret_cast<const float4*>(xs + (long long)t * E + 4 * j));
  }
  const double n = (double)nt * cg;
  double S1 = (double)s0 + (double)s1, S2 = (double)q0 + (double)q1;
  block_sum2_d(S1, S2, shd);
  if (threadIdx.x == 0) {
    float* w = ws + (((long long)b * G + g) * c.chunks + ch) * 2;
    double m2 = S2 - S1 * S1 / n;
    w[0] = (float)((double)p0 + S1 / n);
    w[1] = (float)(m2 > 0.0 ? m2 : 0.0);
  }
}

// merge of the chunk partials of slab (b, g) (Chan et al.), identical in every workgroup of the slab
__device__ __forceinline__ void gn_chunk_merge(const float* __restrict__ ws, int b, int g, int G, int T, int cg, GnChunks c,
                                               float eps, float& mu, float& rs) {
  const float* w = ws + ((long long)b * G + g) * c.chunks * 2;
  double sm = 0.0;
  for (int k = 0; k < c.chunks; ++k) {
    const int nt = T - k * c.TC < c.TC ? T - k * c.TC : c.TC;
    sm += (double)w[2 * k] * ((double)nt * cg);
  }
  const double N = (double)T * cg, mean = sm / N;
  double m2 = 0.0;
  for (int k = 0; k < c.chunks; ++k) {
    const int nt = T - k * c.TC < c.TC ? T - k * c.TC : c.TC;
    const double d = (double)w[2 * k] - mean;
    m2 += (double)w[2 * k + 1] + d * d * ((double)nt * cg);
  }
  mu = (float)mean;
  rs = 1.0f / sqrtf((float)(m2 / N) + eps);
}

// forward, pass B
__global__ __launch_bounds__(GN_THREADS) void gn_chunk_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float* __restrict__ y,
                                                                    float* __restrict__ mean, float* __restrict__ rstd,
                                                                    const float* __restrict__ ws, int T, int E, int G,
                                                                    GnChunks c, float eps) {
  const int ch = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int cg = E / G, q4 = cg >> 2;
  float mu, rs;
  gn_chunk_merge(ws, b, g, G, T, cg, c, eps, mu, rs);
  if (ch == 0 && threadIdx.x == 0) {
    mean[b * G + g] = mu;
    rstd[b * G + g] = rs;
  }
  if (!y) return;                                          // statistics only (grid (1, G, B))
  const int t0 = ch * c.TC;
  const int nt = T - t0 < c.TC ? T - t0 : c.TC;
  const unsigned nq = (unsigned)nt * (unsigned)q4;
  const long long off = ((long long)b * T + t0) * E + g * cg;
  for (unsigned i = threadIdx.x; i < nq; i += GN_THREADS) {
    const unsigned t = i / (unsigned)q4, j = i - t * (unsigned)q4;
    const long long o = off + (long long)t * E + 4 * j;
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    float4 ga = *reinterpret_cast<const float4*>(gamma + g * cg + 4 * j);
    const float4 be = *reinterpret_cast<const float4*>(beta + g * cg + 4 * j);
    ga.x *= rs; ga.y *= rs; ga.z *= rs; ga.w *= rs;
    *reinterpret_cast<float4*>(y + o) = make_float4(fmaf(v.x - mu, ga.x, be.x), fmaf(v.y - mu, ga.y, be.y),
                                                    fmaf(v.z - mu, ga.z, be.z), fmaf(v.w - mu, ga.w, be.w));
  }
}

// backward, pass A: per-chunk per-channel sums  ws[((b*chunks + ch)*2 + k)*E + c]:  k = 0: sum_t dy*xhat, k = 1: sum_t dy
__global__ __launch_bounds__(GN_THREADS) void gn_chunk_bwd_part_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                       const float* __restrict__ mean,
                                                                       const float* __restrict__ rstd, float* __restrict__ ws,
                                                                       int T, int E, int G, GnChunks c) {
  __shared__ float red[GN_THREADS][8];
  const int ch = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int cg = E / G, q4 = cg >> 2;
  const int TT = GN_THREADS / q4;                         // token lanes; threads >= TT * q4 idle
  const int tj = threadIdx.x % q4, tt = threadIdx.x / q4;
  const int t0 = ch * c.TC;
  const int nt = T - t0 < c.TC ? T - t0 : c.TC;
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  const long long off = ((long long)b * T + t0) * E + g * cg + 4 * tj;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (tt < TT) {
    for (int t = tt; t < nt; t += TT) {
      const float4 d = *reinterpret_cast<const float4*>(dy + off + (long long)t * E);
      const float4 v = *reinterpret_cast<const float4*>(x + off + (long long)t * E);
      a[0] = fmaf(d.x, (v.x - mu) * rs, a[0]); a[1] = fmaf(d.y, (v.y - mu) * rs, a[1]);
      a[2] = fmaf(d.z, (v.z - mu) * rs, a[2]); a[3] = fmaf(d.w, (v.w - mu) * rs, a[3]);
      a[4] += d.x; a[5] += d.y; a[6] += d.z; a[7] += d.w;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = a[k];
  __syncthreads();
  if (tt == 0) {                                          // fixed order over the token lanes
    float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < TT; ++l)
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] += red[l * q4 + tj][k];
    float* w = ws + ((long long)b * c.chunks + ch) * 2 * E + g * cg + 4 * tj;
    *reinterpret_cast<float4*>(w) = make_float4(r[0], r[1], r[2], r[3]);
    *reinterpret_cast<float4*>(w + E) = make_float4(r[4], r[5], r[6], r[7]);
  }
}

// backward, pass B: per-channel totals over the chunks (-> the per-sample parameter-gradient partials, written by chunk 0),
// group means m1 = mean_g(gamma dy), m2 = mean_g(gamma dy xhat), then dx for this chunk
__global__ __launch_bounds__(GN_THREADS) void gn_chunk_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ add,
    float* __restrict__ dx, float* __restrict__ part, const float* __restrict__ ws, int B, int T, int E, int G,
    GnChunks c) {
  __shared__ double shd[32];
  const int ch = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int cg = E / G, q4 = cg >> 2;
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  double s1 = 0.0, s2 = 0.0;
  if ((int)threadIdx.x < cg) {
    const int col = g * cg + threadIdx.x;
    float tdx = 0.f, td = 0.f;
    for (int k = 0; k < c.chunks; ++k) {
      const float* w = ws + ((long long)b * c.chunks + k) * 2 * E + col;
      tdx += w[0];
      td += w[E];
    }
    if (ch == 0) {
      part[((long long)0 * B + b) * E + col] = tdx;
      part[((long long)1 * B + b) * E + col] = td;
    }
    const float gm = gamma[col];
    s1 = (double)(gm * td);
    s2 = (double)(gm * tdx);
  }
  const double n = (double)T * cg;
  block_sum2_d(s1, s2, shd);
  const float m1 = (float)(s1 / n);
  const float m2 = (float)(s2 / n);
  const int t0 = ch * c.TC;
  const int nt = T - t0 < c.TC ? T - t0 : c.TC;
  const unsigned nq = (unsigned)nt * (unsigned)q4;
  const long long off = ((long long)b * T + t0) * E + g * cg;
  for (unsigned i = threadIdx.x; i < nq; i += GN_THREADS) {
    const unsigned t = i / (unsigned)q4, j = i - t * (unsigned)q4;
    const long long o = off + (long long)t * E + 4 * j;
    const float4 d = *reinterpret_cast<const float4*>(dy + o);
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + g * cg + 4 * j);
    float4 r;
    r.x = rs * (ga.x * d.x - m1 - (v.x - mu) * rs * m2);
    r.y = rs * (ga.y * d.y - m1 - (v.y - mu) * rs * m2);
    r.z = rs * (ga.z * d.z - m1 - (v.z - mu) * rs * m2);
    r.w = rs * (ga.w * d.w - m1 - (v.w - mu) * rs * m2);
    if (add) {
      const float4 a4 = *reinterpret_cast<const float4*>(add + o);
      r.x += a4.x; r.y += a4.y; r.z += a4.z; r.w += a4.w;
    }
    *reinterpret_cast<float4*>(dx + o) = r;
  }
}

// pass B with the gradient packs (round 5, see groupnorm_bwd_cached_kernel<.., PK>): the same dx, written ALSO as the bf16 operand
// packs + column-sum partials of the previous block's channel-MLP backward.  Few, large slabs (DPOT-L: 1024 tokens x 192 channels
// per (sample, group), chunks of 256 tokens): the chunk is walked in sub-tiles of 32 tokens = one row tile of the pack, staged in
// fp32 in LDS ([32][cg]) - the packs leave as whole 1 KiB blocks, the column sums are formed from the staged tile in a fixed order.
// Host-checked: cg % 32 == 0, cg <= 256, TC % 32 == 0, T % 32 == 0.  colsum: [B * chunks, E] (one row per chunk).
__global__ __launch_bounds__(GN_THREADS) void gn_chunk_bwd_apply_pk_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ add,
    float* __restrict__ dx, float* __restrict__ part, const float* __restrict__ ws, int B, int T, int E, int G,
    GnChunks c, const GnBwdPacks pk) {
  __shared__ double shd[32];
  __shared__ __attribute__((aligned(16))) float tile[32 * 256];
  const int ch = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int cg = E / G, q4 = cg >> 2;
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  double s1 = 0.0, s2 = 0.0;
  if ((int)threadIdx.x < cg) {
    const int col = g * cg + threadIdx.x;
    float tdx = 0.f, td = 0.f;
    for (int k = 0; k < c.chunks; ++k) {
      const float* w = ws + ((long long)b * c.chunks + k) * 2 * E + col;
      tdx += w[0];
      td += w[E];
    }
    if (ch == 0) {
      part[((long long)0 * B + b) * E + col] = tdx;
      part[((long long)1 * B + b) * E + col] = td;
    }
    const float gm = gamma[col];
    s1 = (double)(gm * td);
    s2 = (double)(gm * tdx);
  }
  const double n = (double)T * cg;
  block_sum2_d(s1, s2, shd);
  const float m1 = (float)(s1 / n);
  const float m2 = (float)(s2 / n);
  const int t0 = ch * c.TC;
  const int nt = T - t0 < c.TC ? T - t0 : c.TC;
  const int tid = threadIdx.x;
  const int nfb = cg >> 4, nft = cg >> 5;                  // 16-feature blocks / 32-feature tiles of the group
  float csum = 0.f;
  for (int st = 0; st * 32 < nt; ++st) {
    const int ts = t0 + 32 * st;                           // first token of the sub-tile (a multiple of 32)
    const long long off = ((long long)b * T + ts) * E + g * cg;
    for (unsigned i = tid; i < 32u * (unsigned)q4; i += GN_THREADS) {
      const unsigned t = i / (unsigned)q4, j = i - t * (unsigned)q4;
      const long long o = off + (long long)t * E + 4 * j;
      const float4 d = *reinterpret_cast<const float4*>(dy + o);
      const float4 v = *reinterpret_cast<const float4*>(x + o);
      const float4 ga = *reinterpret_cast<const float4*>(gamma + g * cg + 4 * j);
      float4 r;
      r.x = rs * (ga.x * d.x - m1 - (v.x - mu) * rs * m2);
      r.y = rs * (ga.y * d.y - m1 - (v.y - mu) * rs * m2);
      r.z = rs * (ga.z * d.z - m1 - (v.z - mu) * rs * m2);
      r.w = rs * (ga.w * d.w - m1 - (v.w - mu) * rs * m2);
      if (add) {
        const float4 a4 = *reinterpret_cast<const float4*>(add + o);
        r.x += a4.x; r.y += a4.y; r.z += a4.z; r.w += a4.w;
      }
      *reinterpret_cast<float4*>(dx + o) = r;
      *reinterpret_cast<float4*>(&tile[t * cg + 4 * j]) = r;
    }
    __syncthreads();
    const long long rt = ((long long)b * T + ts) >> 5;     // row tile of the pack
    for (int cidx = tid; cidx < nfb * 64; cidx += GN_THREADS) {
      // row form: block fb, chunk l = (token l & 31, features 8 (l >> 5) .. + 7)
      const int fb = cidx >> 6, l = cidx & 63;
      const float* sp = &tile[(l & 31) * cg + 16 * fb + 8 * (l >> 5)];
      const float4 a0 = *reinterpret_cast<const float4*>(sp), a1 = *reinterpret_cast<const float4*>(sp + 4);
      pk.rows[(rt * (E >> 4) + (g * cg >> 4) + fb) * 64 + l] =
          make_uint4(gn_pack2(a0.x, a0.y), gn_pack2(a0.z, a0.w), gn_pack2(a1.x, a1.y), gn_pack2(a1.z, a1.w));
    }
    for (int cidx = tid; cidx < nft * 128; cidx += GN_THREADS) {
      // transposed form: block (feature tile ft, token block tb), chunk l = (feature l & 31, tokens 8 (l >> 5) .. + 7)
      const int bt = cidx >> 6, l = cidx & 63, ft = bt >> 1, tb = bt & 1;
      const float* sp = &tile[(16 * tb + 8 * (l >> 5)) * cg + 32 * ft + (l & 31)];
      unsigned w[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) w[jj] = gn_pack2(sp[(2 * jj) * cg], sp[(2 * jj + 1) * cg]);
      pk.trans[((long long)((g * cg >> 5) + ft) * (((long long)B * T) >> 4) + (((long long)b * T + ts) >> 4) + tb) * 64 + l] =
          make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (tid < cg) {                                        // column sums of the sub-tile, tokens ascending
      float a = 0.f;
      for (int t = 0; t < 32; ++t) a += tile[t * cg + tid];
      csum += a;
    }
    __syncthreads();
  }
  if (tid < cg) pk.colsum[((long long)b * c.chunks + ch) * E + g * cg + tid] = csum;
}

// chunking of a slab, or {0, 0} when the one-workgroup-per-slab kernels are the right ones
static GnChunks gn_chunking(int B, int T, int E, int G) {
  constexpr int enabled = 1;
  GnChunks c{0, 0};
  const int cg = E / G;
  if (!enabled || cg % 4 || cg > GN_THREADS || gn_cached_items(T, E, G) != 0) return c;
  const long long slabs = (long long)B * G;
  if (slabs >= 192 || T < 64) return c;                   // enough slabs to fill the chip as they are
  long long chunks = (512 + slabs - 1) / slabs;
  if (chunks > T / 16) chunks = T / 16;
  if (chunks > 64) chunks = 64;
  if (chunks < 2) return c;
  c.TC = (int)((T + chunks - 1) / chunks);
  c.chunks = (T + c.TC - 1) / c.TC;
  return c;
}

}  // namespace dpot

using namespace dpot;

extern "C" int64_t dpot_groupnorm_ws_elems(int B, int T, int E, int G) {
  if (B <= 0 || T <= 0 || E <= 0 || G <= 0 || E % G) return 0;
  const GnChunks c = gn_chunking(B, T, E, G);
  return c.chunks ? (int64_t)B * c.chunks * 2 * E : 0;    // backward partials (the forward needs B*G*chunks*2 <= this)
}

extern "C" int dpot_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                  float* rstd, float* workspace, int B, int T, int E, int G, float eps,
                                  dpot_stream_t stream) {
  DPOT_REQUIRE(x && gamma && beta && mean && rstd, "groupnorm_fwd: null pointer");
  DPOT_REQUIRE(B > 0 && T > 0 && E > 0 && G > 0 && E % G == 0 && B <= 65535, "groupnorm_fwd: bad shape");
  const bool vec = ((E / G) % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta);
  const int items = vec ? gn_cached_items(T, E, G) : 0;
  const GnChunks ck = vec && workspace && aligned16(workspace) ? gn_chunking(B, T, E, G) : GnChunks{0, 0};
  if (ck.chunks) {
    const dim3 grid(ck.chunks, G, B);
    hipLaunchKernelGGL(gn_chunk_stats_kernel, grid, dim3(GN_THREADS), 0, as_stream(stream), x, workspace, T, E, G, ck);
    int rc = check_launch("gn_chunk_stats_kernel");
    if (rc) return rc;
    // y == NULL: statistics only - the consumer applies the affine map itself (rfft2 with `norm` operands, the bf16 pack of
    // the channel MLP's input): the merge of the chunk partials runs once per (sample, group)
    hipLaunchKernelGGL(gn_chunk_apply_kernel, y ? grid : dim3(1, G, B), dim3(GN_THREADS), 0, as_stream(stream), x, gamma,
                       beta, y, mean, rstd, (const float*)workspace, T, E, G, ck, eps);
    return check_launch("gn_chunk_apply_kernel");
  }
  DPOT_REQUIRE(y, "groupnorm_fwd: statistics-only calls (y == NULL) need the chunked path (dpot_groupnorm_ws_elems > 0)");
  if (items == 4)
    hipLaunchKernelGGL(groupnorm_fwd_cached_kernel<4>, dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), x, gamma, beta,
                       y, mean, rstd, T, E, G, eps);
  else if (items == 8)
    hipLaunchKernelGGL(groupnorm_fwd_cached_kernel<8>, dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), x, gamma, beta,
                       y, mean, rstd, T, E, G, eps);
  else if (vec)
    hipLaunchKernelGGL(groupnorm_fwd_kernel<4>, dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), x, gamma, beta, y,
                       mean, rstd, T, E, G, eps);
  else
    hipLaunchKernelGGL(groupnorm_fwd_kernel<1>, dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), x, gamma, beta, y,
                       mean, rstd, T, E, G, eps);
  return check_launch("groupnorm_fwd_kernel");
}

extern "C" int dpot_groupnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                                  const float* gamma, const float* add, float* dx, float* dgamma, float* dbeta,
                                  float* part, float* workspace, int B, int T, int E, int G, dpot_stream_t stream) {
  DPOT_REQUIRE(dy && x && mean && rstd && gamma && dx && part && ((dgamma == nullptr) == (dbeta == nullptr)),
               "groupnorm_bwd: null pointer");
  DPOT_REQUIRE(B > 0 && T > 0 && E > 0 && G > 0 && E % G == 0 && B <= 65535, "groupnorm_bwd: bad shape");
  const bool vec = ((E / G) % 4 == 0) && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(gamma) &&
                   (add == nullptr || aligned16(add));
  const int items = vec ? gn_cached_items(T, E, G) : 0;
  const GnChunks ck = vec && workspace && aligned16(workspace) ? gn_chunking(B, T, E, G) : GnChunks{0, 0};
  if (ck.chunks) {
    const dim3 grid(ck.chunks, G, B);
    hipLaunchKernelGGL(gn_chunk_bwd_part_kernel, grid, dim3(GN_THREADS), 0, as_stream(stream), dy, x, mean, rstd, workspace,
                       T, E, G, ck);
    int rc0 = check_launch("gn_chunk_bwd_part_kernel");
    if (rc0) return rc0;
    hipLaunchKernelGGL(gn_chunk_bwd_apply_kernel, grid, dim3(GN_THREADS), 0, as_stream(stream), dy, x, mean, rstd, gamma, add,
                       dx, part, (const float*)workspace, B, T, E, G, ck);
  } else if (items == 4)
    hipLaunchKernelGGL((groupnorm_bwd_cached_kernel<4, false>), dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), dy, x, mean,
                       rstd, gamma, add, dx, part, B, T, E, G, GnBwdPacks{nullptr, nullptr, nullptr});
  else if (items == 8)
    hipLaunchKernelGGL((groupnorm_bwd_cached_kernel<8, false>), dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), dy, x, mean,
                       rstd, gamma, add, dx, part, B, T, E, G, GnBwdPacks{nullptr, nullptr, nullptr});
  else if (vec)
    hipLaunchKernelGGL(groupnorm_bwd_kernel<4>, dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), dy, x, mean, rstd,
                       gamma, add, dx, part, B, T, E, G);
  else
    hipLaunchKernelGGL(groupnorm_bwd_kernel<1>, dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), dy, x, mean, rstd,
                       gamma, add, dx, part, B, T, E, G);
  int rc = check_launch("groupnorm_bwd_kernel");
  if (rc || !dgamma) return rc;          // dgamma == dbeta == NULL: the caller reduces `part` later (dpot_groupnorm_param_grads)
  GnParamJobs jobs{};
  jobs.part[0] = part; jobs.dgamma[0] = dgamma; jobs.dbeta[0] = dbeta;
  hipLaunchKernelGGL(groupnorm_param_grad_kernel, dim3(cdiv(E, 64), 1), dim3(256), 0, as_stream(stream), jobs, B, E);
  return check_launch("groupnorm_param_grad_kernel");
}

// dpot_groupnorm_bwd (partials left for dpot_groupnorm_param_grads / dpot_block_finalize) that ALSO writes dx as the bf16 operand
// packs + per-sample column sums of the previous block's channel-MLP backward (see groupnorm_bwd_cached_kernel<.., PK>)
// (returns the number of column-sum partial ROWS per sample the call writes: 1 for the one-workgroup-per-slab kernel, the number
// of chunks for the chunked one; 0 = not supported)
extern "C" int dpot_groupnorm_bwd_packs_rows(int B, int T, int E, int G) {
  if (B <= 0 || T <= 0 || E <= 0 || G <= 0 || E % G || T % 32 || E % 32) return 0;
  const int cg = E / G;
  const int items = gn_cached_items(T, E, G);
  if (cg == 128 && (items == 4 || items == 8)) return 1;
  const GnChunks ck = gn_chunking(B, T, E, G);
  if (ck.chunks && cg % 32 == 0 && cg <= 256 && ck.TC % 32 == 0) return ck.chunks;
  return 0;
}
extern "C" int dpot_groupnorm_bwd_packs_supported(int T, int E, int G) {
  if (T <= 0 || E <= 0 || G <= 0 || E % G || E / G != 128 || T % 32 || E % 32) return 0;
  const int items = gn_cached_items(T, E, G);
  return (items == 4 || items == 8) ? 1 : 0;
}
extern "C" int dpot_groupnorm_bwd_packs(const float* dy, const float* x, const float* mean, const float* rstd,
                                        const float* gamma, const float* add, float* dx, float* part, void* dx_rows_bf16,
                                        void* dx_trans_bf16, float* dx_colsum, float* workspace, int B, int T, int E, int G,
                                        dpot_stream_t stream) {
  DPOT_REQUIRE(dy && x && mean && rstd && gamma && dx && part && dx_rows_bf16 && dx_trans_bf16 && dx_colsum,
               "groupnorm_bwd_packs: null pointer");
  const int rows = dpot_groupnorm_bwd_packs_rows(B, T, E, G);
  DPOT_REQUIRE(B > 0 && B <= 65535 && rows > 0,
               "groupnorm_bwd_packs: needs T %% 32 == 0 and either 128 channels per group in a slab the cached kernel holds or a "
               "chunked slab with chunks of a multiple of 32 tokens (dpot_groupnorm_bwd_packs_rows)");
  DPOT_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(gamma) && aligned16(add) && aligned16(dx_rows_bf16) &&
                   aligned16(dx_trans_bf16),
               "groupnorm_bwd_packs: pointers must be 16-byte aligned");
  const GnBwdPacks pk{reinterpret_cast<uint4*>(dx_rows_bf16), reinterpret_cast<uint4*>(dx_trans_bf16), dx_colsum};
  if (dpot_groupnorm_bwd_packs_supported(T, E, G)) {
    if (gn_cached_items(T, E, G) == 4)
      hipLaunchKernelGGL((groupnorm_bwd_cached_kernel<4, true>), dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), dy, x, mean,
                         rstd, gamma, add, dx, part, B, T, E, G, pk);
    else
      hipLaunchKernelGGL((groupnorm_bwd_cached_kernel<8, true>), dim3(G, B), dim3(GN_THREADS), 0, as_stream(stream), dy, x, mean,
                         rstd, gamma, add, dx, part, B, T, E, G, pk);
    return check_launch("groupnorm_bwd_cached_kernel");
  }
  DPOT_REQUIRE(workspace && aligned16(workspace), "groupnorm_bwd_packs: the chunked form needs the workspace of dpot_groupnorm_ws_elems");
  const GnChunks ck = gn_chunking(B, T, E, G);
  const dim3 grid(ck.chunks, G, B);
  hipLaunchKernelGGL(gn_chunk_bwd_part_kernel, grid, dim3(GN_THREADS), 0, as_stream(stream), dy, x, mean, rstd, workspace, T, E,
                     G, ck);
  int rc0 = check_launch("gn_chunk_bwd_part_kernel");
  if (rc0) return rc0;
  hipLaunchKernelGGL(gn_chunk_bwd_apply_pk_kernel, grid, dim3(GN_THREADS), 0, as_stream(stream), dy, x, mean, rstd, gamma, add,
                     dx, part, (const float*)workspace, B, T, E, G, ck, pk);
  return check_launch("gn_chunk_bwd_apply_pk_kernel");
}

extern "C" int dpot_groupnorm_param_grads(const float* const* parts, float* const* dgammas, float* const* dbetas,
                                          int njobs, int B, int E, dpot_stream_t stream) {
  DPOT_REQUIRE(parts && dgammas && dbetas && njobs > 0 && njobs <= 4 && B > 0 && E > 0, "groupnorm_param_grads: bad argument");
  GnParamJobs jobs{};
  for (int i = 0; i < njobs; ++i) {
    DPOT_REQUIRE(parts[i] && dgammas[i] && dbetas[i], "groupnorm_param_grads: null pointer in job %d", i);
    jobs.part[i] = parts[i]; jobs.dgamma[i] = dgammas[i]; jobs.dbeta[i] = dbetas[i];
  }
  hipLaunchKernelGGL(groupnorm_param_grad_kernel, dim3(cdiv(E, 64), njobs), dim3(256), 0, as_stream(stream), jobs, B, E);
  return check_launch("groupnorm_param_grad_kernel");
}
