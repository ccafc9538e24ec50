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
  if (C == 4 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)mask) & 15u) == 0)) {
    // the DPOT case: one float4 = the 4 channels of a grid point.  Thread (tc = lane & 3, ts) still owns channel tc
    // in the reduction below, but LOADS whole points: lanes 4q..4q+3 read the same float4 and keep their component
    // (one coalesced 16-byte access per point instead of four 4-byte ones; two points in flight per trip)
    const float4* xb = reinterpret_cast<const float4*>(x) + (long long)b * S;
    const float4* yb = reinterpret_cast<const float4*>(y) + (long long)b * S;
    const float4* mb = mask ? reinterpret_cast<const float4*>(mask) + (long long)b * (S / Tt) : nullptr;
    auto comp = [&](const float4& v) __attribute__((always_inline)) {
      return tc == 0 ? v.x : tc == 1 ? v.y : tc == 2 ? v.z : v.w;
    };
    auto one = [&](int s2, const float4& xv, const float4& yv4) __attribute__((always_inline)) {
      const float m = mb ? comp(mb[s2 / Tt]) : 1.f;
      const float yv = comp(yv4) * m;
      const float d = comp(xv) * m - yv;
      d2 += (double)d * d;
      y2 += (double)yv * yv;
      if ((s2 % Tt) == 0) ms += m;
    };
    int s2 = s_beg + ts;
    for (; s2 + L.TS < s_end; s2 += 2 * L.TS) {
      const float4 x0 = xb[s2], y0 = yb[s2], x1 = xb[s2 + L.TS], y1 = yb[s2 + L.TS];
      one(s2, x0, y0);
      one(s2 + L.TS, x1, y1);
    }
    if (s2 < s_end) one(s2, xb[s2], yb[s2]);
  } else if (tc < C) {
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
  // fixed-shape tree over the row lanes (a serial 256-long loop of double LDS reads by C threads cost 15 of 20 us)
  for (int half = L.TS >> 1; half >= 1; half >>= 1) {
    if (ts < half) {
      const int o = half * L.CP;
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
      red[2][threadIdx.x] += red[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (ts == 0 && tc < C) {
    const double a = red[0][tc], bq = red[1][tc], cq = red[2][tc];
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
  // phase 1: one thread per (sample, channel) combines the chunk partials (16-byte loads, fixed order) and leaves the
  // channel's loss term in the 4th slot of its stats entry; phase 2: one thread per sample adds its channels in order.
  // (One thread per SAMPLE walking C * nchunk * 3 dependent 4-byte loads took 5-11 us.)
  for (int idx = threadIdx.x; idx < B * C; idx += blockDim.x) {
    const int b = idx / C, c = idx - b * C;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    const float4* pp = reinterpret_cast<const float4*>(part) + ((long long)b * nchunk) * C + c;
#pragma unroll 4
    for (int k = 0; k < nchunk; ++k) {
      const float4 v = pp[(long long)k * C];
      a0 += (double)v.x;
      a1 += (double)v.y;
      a2 += (double)v.z;
    }
    const float s0 = (float)a0, s1 = (float)a1;
    reinterpret_cast<float4*>(stats)[idx] = make_float4(s0, s1, (float)a2, sqrtf(s0) / (sqrtf(s1) + 1e-8f));
  }
  __syncthreads();
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    double s = 0.0;
    int nch = 0;
    for (int c = 0; c < C; ++c) {
      const float4 st = reinterpret_cast<const float4*>(stats)[(long long)b * C + c];
      s += (double)st.w;
      if (st.z != 0.f) ++nch;
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
  if (C == 4 && total < (1ll << 31) && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)mask) & 15u) == 0)) {
    // the DPOT case: one float4 per grid point, 32-bit index arithmetic (the generic loop below divides 64-bit
    // integers three times per element: ~300 VALU instructions, 25 MB in 14 us)
    const unsigned n4 = (unsigned)(total >> 2), S_u = (unsigned)S;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < n4; q += gridDim.x * 256u) {
      const unsigned b = q / S_u, s = q - b * S_u;
      const float* st = stats + (long long)b * 16;          // [c][4] = {d2, y2, msum, -}
      int nch = 4;
      if (mask) nch = (st[2] != 0.f) + (st[6] != 0.f) + (st[10] != 0.f) + (st[14] != 0.f);
      const float4 m = mask ? reinterpret_cast<const float4*>(mask)[(long long)b * (S / Tt) + s / Tt]
                            : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 xv = reinterpret_cast<const float4*>(x)[q], yv = reinterpret_cast<const float4*>(y)[q];
      auto one = [&](float xe, float ye, float me, int c) __attribute__((always_inline)) {
        const float dn = sqrtf(st[4 * c]);
        const float yn = sqrtf(st[4 * c + 1]) + 1e-8f;
        return dn > 0.f ? g * ((xe - ye) * me * me) / (dn * yn * (float)nch) : 0.f;
      };
      reinterpret_cast<float4*>(dx)[q] = make_float4(one(xv.x, yv.x, m.x, 0), one(xv.y, yv.y, m.y, 1),
                                                     one(xv.z, yv.z, m.z, 2), one(xv.w, yv.w, m.w, 3));
    }
    return;
  }
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
  const long long st = (long long)gridDim.x * 256;
  long long i = blockIdx.x * 256ll + threadIdx.x;
  auto sq = [](const float4 v) __attribute__((always_inline)) {
    return (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  };
  for (; i + 3 * st < n4; i += 4 * st) {   // four loads in flight per thread, summed in the order of the one-at-a-time loop
    const float4 v0 = g4[i], v1 = g4[i + st], v2 = g4[i + 2 * st], v3 = g4[i + 3 * st];
    s += sq(v0);
    s += sq(v1);
    s += sq(v2);
    s += sq(v3);
  }
  for (; i < n4; i += st) s += sq(g4[i]);
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
  const AdamCoef c = adam_coef(hyper, sumsq, grad_scale);
  // one float per lane and array.  (Round 4 tried 16 bytes per lane and stream - seven 1 KiB-per-wave streams: DPOT-M's
  // 110 M parameters 689 -> 778 us, DPOT-Tiny unchanged at 34 us = 6.2 TB/s; profiles/r04_step_census_M_bf16_bd_first.txt)
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float pv = p[i], mv = m[i], vv = v[i];
    adam_update(c, pv, g[i], mv, vv);
    m[i] = mv;
    v[i] = vv;
    p[i] = pv;
  }
}

// host side of a step without a host buffer: all hyper-parameters travel BY VALUE in the kernarg segment of this
// one-thread launch (copied at enqueue time), and the step counter / bias corrections live on the device - so a host
// that runs several graph replays ahead of the GPU can never overwrite the values a queued step still has to read
// (the betas arrive as DOUBLES: utils/optimizer.py:140-141 forms 1 - beta^step in Python floats; with beta2 rounded to
// fp32 first, the correction at step 1 is off by 1.3e-5 relative)
__global__ void adam_stage_kernel(float* __restrict__ hyper, long long* __restrict__ step, float lr, double b1,
                                  double b2, float eps, float wd, float max_norm, int advance) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long long t = step[0] + advance;
  step[0] = t;
  hyper[0] = lr; hyper[1] = (float)b1; hyper[2] = (float)b2; hyper[3] = eps; hyper[4] = wd;
  hyper[5] = (float)(1.0 - pow(b1, (double)t));
  hyper[6] = (float)(1.0 - pow(b2, (double)t));
  hyper[7] = max_norm;
}

// ---- noise injection ----------------------------------------------------------------------------------------
// grid (NCH, B): sum of squares per channel over one chunk of the (X,Y,T) axis of sample b -> part[b][chunk][c].
// rng (may be NULL) = {seed, offset} of the fused-noise path: block (0,0) advances the offset here, one kernel
// BEFORE the kernel that draws from it, so that every block of that kernel sees the same new value.
__global__ __launch_bounds__(256) void chan_sumsq_part_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                              int S, int C, int nch,
                                                              unsigned long long* __restrict__ rng) {
  __shared__ float red[256 * 4];
  const int b = blockIdx.y, ch = blockIdx.x;
  if (rng && b == 0 && ch == 0 && threadIdx.x == 0) rng[1] += 1ull;
  const int per = (S + nch - 1) / nch;
  const int s0 = ch * per, s1 = min(S, s0 + per);
  const float* xb = x + (long long)b * S * C;
  if (C == 4 && (((uintptr_t)xb) & 15u) == 0) {   // the DPOT case: one float4 = the 4 channels of a grid point
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a, a2 = a, a3 = a;
    const float4* x4 = reinterpret_cast<const float4*>(xb);
    int s = s0 + threadIdx.x;
    for (; s + 768 < s1; s += 1024) {      // four loads in flight per thread (one at a time is latency bound: 0.9 TB/s)
      const float4 v = x4[s], v1 = x4[s + 256], v2 = x4[s + 512], v3 = x4[s + 768];
      a.x = fmaf(v.x, v.x, a.x); a.y = fmaf(v.y, v.y, a.y); a.z = fmaf(v.z, v.z, a.z); a.w = fmaf(v.w, v.w, a.w);
      a1.x = fmaf(v1.x, v1.x, a1.x); a1.y = fmaf(v1.y, v1.y, a1.y); a1.z = fmaf(v1.z, v1.z, a1.z); a1.w = fmaf(v1.w, v1.w, a1.w);
      a2.x = fmaf(v2.x, v2.x, a2.x); a2.y = fmaf(v2.y, v2.y, a2.y); a2.z = fmaf(v2.z, v2.z, a2.z); a2.w = fmaf(v2.w, v2.w, a2.w);
      a3.x = fmaf(v3.x, v3.x, a3.x); a3.y = fmaf(v3.y, v3.y, a3.y); a3.z = fmaf(v3.z, v3.z, a3.z); a3.w = fmaf(v3.w, v3.w, a3.w);
    }
    for (; s < s1; s += 256) {
      const float4 v = x4[s];
      a.x = fmaf(v.x, v.x, a.x); a.y = fmaf(v.y, v.y, a.y); a.z = fmaf(v.z, v.z, a.z); a.w = fmaf(v.w, v.w, a.w);
    }
    a.x = (a.x + a1.x) + (a2.x + a3.x); a.y = (a.y + a1.y) + (a2.y + a3.y);
    a.z = (a.z + a1.z) + (a2.z + a3.z); a.w = (a.w + a1.w) + (a2.w + a3.w);
    // fixed-shape reduction (bit-reproducible): xor-butterfly inside each wave, then the 4 wave sums in order
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      a.x += __shfl_xor(a.x, o); a.y += __shfl_xor(a.y, o); a.z += __shfl_xor(a.z, o); a.w += __shfl_xor(a.w, o);
    }
    if ((threadIdx.x & 63) == 0) {
      const int wv = threadIdx.x >> 6;
      red[wv * 4 + 0] = a.x; red[wv * 4 + 1] = a.y; red[wv * 4 + 2] = a.z; red[wv * 4 + 3] = a.w;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      double t = 0.0;
      for (int r = 0; r < 4; ++r) t += (double)red[r * 4 + threadIdx.x];
      part[((long long)b * nch + ch) * 4 + threadIdx.x] = (float)t;
    }
    return;
  }
  const ChanLayout L = chan_layout(C, 256);
  const int tc = threadIdx.x % L.CP, ts = threadIdx.x / L.CP;
  float s2 = 0.f;
  if (tc < C) {
    for (int s = s0 + ts; s < s1; s += L.TS) {
      const float v = xb[(long long)s * C + tc];
      s2 = fmaf(v, v, s2);
    }
  }
  red[threadIdx.x] = s2;
  __syncthreads();
  if (ts == 0 && tc < C) {
    double a = 0.0;
    for (int r = 0; r < L.TS; ++r) a += (double)red[r * L.CP + tc];      // fixed order
    part[((long long)b * nch + ch) * C + tc] = (float)a;
  }
}

// Philox4x32-10 counter-based generator (Salmon et al., SC'11) + Box-Muller: 4 standard normals per call
__device__ __forceinline__ float4 philox_normal4(unsigned long long seed, unsigned long long offset,
                                                 unsigned long long index) {
  unsigned c0 = (unsigned)index, c1 = (unsigned)(index >> 32), c2 = (unsigned)offset, c3 = (unsigned)(offset >> 32);
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float s = 2.3283064365386963e-10f;              // 2^-32
  const float u0 = fmaf((float)c0, s, 0.5f * s), u1 = (float)c1 * s;
  const float u2 = fmaf((float)c2, s, 0.5f * s), u3 = (float)c3 * s;
  // -2 ln u = -2 ln2 * log2 u (v_log_f32); v_sin_f32 / v_cos_f32 take their argument in revolutions, u1 in [0, 1)
  const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));
  const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
  const float s0 = __builtin_amdgcn_sinf(u1), q0 = __builtin_amdgcn_cosf(u1);
  const float s1 = __builtin_amdgcn_sinf(u3), q1 = __builtin_amdgcn_cosf(u3);
  return make_float4(r0 * q0, r0 * s0, r1 * q1, r1 * s1);
}

// grid (G, B): norms[b][c] = sqrt(sum_chunks part) (fixed order), out = x + noise_scale * norm[c] * eps
__global__ __launch_bounds__(256) void noise_axpy_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                         const float* __restrict__ part, float* __restrict__ norms,
                                                         float* __restrict__ out, float noise_scale, int S, int C,
                                                         int nch, const unsigned long long* __restrict__ rng) {
  extern __shared__ float snorm[];
  const int b = blockIdx.y;
  for (int c = threadIdx.x; c < C; c += 256) {
    double a = 0.0;
    for (int k = 0; k < nch; ++k) a += (double)part[((long long)b * nch + k) * C + c];
    const float n = sqrtf((float)a);
    snorm[c] = noise_scale * n;
    if (blockIdx.x == 0) norms[(long long)b * C + c] = n;
  }
  __syncthreads();
  const long long n = (long long)S * C;
  const float* xb = x + b * n;
  float* ob = out + b * n;
  if (eps == nullptr) {   // fused generator: no eps tensor is ever written or read (host guarantees n % 4 == 0, 16-B)
    const unsigned long long seed = rng[0], offset = rng[1];
    const unsigned long long base = (unsigned long long)b * (unsigned long long)(n / 4);
    if (C == 4) {        // the DPOT case: one float4 = the 4 channels of a grid point (no index arithmetic at all)
      const float4 sn = make_float4(snorm[0], snorm[1], snorm[2], snorm[3]);
      // two float4 per thread and trip, both loads issued before the generator (one dependent load per ~200-instruction
      // trip leaves the kernel latency bound)
      const long long n4 = n / 4;
      for (long long q = blockIdx.x * 512ll + threadIdx.x; q < n4; q += (long long)gridDim.x * 512) {
        const long long q1 = q + 256;
        const bool two = q1 < n4;
        const float4 xv = reinterpret_cast<const float4*>(xb)[q];
        const float4 xw = reinterpret_cast<const float4*>(xb)[two ? q1 : q];
        const float4 ev = philox_normal4(seed, offset, base + q);
        reinterpret_cast<float4*>(ob)[q] =
            make_float4(fmaf(sn.x, ev.x, xv.x), fmaf(sn.y, ev.y, xv.y), fmaf(sn.z, ev.z, xv.z), fmaf(sn.w, ev.w, xv.w));
        if (two) {
          const float4 ew = philox_normal4(seed, offset, base + q1);
          reinterpret_cast<float4*>(ob)[q1] =
              make_float4(fmaf(sn.x, ew.x, xw.x), fmaf(sn.y, ew.y, xw.y), fmaf(sn.z, ew.z, xw.z), fmaf(sn.w, ew.w, xw.w));
        }
      }
      return;
    }
    for (long long q = blockIdx.x * 256ll + threadIdx.x; q < n / 4; q += (long long)gridDim.x * 256) {
      const float4 xv = reinterpret_cast<const float4*>(xb)[q];
      const float4 ev = philox_normal4(seed, offset, base + q);
      const int c0 = (int)((q * 4) % C);
      float4 o;
      o.x = fmaf(snorm[c0], ev.x, xv.x);
      o.y = fmaf(snorm[(c0 + 1) % C], ev.y, xv.y);
      o.z = fmaf(snorm[(c0 + 2) % C], ev.z, xv.z);
      o.w = fmaf(snorm[(c0 + 3) % C], ev.w, xv.w);
      reinterpret_cast<float4*>(ob)[q] = o;
    }
    return;
  }
  const float* eb = eps + b * n;
  if ((C & 3) == 0 || C == 1 || C == 2 || C == 4) {
    if ((n & 3) == 0 && (C == 1 || C == 2 || (C & 3) == 0) && (((uintptr_t)xb | (uintptr_t)eb | (uintptr_t)ob) & 15u) == 0) {
      for (long long q = blockIdx.x * 256ll + threadIdx.x; q < n / 4; q += (long long)gridDim.x * 256) {
        const float4 xv = reinterpret_cast<const float4*>(xb)[q];
        const float4 ev = reinterpret_cast<const float4*>(eb)[q];
        const int c0 = (int)((q * 4) % C);
        float4 o;
        o.x = fmaf(snorm[c0], ev.x, xv.x);
        o.y = fmaf(snorm[(c0 + 1) % C], ev.y, xv.y);
        o.z = fmaf(snorm[(c0 + 2) % C], ev.z, xv.z);
        o.w = fmaf(snorm[(c0 + 3) % C], ev.w, xv.w);
        reinterpret_cast<float4*>(ob)[q] = o;
      }
      return;
    }
  }
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256)
    ob[idx] = fmaf(snorm[(int)(idx % C)], eb[idx], xb[idx]);
}

// ---- backward of the noise injection (AR steps > 0: xx depends on earlier predictions) -----------------------
// y = x + s * n(x) * eps, n = ||x||_2 over (X,Y,T) per (b,c)  ->  dx = g + s * x / n * sum_s(g * eps)
// grid (NCH, B): part[b][chunk][c] = sum over the chunk of g * eps; eps is read, or re-drawn from the generator
// state {seed, offset} the forward used (rng != NULL; needs S*C % 4 == 0, like the forward)
__global__ __launch_bounds__(256) void noise_bwd_part_kernel(const float* __restrict__ g, const float* __restrict__ eps,
                                                             const unsigned long long* __restrict__ rng,
                                                             float* __restrict__ part, int S, int C, int nch) {
  __shared__ float red[256];
  const int b = blockIdx.y, ch = blockIdx.x;
  const long long n = (long long)S * C;
  const long long per = ((n / 4 + nch - 1) / nch) * 4;          // chunk of the flattened (s, c) axis, multiple of 4
  const long long i0 = ch * per, i1 = min(n, i0 + per);
  const float* gb = g + b * n;
  // one pass per channel (C is the handful of PDE field components): threads stride over the chunk's grid points,
  // the 256 partial sums are combined in a fixed order
  for (int c = 0; c < C; ++c) {
    float a = 0.f;
    // elements of channel c inside [i0, i1): idx = s*C + c
    long long sfirst = (i0 - c + C - 1) / C;
    if (sfirst < 0) sfirst = 0;
    for (long long sidx = sfirst + threadIdx.x; sidx * C + c < i1; sidx += 256) {
      const long long idx = sidx * C + c;
      float e;
      if (rng) {
        const float4 ev = philox_normal4(rng[0], rng[1], (unsigned long long)b * (unsigned long long)(n / 4) + idx / 4);
        const int k = (int)(idx & 3);
        e = k == 0 ? ev.x : k == 1 ? ev.y : k == 2 ? ev.z : ev.w;
      } else {
        e = eps[b * n + idx];
      }
      a = fmaf(gb[idx], e, a);
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int r = 0; r < 256; ++r) t += (double)red[r];         // fixed order
      part[((long long)b * nch + ch) * C + c] = (float)t;
    }
    __syncthreads();
  }
}
// grid (G, B): dx = g + s * x / n[b,c] * sum_chunks part[b][.][c]
__global__ __launch_bounds__(256) void noise_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                              const float* __restrict__ norms,
                                                              const float* __restrict__ part, float* __restrict__ dx,
                                                              float noise_scale, int S, int C, int nch) {
  extern __shared__ float coef[];
  const int b = blockIdx.y;
  for (int c = threadIdx.x; c < C; c += 256) {
    double a = 0.0;
    for (int k = 0; k < nch; ++k) a += (double)part[((long long)b * nch + k) * C + c];
    const float nrm = norms[(long long)b * C + c];
    coef[c] = nrm > 1e-30f ? noise_scale * (float)a / nrm : 0.f;
  }
  __syncthreads();
  const long long n = (long long)S * C;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256)
    dx[b * n + idx] = fmaf(coef[(int)(idx % C)], x[b * n + idx], g[b * n + idx]);
}

// ---- AR window slide (train_temporal.py:219: xx = cat(xx[..., T_bundle:, :], im)) -----------------------------
// rows = B*X*Y; xx [rows, T, C], im [rows, Tb, C] -> out [rows, T, C]
__global__ __launch_bounds__(256) void window_slide_kernel(const float* __restrict__ xx, const float* __restrict__ im,
                                                           float* __restrict__ out, long long rows, int T, int Tb,
                                                           int C) {
  const int TC = T * C, keep = (T - Tb) * C, TbC = Tb * C;
  const long long total = rows * TC;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long r = idx / TC;
    const int j = (int)(idx - r * TC);
    out[idx] = j < keep ? xx[r * TC + j + TbC] : im[r * TbC + (j - keep)];
  }
}
__global__ __launch_bounds__(256) void window_slide_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dxx,
                                                               float* __restrict__ dim, long long rows, int T, int Tb,
                                                               int C) {
  const int TC = T * C, keep = (T - Tb) * C, TbC = Tb * C;
  const long long total = rows * TC;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long r = idx / TC;
    const int j = (int)(idx - r * TC);
    if (dxx) dxx[idx] = j >= TbC ? dout[r * TC + j - TbC] : 0.f;
    if (dim && j >= keep) dim[r * TbC + (j - keep)] = dout[idx];
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
  long long n = ((long long)S * C + 4095) / 4096;       // 1024 points of 4 channels per 1024-thread block and trip pair
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

extern "C" int dpot_adam_stage(float* hyper, int64_t* step, float lr, double beta1, double beta2, float eps,
                               float weight_decay, float max_norm, int advance, dpot_stream_t stream) {
  DPOT_REQUIRE(hyper && step, "adam_stage: null pointer");
  DPOT_REQUIRE(advance >= 0, "adam_stage: advance must be >= 0");
  hipLaunchKernelGGL(adam_stage_kernel, dim3(1), dim3(64), 0, as_stream(stream), hyper,
                     reinterpret_cast<long long*>(step), lr, beta1, beta2, eps, weight_decay, max_norm, advance);
  return check_launch("adam_stage_kernel");
}

extern "C" int dpot_noise_chunks(int S, int C) {
  long long n = ((long long)S * C + 8191) / 8192;
  if (n > 64) n = 64;
  if (n < 1) n = 1;
  return (int)n;
}

static int noise_launch(const float* xx, const float* eps, float* out, float* norms, unsigned long long* rng,
                        float noise_scale, int B, int S, int C, dpot_stream_t stream) {
  const int nch = dpot_noise_chunks(S, C);
  float* part = norms + (size_t)B * C;
  hipLaunchKernelGGL(chan_sumsq_part_kernel, dim3(nch, B), dim3(256), 0, as_stream(stream), xx, part, S, C, nch, rng);
  int rc = check_launch("chan_sumsq_part_kernel");
  if (rc) return rc;
  // 2 float4 per thread and trip (every thread's loads issue up front); ~2048 workgroups in all (8 per CU, one
  // resident round): every workgroup first re-reduces the norm partials (nch dependent-latency loads), which a grid
  // of one trip per workgroup (10240 of them at DPOT-Tiny B=32) paid five rounds deep
  long long g = ((long long)S * C / 4 + 511) / 512;
  const long long cap = B >= 2048 ? 1 : (2048 + B - 1) / B;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(noise_axpy_kernel, dim3((unsigned)g, B), dim3(256), C * sizeof(float), as_stream(stream), xx, eps,
                     (const float*)part, norms, out, noise_scale, S, C, nch, (const unsigned long long*)rng);
  return check_launch("noise_axpy_kernel");
}

extern "C" int dpot_noise_inject_bwd(const float* xx, const float* eps, const uint64_t* rng_state, const float* g,
                                     const float* norms, float* dx, float* part, float noise_scale, int B, int S, int C,
                                     dpot_stream_t stream) {
  DPOT_REQUIRE(xx && g && norms && dx && part && B > 0 && S > 0 && C > 0 && C <= 1024, "noise_inject_bwd: bad argument");
  DPOT_REQUIRE((eps != nullptr) != (rng_state != nullptr), "noise_inject_bwd: exactly one of eps / rng_state");
  DPOT_REQUIRE(B <= 65535, "noise_inject_bwd: batch too large");
  DPOT_REQUIRE(rng_state == nullptr || ((long long)S * C) % 4 == 0, "noise_inject_bwd: rng path needs S*C %% 4 == 0");
  const int nch = dpot_noise_chunks(S, C);
  hipLaunchKernelGGL(noise_bwd_part_kernel, dim3(nch, B), dim3(256), 0, as_stream(stream), g, eps,
                     reinterpret_cast<const unsigned long long*>(rng_state), part, S, C, nch);
  int rc = check_launch("noise_bwd_part_kernel");
  if (rc) return rc;
  long long gr = ((long long)S * C + 255) / 256;
  if (gr > 256) gr = 256;
  hipLaunchKernelGGL(noise_bwd_apply_kernel, dim3((unsigned)gr, B), dim3(256), C * sizeof(float), as_stream(stream), xx,
                     g, norms, (const float*)part, dx, noise_scale, S, C, nch);
  return check_launch("noise_bwd_apply_kernel");
}

extern "C" int dpot_window_slide(const float* xx, const float* im, float* out, int64_t rows, int T, int Tb, int C,
                                 dpot_stream_t stream) {
  DPOT_REQUIRE(xx && im && out && rows > 0 && T > 0 && Tb > 0 && Tb <= T && C > 0, "window_slide: bad argument");
  hipLaunchKernelGGL(window_slide_kernel, dim3(grid_for(rows * T * C, 8192)), dim3(256), 0, as_stream(stream), xx, im,
                     out, (long long)rows, T, Tb, C);
  return check_launch("window_slide_kernel");
}

extern "C" int dpot_window_slide_bwd(const float* dout, float* dxx, float* dim, int64_t rows, int T, int Tb, int C,
                                     dpot_stream_t stream) {
  DPOT_REQUIRE(dout && (dxx || dim) && rows > 0 && T > 0 && Tb > 0 && Tb <= T && C > 0, "window_slide_bwd: bad argument");
  hipLaunchKernelGGL(window_slide_bwd_kernel, dim3(grid_for(rows * T * C, 8192)), dim3(256), 0, as_stream(stream), dout,
                     dxx, dim, (long long)rows, T, Tb, C);
  return check_launch("window_slide_bwd_kernel");
}

extern "C" int dpot_noise_inject(const float* xx, const float* eps, float* out, float* norms, float noise_scale,
                                 int B, int S, int C, dpot_stream_t stream) {
  DPOT_REQUIRE(xx && eps && out && norms && B > 0 && S > 0 && C > 0 && C <= 1024, "noise_inject: bad argument");
  DPOT_REQUIRE(B <= 65535, "noise_inject: batch too large");
  return noise_launch(xx, eps, out, norms, nullptr, noise_scale, B, S, C, stream);
}

extern "C" int dpot_noise_inject_rng(const float* xx, float* out, float* norms, uint64_t* rng_state, float noise_scale,
                                     int B, int S, int C, dpot_stream_t stream) {
  DPOT_REQUIRE(xx && out && norms && rng_state && B > 0 && S > 0 && C > 0 && C <= 1024, "noise_inject_rng: bad argument");
  DPOT_REQUIRE(B <= 65535, "noise_inject_rng: batch too large");
  DPOT_REQUIRE(((long long)S * C) % 4 == 0 && aligned16(xx) && aligned16(out),
               "noise_inject_rng: needs S*C %% 4 == 0 and 16-byte aligned fields");
  return noise_launch(xx, nullptr, out, norms, reinterpret_cast<unsigned long long*>(rng_state), noise_scale, B, S, C,
                      stream);
}
