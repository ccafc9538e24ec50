// afno_fused.hip - SURVEY 8 f4: one AFNO layer's forward in ONE launch (models/dpot.py:59-102 and :165-175):
//
//      GroupNorm1 statistics -> rfft2 -> block-diagonal complex 2-layer MLP -> irfft2 -> + x_orig -> GroupNorm2
//
// The three-launch form (gn_dft.hip gn_rfft2 -> afno_mlp.hip afno_mlp3 -> gn_dft.hip irfft2_gn) writes the spectrum S and
// the mixer output O2 to HBM and reads them (and x) back; here a (sample, channel block) workgroup keeps everything on chip.
// 16 x 16 latent grid, bs = 128 channels per block, every mode kept (mx = 16, my = 9): DPOT-Tiny / -Small / -Medium at 128^2.
//
//   * a workgroup = 8 waves = one sample x one channel block (128 channels = 1 or 2 GroupNorm groups: both norms are
//     workgroup-local).  Wave w owns channels 16 w .. 16 w + 15; lane (c = lane & 15, q = lane >> 4) owns channel c and
//     the four token rows q, q + 4, q + 8, q + 12 - 64 values in registers.
//   * rfft2 entirely in registers: 16-point row FFTs per lane (fft_regs, compile-time twiddles), the 16-point column FFT
//     as 4 x 4 Cooley-Tukey - a 4-point DFT in the lane, a lane-dependent twiddle, a 4-point DFT ACROSS the four lanes of
//     a channel with v_permlane32_swap / v_permlane16_swap (no LDS).  Lane (c, q) ends up with the modes
//     kx = 4 brev2(q) + k1, k1 = 0..3, of all nine ky: 36 complex values.
//   * the spectrum becomes the A operand of the MLP: mode (kx, ky) is row 4 q + k1 of row tile ky - exactly the rows the
//     accumulator layout of v_mfma_f32_16x16x4_f32 gives lane group q back, so the layer-2 OUTPUT of wave w (16 real + 16
//     imaginary columns = its own 16 channels) is already distributed for the inverse transform: no shuffle between the
//     MLP and irfft2.  The operand (144 modes x 256 re|im columns = 144 KiB) sits in LDS as [ky][re|im][K-slab][1 KiB
//     fragment block], first the spectrum, then - after layer 1 - the activated hidden layer in the same place.
//   * the MLP is the three-product form of afno_mlp3_kernel (P1 = Sr Wr, P2 = Si Wi, P3 = (Sr+Si)(Wr+Wi)); wave w owns
//     column tile w of all 144 rows: 108 accumulator registers, 2 waves per SIMD.  The weights never touch LDS: a wave
//     needs only ITS column tile of Wr / Wi (1 KiB per K-slab and part, contiguous in dpot_afno_pack_all's layout 1),
//     loaded straight into registers one slab ahead (L2 resident: every workgroup of a channel block reads the same 256 KiB).
//   * irfft2 mirrors rfft2 (cross-lane 4-point DFT, twiddle, in-lane 4-point DFT, row FFTs), + GroupNorm1(x) re-derived
//     from x, GroupNorm2 statistics through LDS, y1 / xn2 stored.
// Saved for the backward (training): S and the layer-1 pre-activation, in the layouts of the three-launch path - the
// backward kernels are unchanged.  Inference passes S = pre = NULL.
#include <type_traits>

#include "common.h"
#define DPOT_DFT_NO_KERNELS
#include "dft_fast.h"

namespace dpot {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AfnoFusedArgs {
  const float* x;       // [B, 256, E]
  const float* g1;      // GroupNorm1 gamma / beta [E]; g1 == NULL: no norm1 (the mixer alone, AFNO2D)
  const float* b1;
  const float* Wa;      // layout-1 packs (dpot_afno_pack_all): [nb][8 slabs][2 parts][8 col tiles][256]
  const float* ba;      // [nb][256] = [br | bi]
  const float* Wb;
  const float* bb;
  const float* g2;      // GroupNorm2 gamma / beta; g2 == NULL: no norm2 (y1 only)
  const float* b2;
  float* S;             // [B, 16, 9, nb, 2, 128] or NULL
  float* pre;           // same shape or NULL
  float* y1;            // [B, 256, E] or NULL
  float* xn2;           // [B, 256, E] or NULL
  float* mean1;         // [B, G] (with g1)
  float* rstd1;
  float* mean2;         // [B, G] (with g2)
  float* rstd2;
  // optional (with g2; round 5): GroupNorm2(y1) as the two bf16 operand packs of the channel MLP (csrc/gemm_bf16p.hip) -
  // xp: row form [B*256 / 32][E / 16][64 chunks][8] (A operand of fc1), xpT: transposed form [E / 32][B*256 / 16][64][8]
  // (operand of the fc1 weight gradient) - what dpot_bf16_pack_both_norm(y1, statistics) would write, bit for bit
  uint4* xp;
  uint4* xpT;
  int B, E, G, nb, act;
  float eps;
};

constexpr int AF_H = 16, AF_W = 16, AF_WF = 9, AF_BS = 128, AF_NW = 8;
constexpr int AF_TILE = 16 * 256;                 // floats of one row tile of the operand: 16 K-slab blocks of 1 KiB
constexpr int AF_AREG = AF_WF * AF_TILE;          // 144 KiB

// both halves of a lane pair: lo = the value of the lane whose bit 5 (xchg32) / bit 4 (xchg16) is clear, hi = the other's
__device__ __forceinline__ void xchg32(float v, float& lo, float& hi) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  lo = __uint_as_float(r[0]);
  hi = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xchg16(float v, float& lo, float& hi) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  lo = __uint_as_float(r[0]);
  hi = __uint_as_float(r[1]);
}

__device__ __forceinline__ unsigned af_pack2(float lo, float hi) {   // v_cvt_pk_bf16_f32, round to nearest even
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// 4-point DFT in place, SGN = -1 forward / +1 inverse: out[k] = sum_n in[n] e^{SGN 2 pi i n k / 4}
template <int SGN>
__device__ __forceinline__ void dft4(float (&r)[4], float (&i)[4]) {
  const float a0r = r[0] + r[2], a0i = i[0] + i[2], a1r = r[0] - r[2], a1i = i[0] - i[2];
  const float a2r = r[1] + r[3], a2i = i[1] + i[3], a3r = r[1] - r[3], a3i = i[1] - i[3];
  r[0] = a0r + a2r;
  i[0] = a0i + a2i;
  r[2] = a0r - a2r;
  i[2] = a0i - a2i;
  // SGN i * (a3r + i a3i) = SGN (-a3i + i a3r)
  constexpr float s = (float)SGN;
  r[1] = a1r - s * a3i;
  i[1] = a1i + s * a3r;
  r[3] = a1r + s * a3i;
  i[3] = a1i - s * a3r;
}

// PK: also emit GroupNorm2's output as the channel MLP's bf16 operand packs (a separate instantiation: the extra live values cost
// the plain form 4 spilled registers and ~2 us)
template <int CG, int ACTK, bool PK>
__global__ __launch_bounds__(64 * AF_NW) void afno_fused_fwd_kernel(const AfnoFusedArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[AF_AREG + 128];
  float* const A = lds;                                          // operand region
  double* const shd = reinterpret_cast<double*>(lds + AF_AREG);  // [2 norms][3 sums][8 waves] = 48 doubles

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int hb = q >> 1, mb = q & 1;
  const float sgn_h = hb ? -1.f : 1.f, sgn_m = mb ? -1.f : 1.f;
  const bool rot = (hb & mb) != 0;
  const int nb = p.nb, E = p.E;
  const int kblk = (int)(blockIdx.x % (unsigned)nb), b = (int)(blockIdx.x / (unsigned)nb);
  const int ch = kblk * AF_BS + 16 * wave + c;                   // this lane's channel
  const int grp = (kblk * AF_BS + 16 * wave) / CG;               // its GroupNorm group (uniform per wave)
  constexpr int WPG = CG / 16;                                   // waves per group
  const int gw0 = (wave / WPG) * WPG;                            // first wave of this wave's group
  const float scale = 1.0f / 16.0f;                              // ortho: 1 / sqrt(16 * 16)

  // lane-dependent twiddles of the 4 x 4 split: e^{-+ 2 pi i q k / 16}, k = 1..3 (cos, sin of the positive angle)
  float tc[4], ts[4];
  tc[0] = 1.f;
  ts[0] = 0.f;
#pragma unroll
  for (int k = 1; k < 4; ++k) sincospif((float)(q * k) * 0.125f, &ts[k], &tc[k]);

  auto bar = [&]() __attribute__((always_inline)) { __syncthreads(); };
#ifdef AF_TIMING   /* kernel experiments only (scripts/afno_layer_phases.py): shader-clock stamps of waves 0 and 7 */
  unsigned long long ts_[12];
  int nts_ = 0;
#define AF_STAMP() ts_[nts_++] = __builtin_amdgcn_s_memtime()
#else
#define AF_STAMP()
#endif
  AF_STAMP();

  // three sums over this wave's GroupNorm group -> (mean, rstd); `slot` = 0 (norm1) / 1 (norm2).  One barrier.
  auto group_stats = [&](double s_m, double s_mm, double s_q, int slot, float& mu_f, double& mu_d, float& rs)
                         __attribute__((always_inline)) {
    s_m = wave_sum_d(s_m);
    s_mm = wave_sum_d(s_mm);
    s_q = wave_sum_d(s_q);
    double* sh = shd + slot * 24;
    if (lane == 0) {
      sh[wave] = s_m;
      sh[8 + wave] = s_mm;
      sh[16 + wave] = s_q;
    }
    bar();
    double a = 0.0, bq = 0.0, cq = 0.0;
#pragma unroll
    for (int i = 0; i < WPG; ++i) {
      a += sh[gw0 + i];
      bq += sh[8 + gw0 + i];
      cq += sh[16 + gw0 + i];
    }
    constexpr double NR = (double)(AF_H * CG), NE = (double)(AF_H * AF_W * CG);
    mu_d = a / NR;
    const double m2 = cq + (double)AF_W * (bq - a * a / NR);
    mu_f = (float)mu_d;
    rs = 1.0f / sqrtf((float)(m2 / NE) + p.eps);
  };

  // ================================ phase A: GroupNorm1 statistics + rfft2 ================================
  const float* __restrict__ xb = p.x + (long long)b * (AF_H * AF_W) * E + kblk * AF_BS + 16 * wave;   // uniform
  const int xlo = q * AF_W * E + c;                              // row q, channel c
  float Sr[4][AF_WF], Si[4][AF_WF];                              // [n1 / k1][ky]
  float mu1 = 0.f, rs1 = 1.f;
  {
    float v[4][AF_W];
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1)
#pragma unroll
      for (int y = 0; y < AF_W; ++y) v[n1][y] = xb[xlo + (n1 * 4 * AF_W + y) * E];
    const float piv = xb[c];                                     // the channel's first token (see gn_rfft2_kernel)
#ifdef AF_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    AF_STAMP();                                                  // 1: x loaded
    double s_m = 0.0, s_mm = 0.0, s_q = 0.0;
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) {
      if (p.g1) {
        float s = 0.f;
#pragma unroll
        for (int y = 0; y < AF_W; ++y) s += v[n1][y];
        const float lm = s * (1.0f / AF_W);
        float qq = 0.f;
#pragma unroll
        for (int y = 0; y < AF_W; ++y) {
          const float d = v[n1][y] - lm;
          qq = fmaf(d, d, qq);
        }
        s_m += (double)lm;
        s_mm += (double)lm * (double)lm;
        s_q += (double)qq;
      }
      float vr[AF_W], vi[AF_W];
#pragma unroll
      for (int y = 0; y < AF_W; ++y) {
        vr[y] = v[n1][y] - piv;
        vi[y] = 0.f;
      }
      fft_regs<AF_W, -1>(vr, vi);
      fft_sfor<0, AF_WF>([&](auto KY) __attribute__((always_inline)) {
        constexpr int ky = decltype(KY)::value;
        Sr[n1][ky] = vr[brev<AF_W>(ky)];
        Si[n1][ky] = vi[brev<AF_W>(ky)];
      });
    }
    // column transform: in-lane DFT4 over n1, twiddle e^{-2 pi i q k1 / 16}, DFT4 across the lanes q
#pragma unroll
    for (int ky = 0; ky < AF_WF; ++ky) {
      float yr[4] = {Sr[0][ky], Sr[1][ky], Sr[2][ky], Sr[3][ky]};
      float yi[4] = {Si[0][ky], Si[1][ky], Si[2][ky], Si[3][ky]};
      dft4<-1>(yr, yi);
#pragma unroll
      for (int k1 = 0; k1 < 4; ++k1) {
        float wr = yr[k1], wi = yi[k1];
        if (k1 > 0) {                                            // (wr + i wi)(tc - i ts)
          const float nr = fmaf(wr, tc[k1], wi * ts[k1]);
          const float ni = fmaf(wi, tc[k1], -wr * ts[k1]);
          wr = nr;
          wi = ni;
        }
        float lo, hi;
        xchg32(wr, lo, hi);
        float tr = fmaf(sgn_h, hi, lo);
        xchg32(wi, lo, hi);
        float ti = fmaf(sgn_h, hi, lo);
        const float rr = rot ? ti : tr, ri = rot ? -tr : ti;     // lanes q = 3: times -i
        xchg16(rr, lo, hi);
        Sr[k1][ky] = fmaf(sgn_m, hi, lo);
        xchg16(ri, lo, hi);
        Si[k1][ky] = fmaf(sgn_m, hi, lo);
      }
    }
    AF_STAMP();                                                  // 2: rfft2 done
    // GroupNorm1 as a scale of the spectrum + a DC term: GN(x) = a (x - piv) + (beta + a (piv - mu))
    float a = 1.f, dcv = piv;
    if (p.g1) {
      double mu_d;
      group_stats(s_m, s_mm, s_q, 0, mu1, mu_d, rs1);
      if (lane == 0 && wave == gw0) {
        p.mean1[b * p.G + grp] = mu1;
        p.rstd1[b * p.G + grp] = rs1;
      }
      a = rs1 * p.g1[ch];
      dcv = fmaf((float)((double)piv - mu_d), a, p.b1[ch]);
    }
    const float mul = a * scale;
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
      for (int ky = 0; ky < AF_WF; ++ky) {
        Sr[k1][ky] *= mul;
        Si[k1][ky] *= mul;
      }
    if (q == 0) Sr[0][0] += dcv * (float)(AF_H * AF_W) * scale;  // (kx, ky) = (0, 0) lives in lane q = 0, k1 = 0
  }

  // operand addressing.  Row r = 4 q + k1 of tile ky, K-column k = 16 s + 4 kq + e of slab s is word
  //   (ky * 16 + s) * 256 + kq * 64 + (r ^ kq) * 4 + e:   b128 fragment reads (lane (fr, fq) reads row fr, k-quad fq) and
  //   both scatter writes below (row 4 q + k1, column c of a wave's slab) are bank-conflict free
  const int kq = c >> 2, ke = c & 3;
  const int wofs = wave * 256 + kq * 64 + 16 * q + ke;           // + 4 * (k1 ^ kq) + ky * AF_TILE (+ 8 * 256: imaginary)
  const int rofs = q * 64 + ((c ^ q) << 2);                      // fragment read: + (ky * 16 + s) * 256
  // S / pre rows of this lane: mode (kx = 4 brev2(q) + k1, ky) -> row (b * 16 + kx) * 9 + ky of [., 2E]
  const int ld2 = 2 * E;
  const long long gbase = ((long long)b * (AF_H * AF_WF)) * ld2 + kblk * (2 * AF_BS) + 16 * wave;   // uniform
  const int glo = (4 * (2 * mb + hb)) * AF_WF * ld2 + c;         // + (k1 * 9 + ky) * ld2 + part * 128

  auto publish = [&](float* __restrict__ gout, bool act1) __attribute__((always_inline)) {
    // Sr / Si [k1][ky] of this lane -> global (optional) -> [activation] -> operand region
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
      for (int ky = 0; ky < AF_WF; ++ky) {
        float vr = Sr[k1][ky], vi = Si[k1][ky];
        if (gout) {
          gout[gbase + glo + (k1 * AF_WF + ky) * ld2] = vr;
          gout[gbase + glo + (k1 * AF_WF + ky) * ld2 + AF_BS] = vi;
        }
        if (act1) {
          vr = ACTK == DPOT_ACT_GELU ? gelu_fwd(vr) : act_fwd(p.act, vr);
          vi = ACTK == DPOT_ACT_GELU ? gelu_fwd(vi) : act_fwd(p.act, vi);
        }
        const int o = wofs + ((k1 ^ kq) << 2) + ky * AF_TILE;
        A[o] = vr;
        A[o + 8 * 256] = vi;
      }
  };
  AF_STAMP();                                                    // 3: statistics known, spectrum scaled
  publish(p.S, false);
  bar();                                                         // spectrum complete
  AF_STAMP();                                                    // 4: spectrum published

  // ================================ the two MLP layers ================================
  f32x4 P1[AF_WF], P2[AF_WF], P3[AF_WF];
  // One layer = 8 K-slabs x 9 row tiles x 12 MFMAs, software-pipelined by hand (hipcc left alone issues every fragment
  // read right in front of its first use and sinks the weight loads to the end of the previous slab: LDS and L2 latency
  // exposed 72 + 8 times per layer): the fragments of tile t + 1 are read while the MFMAs of tile t issue, the Wr / Wi
  // tiles of slab u + 2 are requested at the start of slab u (ring of three register sets); sched_barriers pin the order.
  auto layer = [&](const float* __restrict__ Wl) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < AF_WF; ++i) P1[i] = P2[i] = P3[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // slab u: Wr tile at ((u * 2 + 0) * 8 + wave) * 256, Wi tile at ((u * 2 + 1) * 8 + wave) * 256 floats
    const float* wl = Wl + (long long)kblk * (8 * 2 * 8 * 256) + wave * 256 + lane * 4;
    f32x4 wr[3], wi[3];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      wr[u] = *reinterpret_cast<const f32x4*>(wl + u * (2 * 8 * 256));
      wi[u] = *reinterpret_cast<const f32x4*>(wl + u * (2 * 8 * 256) + 8 * 256);
    }
    f32x4 ar = *reinterpret_cast<const f32x4*>(A + rofs);
    f32x4 ai = *reinterpret_cast<const f32x4*>(A + rofs + 8 * 256);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u + 2 < 8) {
        wr[(u + 2) % 3] = *reinterpret_cast<const f32x4*>(wl + (u + 2) * (2 * 8 * 256));
        wi[(u + 2) % 3] = *reinterpret_cast<const f32x4*>(wl + (u + 2) * (2 * 8 * 256) + 8 * 256);
      }
      const f32x4 br = wr[u % 3], bi = wi[u % 3];
      const f32x4 bs = br + bi;
#pragma unroll
      for (int i = 0; i < AF_WF; ++i) {
        const int un = i + 1 < AF_WF ? u : u + 1, in = i + 1 < AF_WF ? i + 1 : 0;    // the next tile
        f32x4 arn = ar, ain = ai;
        if (un < 8) {
          arn = *reinterpret_cast<const f32x4*>(A + rofs + (in * 16 + un) * 256);
          ain = *reinterpret_cast<const f32x4*>(A + rofs + (in * 16 + 8 + un) * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 as = ar + ai;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          P1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[s2], br[s2], P1[i], 0, 0, 0);
          P2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[s2], bi[s2], P2[i], 0, 0, 0);
          P3[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[s2], bs[s2], P3[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        ar = arn;
        ai = ain;
      }
    }
  };
  // accumulators -> Sr / Si [k1 = e][ky = i] (+ bias): lane (n = lane & 15, q) holds rows 4 q + e of column n
  auto recombine = [&](const float* __restrict__ bias) __attribute__((always_inline)) {
    const float bre = bias ? bias[kblk * (2 * AF_BS) + 16 * wave + c] : 0.f;
    const float bim = bias ? bias[kblk * (2 * AF_BS) + AF_BS + 16 * wave + c] : 0.f;
#pragma unroll
    for (int i = 0; i < AF_WF; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Sr[e][i] = P1[i][e] - P2[i][e] + bre;
        Si[e][i] = P3[i][e] - P1[i][e] - P2[i][e] + bim;
      }
  };

  layer(p.Wa);
  recombine(p.ba);
  AF_STAMP();                                                    // 5: layer 1 done
  bar();                                                         // every wave is done reading the spectrum
  publish(p.pre, true);                                          // pre-activation saved, activated layer-1 output -> operand
  bar();
  AF_STAMP();                                                    // 6: hidden layer published
  layer(p.Wb);
  recombine(p.bb);                                               // O2: this wave's 16 channels, modes as after phase A
  AF_STAMP();                                                    // 7: layer 2 done

  // ================================ phase C: irfft2 + x_orig + GroupNorm2 ================================
  // column transform: DFT4 across the lanes (input n2 = brev2(q)), twiddle e^{+2 pi i n1 q / 16}, in-lane DFT4 over n1
#pragma unroll
  for (int ky = 0; ky < AF_WF; ++ky) {
    float yr[4], yi[4];
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) {
      float lo, hi;
      xchg16(Sr[n1][ky], lo, hi);
      const float tr = fmaf(sgn_m, hi, lo);
      xchg16(Si[n1][ky], lo, hi);
      const float ti = fmaf(sgn_m, hi, lo);
      float er, orr, ei, oi;
      xchg32(tr, er, orr);
      xchg32(ti, ei, oi);
      const float o2r = mb ? -oi : orr, o2i = mb ? orr : oi;     // odd half times i^m
      float wr = fmaf(sgn_h, o2r, er), wi = fmaf(sgn_h, o2i, ei);
      if (n1 > 0) {                                              // (wr + i wi)(tc + i ts)
        const float nr = fmaf(wr, tc[n1], -wi * ts[n1]);
        const float ni = fmaf(wi, tc[n1], wr * ts[n1]);
        wr = nr;
        wi = ni;
      }
      yr[n1] = wr;
      yi[n1] = wi;
    }
    dft4<1>(yr, yi);
    const float wgt = (ky == 0 || ky == AF_W / 2) ? 1.f : 2.f;   // Hermitian column weights of the one-sided inverse
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      Sr[k1][ky] = yr[k1] * wgt;
      Si[k1][ky] = yi[k1] * wgt;
    }
  }
  // rows: lane (c, q) now holds row xr = 4 k1 + q of its channel, half-complex over ky
  // (the addresses of this phase hang off an OPAQUE copy of the lane offset: shared with phase A, the compiler keeps the
  // 64 load addresses alive across both MLP layers - 120 spilled registers)
  int xlo2 = xlo;
  asm volatile("" : "+v"(xlo2));
  float a1 = 1.f, c1 = 0.f;
  if (p.g1) {
    a1 = rs1 * p.g1[ch];
    c1 = p.b1[ch] - mu1 * a1;
  }
  float yv[4][AF_W];
  double s_m = 0.0, s_mm = 0.0, s_q = 0.0;
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    float ur[AF_W], ui[AF_W];
#pragma unroll
    for (int ky = 0; ky < AF_W; ++ky) {
      ur[ky] = ky < AF_WF ? Sr[k1][ky] : 0.f;
      ui[ky] = ky < AF_WF ? Si[k1][ky] : 0.f;
    }
    fft_regs<AF_W, 1>(ur, ui);
    float xq[AF_W];
#pragma unroll
    for (int y = 0; y < AF_W; ++y) xq[y] = xb[xlo2 + (k1 * 4 * AF_W + y) * E];
    fft_sfor<0, AF_W>([&](auto YY) __attribute__((always_inline)) {
      constexpr int yy = decltype(YY)::value;
      yv[k1][yy] = fmaf(ur[brev<AF_W>(yy)], scale, fmaf(xq[yy], a1, c1));
    });
    if (p.g2) {
      float s = 0.f;
#pragma unroll
      for (int y = 0; y < AF_W; ++y) s += yv[k1][y];
      const float lm = s * (1.0f / AF_W);
      float qq = 0.f;
#pragma unroll
      for (int y = 0; y < AF_W; ++y) {
        const float d = yv[k1][y] - lm;
        qq = fmaf(d, d, qq);
      }
      s_m += (double)lm;
      s_mm += (double)lm * (double)lm;
      s_q += (double)qq;
    }
  }
  AF_STAMP();                                                    // 8: irfft2 + x_orig done
  float mu2 = 0.f, a2 = 1.f, c2 = 0.f;
  if (p.g2) {
    float rs2;
    double mu_d;
    group_stats(s_m, s_mm, s_q, 1, mu2, mu_d, rs2);
    if (lane == 0 && wave == gw0) {
      p.mean2[b * p.G + grp] = mu2;
      p.rstd2[b * p.G + grp] = rs2;
    }
    a2 = rs2 * p.g2[ch];
    c2 = p.b2[ch];
  }
  const long long obase = (long long)b * (AF_H * AF_W) * E + kblk * AF_BS + 16 * wave;   // uniform
  int xlo3 = xlo2;
  asm volatile("" : "+v"(xlo3));
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
    for (int y = 0; y < AF_W; ++y) {
      const int o = xlo3 + (k1 * 4 * AF_W + y) * E;
      if (p.y1) p.y1[obase + o] = yv[k1][y];
      if (p.g2 && p.xn2) p.xn2[obase + o] = fmaf(yv[k1][y] - mu2, a2, c2);
    }
  if constexpr (PK) {
  if (p.g2 && (p.xp || p.xpT)) {
    // GroupNorm2(y1) straight into the bf16 operand packs of the channel MLP (the separate pack pass re-read y1: one launch
    // and one field-sized read per block less).  Same expression as bf16_pack_both_kernel: fmaf(v - mean, rstd * gamma, beta).
    const int Mtok = p.B * (AF_H * AF_W);
    unsigned short* const Aw = reinterpret_cast<unsigned short*>(A) + wave * (AF_H * AF_W * 16);   // [token][16 channels] bf16
    // (the operand region is free: the GroupNorm2 barrier above came after every wave's last fragment read)
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      const int xr = 4 * k1 + q;
      unsigned w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        w[j] = af_pack2(fmaf(yv[k1][2 * j] - mu2, a2, c2), fmaf(yv[k1][2 * j + 1] - mu2, a2, c2));
      if (p.xpT) {
        // transposed form: chunk = (feature, 8 consecutive tokens) - the lane's 16 values of a row are two chunks
        uint4* tp = p.xpT + ((long long)(ch >> 5) * (Mtok >> 4) + b * AF_H + xr) * 64 + (ch & 31);
        tp[0] = make_uint4(w[0], w[1], w[2], w[3]);
        tp[32] = make_uint4(w[4], w[5], w[6], w[7]);
      }
      if (p.xp) {
#pragma unroll
        for (int y = 0; y < AF_W; ++y)
          Aw[(xr * AF_W + y) * 16 + c] = (unsigned short)((y & 1) ? (w[y >> 1] >> 16) : (w[y >> 1] & 0xffffu));
      }
    }
    if (p.xp) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // row form: chunk = (token, 8 consecutive features); this wave's 16 channels are feature block kblk * 8 + wave
      const int fblk = kblk * (AF_BS / 16) + wave;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int id = i * 64 + lane, t = id >> 1, h = id & 1;
        const uint4 v = *reinterpret_cast<const uint4*>(Aw + t * 16 + h * 8);
        const int T = b * (AF_H * AF_W) + t;
        p.xp[((long long)(T >> 5) * (E >> 4) + fblk) * 64 + (T & 31) + 32 * h] = v;
      }
    }
  }
  }
#ifdef AF_TIMING
  AF_STAMP();                                                    // 9: stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  AF_STAMP();                                                    // 10: stores drained
  if ((wave == 0 || wave == 7) && lane == 0 && p.y1) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y1 + (long long)b * (AF_H * AF_W) * E + kblk * AF_BS) +
                            (wave ? 12 : 0);
    for (int i = 0; i < 11; ++i) o[i] = ts_[i];
  }
#endif
}

// (Round 5 also built the BACKWARD of the layer as one launch - GroupNorm2 backward -> weighted rfft2 -> both layers' data gradient ->
// inverse + GroupNorm1 backward: parity-green and 15 % SLOWER than the four launches it replaced (205 vs 179 us at 256
// workgroups, DPOT-S 5.27 -> 5.44 ms: a workgroup that owns its CU alone runs its load phases with nothing to hide them behind;
// profiles/r05_f4_bwd_fused_vs_launches.txt, r05_f4_bwd_step_ab.txt).  Removed in round 6; the source is in the history.)

template <int CG>
static int launch_fused(const AfnoFusedArgs& p, hipStream_t s) {
  const dim3 grid((unsigned)(p.B * p.nb)), blk(64 * AF_NW);
  const bool pk = p.xp || p.xpT;
  if (p.act == DPOT_ACT_GELU) {
    if (pk) hipLaunchKernelGGL((afno_fused_fwd_kernel<CG, DPOT_ACT_GELU, true>), grid, blk, 0, s, p);
    else hipLaunchKernelGGL((afno_fused_fwd_kernel<CG, DPOT_ACT_GELU, false>), grid, blk, 0, s, p);
  } else {
    if (pk) hipLaunchKernelGGL((afno_fused_fwd_kernel<CG, -1, true>), grid, blk, 0, s, p);
    else hipLaunchKernelGGL((afno_fused_fwd_kernel<CG, -1, false>), grid, blk, 0, s, p);
  }
  return check_launch("afno_fused_fwd_kernel");
}

}  // namespace dpot

using namespace dpot;

extern "C" int dpot_afno_fused_supported(int h, int w, int E, int G, int nb, int mx, int my) {
  if (h != AF_H || w != AF_W || mx != AF_H || my != AF_WF || nb <= 0 || E <= 0 || E % nb || E / nb != AF_BS) return 0;
  if (G <= 0) return 1;                                           // no GroupNorm: the mixer alone
  if (E % G) return 0;
  const int cg = E / G;
  return cg == 64 || cg == 128 ? 1 : 0;
}

extern "C" int dpot_afno_fused_fwd(const float* x, const float* gamma1, const float* beta1, const float* Wa,
                                   const float* ba, const float* Wb, const float* bb, const float* gamma2,
                                   const float* beta2, float* S, float* pre, float* y1, float* xn2, float* mean1,
                                   float* rstd1, float* mean2, float* rstd2, void* xn2_rows_bf16, void* xn2_trans_bf16,
                                   int B, int h, int w, int E, int G, int nb, int mx, int my, int act, float eps,
                                   dpot_stream_t stream) {
  const bool norm = gamma1 || gamma2;
  DPOT_REQUIRE(dpot_afno_fused_supported(h, w, E, norm ? G : 0, nb, mx, my),
               "afno_fused_fwd: needs a 16x16 latent grid, 128 channels per block, all modes kept, 64 or 128 channels per group");
  DPOT_REQUIRE(x && Wa && Wb && (y1 || xn2 || xn2_rows_bf16) && B > 0, "afno_fused_fwd: bad argument");
  DPOT_REQUIRE(!(xn2_rows_bf16 || xn2_trans_bf16) || (gamma2 && E % 32 == 0 && aligned16(xn2_rows_bf16) && aligned16(xn2_trans_bf16)),
               "afno_fused_fwd: the bf16 packs are packs of GroupNorm2's output (gamma2), E %% 32 == 0, 16-byte aligned");
  DPOT_REQUIRE(!gamma1 || (beta1 && mean1 && rstd1), "afno_fused_fwd: norm1 needs beta1, mean1, rstd1");
  DPOT_REQUIRE(!gamma2 || (beta2 && mean2 && rstd2), "afno_fused_fwd: norm2 needs beta2, mean2, rstd2");
  DPOT_REQUIRE(gamma2 || y1, "afno_fused_fwd: without norm2 the output is y1");
  DPOT_REQUIRE(aligned16(Wa) && aligned16(Wb), "afno_fused_fwd: weight packs must be 16-byte aligned");
  DPOT_REQUIRE((long long)B * nb <= 0x7fffffffLL && (long long)AF_H * AF_W * E < (1ll << 29), "afno_fused_fwd: too large");
  AfnoFusedArgs p;
  p.x = x; p.g1 = gamma1; p.b1 = beta1; p.Wa = Wa; p.ba = ba; p.Wb = Wb; p.bb = bb; p.g2 = gamma2; p.b2 = beta2;
  p.S = S; p.pre = pre; p.y1 = y1; p.xn2 = xn2; p.mean1 = mean1; p.rstd1 = rstd1; p.mean2 = mean2; p.rstd2 = rstd2;
  p.xp = reinterpret_cast<uint4*>(xn2_rows_bf16); p.xpT = reinterpret_cast<uint4*>(xn2_trans_bf16);
  p.B = B; p.E = E; p.G = G; p.nb = nb; p.act = act; p.eps = eps;
  hipStream_t s = as_stream(stream);
  const int cg = norm ? E / G : 128;
  return cg == 64 ? launch_fused<64>(p, s) : launch_fused<128>(p, s);
}
