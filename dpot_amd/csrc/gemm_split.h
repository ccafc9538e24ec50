// gemm_split.h - the fp32 GEMM of gemm.hip computed on the bf16 matrix cores by operand splitting ("bf16x6").
// Included by gemm.hip (shares GemmArgs / EpiArgs / epi_fragment / the split-K + column-sum protocol).
//
//   a = a1 + a2 + a3,  a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)        (8 + 8 + 8 significand bits:
//   the three-term sum reproduces the fp32 value to <= 2^-25 |a|)
//   a*b = a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1) + O(2^-24 |a b|)
// Every bf16 x bf16 product is exact in fp32 and the six partial products are accumulated in the fp32 MFMA
// accumulator, so the result carries the same ~2^-24 relative error per product as a native fp32 FMA chain - but
// v_mfma_f32_32x32x16_bf16 retires 16x the MACs per cycle of v_mfma_f32_32x32x2_f32: 6 bf16 MFMAs replace 8 fp32
// MFMAs per 32x32x16 block and take 6*32 instead of 8*64 cycles (2.67x the fp32 matrix-core roof).
//
// The split happens ONCE per staged element, between the global load and the LDS store (v_cvt_pk_bf16_f32 + shift +
// subtract: ~4.5 VALU ops per element); LDS holds three bf16 planes per operand, K-contiguous rows of 32 k + 8 pad
// (80 B = odd multiple of 16 B: the ds_read_b128 fragment reads - 8 consecutive k per lane, exactly the
// 32x32x16 A/B operand layout - are bank-conflict free).  Row-contiguous (transposed) operands are transposed in
// registers: a thread owns a (4 k x 4 rows) [or (2 k x 4 rows) for 64-row tiles] patch, loads it as float4 along the
// rows and writes k-pairs; the lane -> patch map makes those ds_write_b64 / b32 conflict free as well.
//
// Pipeline: two LDS buffers, ONE barrier per 32-k slab.  While the matrix cores work through the 12 term-groups
// (2 k-steps x 6 partial products) of slab t, the same wave splits the registers holding slab t+1 into the other
// buffer, piece by piece between the MFMAs, and re-issues the global loads for slab t+2 as soon as a register set
// has been consumed (>= half a slab of latency cover).  With one wave per SIMD (the 128x128 tile owns 120 KB of
// LDS) nothing else would overlap the split with the MFMAs, so the order is pinned with sched_barrier /
// sched_group_barrier instead of being left to the scheduler (which runs the whole split first).
#pragma once

namespace dpot {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int XLD = 40;  // bf16 elements per LDS row

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  f32x2 v = {lo, hi};
  bf16x2 r = __builtin_convertvector(v, bf16x2);   // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, r);
}

// (a, b) -> three packed bf16 pairs (a in the low half)
__device__ __forceinline__ void split3(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = pk_bf16(a, b);
  a -= __uint_as_float(p1 << 16);
  b -= __uint_as_float(p1 & 0xffff0000u);
  p2 = pk_bf16(a, b);
  a -= __uint_as_float(p2 << 16);
  b -= __uint_as_float(p2 & 0xffff0000u);
  p3 = pk_bf16(a, b);
}

// NPL planes: 3 = the full split above, 1 = plain bf16 rounding (the reduced-precision "bf16" GEMM mode)
template <int NPL>
__device__ __forceinline__ void splitn(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
  if constexpr (NPL == 3) {
    split3(a, b, p1, p2, p3);
  } else {
    p1 = pk_bf16(a, b);
    p2 = p3 = 0u;
  }
}

template <int I>
__device__ __forceinline__ float f4c(const float4& v) {
  if constexpr (I == 0) return v.x;
  else if constexpr (I == 1) return v.y;
  else if constexpr (I == 2) return v.z;
  else return v.w;
}

// ---- row-contiguous source (element (r, k) at base[k*ld + r]): per-thread patches --------------------------
//   16 elements / thread (R = 128, 256 threads): patch = 4 k x 4 rows, kq = tid & 7 (k = 4kq + j), rg = tid >> 3
//    8 elements / thread (R = 64 with 256 threads, R = 128 with 512): patch = 2 k x 4 rows, kp = tid & 15, rg = tid >> 4
//   rows 4rg .. 4rg+3, reg[j] = k-row j
template <int R, bool VEC, int NT>
__device__ __forceinline__ void load_rpatch(float4 (&reg)[R * 8 / NT], const float* __restrict__ base, int ld, int r0,
                                            int rmax, int k0, int kmax, int tid) {
  constexpr int NK = R * 8 / NT;             // k-rows per patch (4 or 2)
  constexpr int KG = 32 / NK;                // patches along k (8 or 16)
  const int kb = k0 + (tid % KG) * NK;
  const int r = r0 + (tid / KG) * 4;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const float* row = base + (long long)min(kb + j, kmax - 1) * ld;
    float4 v;
    if constexpr (VEC) {
      v = *reinterpret_cast<const float4*>(row + min(r, rmax - 4));
    } else {
      const int rl = rmax - 1;
      v.x = row[min(r + 0, rl)]; v.y = row[min(r + 1, rl)]; v.z = row[min(r + 2, rl)]; v.w = row[min(r + 3, rl)];
    }
    reg[j] = v;
  }
}

// zero the k >= kmax part of the registers of a partial slab (called under a workgroup-uniform branch)
template <int R, int NT>
__device__ __forceinline__ void mask_tail_rpatch(float4 (&reg)[R * 8 / NT], int k0, int kmax, int tid) {
  constexpr int NK = R * 8 / NT, KG = 32 / NK;
  const int kb = k0 + (tid % KG) * NK;
#pragma unroll
  for (int j = 0; j < NK; ++j)
    if (kb + j >= kmax) reg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
}
template <int R, int NT>
__device__ __forceinline__ void mask_tail_kcontig(float4 (&reg)[R * 8 / NT], int k0, int kmax, int tid) {
#pragma unroll
  for (int i = 0; i < R * 8 / NT; ++i) {
    const int k = k0 + (((tid + NT * i) & 7) << 2);
    if (k + 0 >= kmax) reg[i].x = 0.f;
    if (k + 1 >= kmax) reg[i].y = 0.f;
    if (k + 2 >= kmax) reg[i].z = 0.f;
    if (k + 3 >= kmax) reg[i].w = 0.f;
  }
}

// K-contiguous source: thread f of a 256-thread pass owns k-quad (f & 7) of tile row xrow(f).  Rows are visited in
// the order 0,4,1,5,2,6,3,7 (+8n): the 16 lanes of one ds_write_b64 group then cover rows r and r+4, whose 64-byte
// runs lie 320 B = 64 (mod 128) apart - all 32 banks exactly once.  (In natural order the runs of rows r, r+1
// overlap on 4 banks and every store takes twice the LDS cycles.)
__device__ __forceinline__ int xrow(int f) {
  const int q = f >> 3;
  return (q & ~7) | ((q & 1) << 2) | ((q >> 1) & 3);
}
template <int R, bool VEC, int NT>
__device__ __forceinline__ void load_kcontig_x(float4 (&reg)[R * 8 / NT], const float* __restrict__ base, int ld,
                                               int r0, int rmax, int k0, int kmax, int tid) {
#pragma unroll
  for (int i = 0; i < R * 8 / NT; ++i) {
    const int f = tid + NT * i;
    const int r = r0 + xrow(f);
    const int k = k0 + ((f & 7) << 2);
    const float* row = base + (long long)min(r, rmax - 1) * ld;
    float4 v;
    if constexpr (VEC) {
      v = *reinterpret_cast<const float4*>(row + min(k, kmax - 4));
    } else {
      const int kl = kmax - 1;
      v.x = row[min(k + 0, kl)]; v.y = row[min(k + 1, kl)]; v.z = row[min(k + 2, kl)]; v.w = row[min(k + 3, kl)];
    }
    reg[i] = v;
  }
}

// ---- split + store, in R/32 equal pieces (4 elements -> 18 VALU + the stores) so that the caller can spread them
//      between the MFMAs.  lds = plane 0 of the operand; planes are R*XLD elements apart.
// K-contiguous source (same thread -> (row, k-quad) map as load_kcontig): piece P = reg[P]
template <int R, int P, int NPL, int NT>
__device__ __forceinline__ void split_piece_kcontig(__bf16* lds, const float4 (&reg)[R * 8 / NT], int tid) {
  const int f = tid + NT * P;
  unsigned a1, a2, a3, b1, b2, b3;
  splitn<NPL>(reg[P].x, reg[P].y, a1, a2, a3);
  splitn<NPL>(reg[P].z, reg[P].w, b1, b2, b3);
  __bf16* dst = lds + xrow(f) * XLD + ((f & 7) << 2);
  *reinterpret_cast<uint2*>(dst) = make_uint2(a1, b1);
  if constexpr (NPL == 3) {
    *reinterpret_cast<uint2*>(dst + R * XLD) = make_uint2(a2, b2);
    *reinterpret_cast<uint2*>(dst + 2 * R * XLD) = make_uint2(a3, b3);
  }
}
// row-contiguous patch, transposed in registers: R = 128: piece P = patch row P (4 k);  R = 64: rows 2P, 2P+1 (2 k)
template <int R, int P, int NPL, int NT>
__device__ __forceinline__ void split_piece_rpatch(__bf16* lds, const float4 (&reg)[R * 8 / NT], int tid) {
  constexpr int NK = R * 8 / NT, KG = 32 / NK;
  const int kofs = (tid % KG) * NK;
  const int row = (tid / KG) * 4;
  if constexpr (NK == 4) {
    __bf16* dst = lds + (row + P) * XLD + kofs;
    unsigned a1, a2, a3, b1, b2, b3;
    splitn<NPL>(f4c<P>(reg[0]), f4c<P>(reg[1]), a1, a2, a3);
    splitn<NPL>(f4c<P>(reg[2]), f4c<P>(reg[3]), b1, b2, b3);
    *reinterpret_cast<uint2*>(dst) = make_uint2(a1, b1);
    if constexpr (NPL == 3) {
      *reinterpret_cast<uint2*>(dst + R * XLD) = make_uint2(a2, b2);
      *reinterpret_cast<uint2*>(dst + 2 * R * XLD) = make_uint2(a3, b3);
    }
  } else {
    __bf16* dst = lds + (row + 2 * P) * XLD + kofs;
    unsigned a1, a2, a3, b1, b2, b3;
    splitn<NPL>(f4c<2 * P>(reg[0]), f4c<2 * P>(reg[1]), a1, a2, a3);
    splitn<NPL>(f4c<2 * P + 1>(reg[0]), f4c<2 * P + 1>(reg[1]), b1, b2, b3);
    *reinterpret_cast<unsigned*>(dst) = a1;
    *reinterpret_cast<unsigned*>(dst + XLD) = b1;
    if constexpr (NPL == 3) {
      *reinterpret_cast<unsigned*>(dst + R * XLD) = a2;
      *reinterpret_cast<unsigned*>(dst + 2 * R * XLD) = a3;
      *reinterpret_cast<unsigned*>(dst + XLD + R * XLD) = b2;
      *reinterpret_cast<unsigned*>(dst + XLD + 2 * R * XLD) = b3;
    }
  }
}

// sum over the patch's k of its 4 rows (fused bias gradient)
template <int NR>
__device__ __forceinline__ void patch_rowsum(float4& cs, const float4 (&reg)[NR]) {
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    cs.x += reg[j].x; cs.y += reg[j].y; cs.z += reg[j].z; cs.w += reg[j].w;
  }
}

// NW waves per workgroup: 4 (2x2 wave grid) or 8 (2x4: two waves per SIMD on the 128x128 tile, so that one wave's
// operand split overlaps the other's MFMAs)
template <int BM, int BN, bool TA, bool TB, bool VEC, int TAG, int NPL, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_f32x_kernel(const GemmArgs p) {
  constexpr int NT = 64 * NW, WCOLS = NW / 2;
  static_assert(BM == BN, "square tiles only");
  static_assert(NPL == 3 || NPL == 1, "3 planes = bf16x6 (fp32-accurate), 1 plane = plain bf16");
  constexpr bool A_KC = !TA;
  constexpr bool B_KC = TB;
  constexpr int WM = BM / 2, WN = BN / WCOLS;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NP = BM * 8 / NT;                                          // split pieces / staging float4 per operand
  constexpr int A_ELEMS = NPL * BM * XLD, B_ELEMS = NPL * BN * XLD;      // bf16 elements per buffer
  constexpr int BUF_BYTES = 2 * (A_ELEMS + B_ELEMS);
  constexpr int STAGE_BYTES = NW * 32 * EPI_LD * 4;
  // two DISTINCT arrays (not one array indexed by kt & 1): the compiler must know that the split's stores into one
  // buffer cannot alias the fragment reads of the other, or it keeps all stores ahead of all reads
  __shared__ __attribute__((aligned(16))) unsigned char buf0[BUF_BYTES > STAGE_BYTES ? BUF_BYTES : STAGE_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char buf1[BUF_BYTES];
  float* smem = reinterpret_cast<float*>(buf0);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WCOLS, wn = wave % WCOLS;
  const int li = lane & 31, kh = lane >> 5;

  const int ntiles = p.tilesM * p.tilesN;
  int tile;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tm_idx = tile / p.tilesN, tn_idx = tile % p.tilesN;
  const int m0 = tm_idx * BM, n0 = tn_idx * BN;
  const int zb = blockIdx.z / p.splits, zs = blockIdx.z % p.splits;
  const bool cs_on = (TA && p.cs_of == 1 && tn_idx == 0) || (!TB && p.cs_of == 2 && tm_idx == 0);
  float4 csA = make_float4(0.f, 0.f, 0.f, 0.f), csB = csA;   // separate accumulators: no pointer selects -> no scratch

  const float* A = p.A + zb * p.sA;
  const float* B = p.B + zb * p.sB;
  const int kbeg = zs * p.ktiles_per_split * BK;
  int kend = kbeg + p.ktiles_per_split * BK;
  if (kend > p.K) kend = p.K;
  const int nk = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[NP], rb[NP];
  auto gloadA = [&](int k0) __attribute__((always_inline)) {
    if constexpr (A_KC) load_kcontig_x<BM, VEC, NT>(ra, A, p.lda, m0, p.M, k0, kend, tid);
    else load_rpatch<BM, VEC, NT>(ra, A, p.lda, m0, p.M, k0, kend, tid);
  };
  auto gloadB = [&](int k0) __attribute__((always_inline)) {
    if constexpr (B_KC) load_kcontig_x<BN, VEC, NT>(rb, B, p.ldb, n0, p.N, k0, kend, tid);
    else load_rpatch<BN, VEC, NT>(rb, B, p.ldb, n0, p.N, k0, kend, tid);
  };
  // uniform-branch work on a freshly loaded slab: zero the k >= kend part of a partial slab, bias-gradient sums
  auto fixup = [&](int k0) __attribute__((always_inline)) {
    if (k0 + BK > kend) {
      if constexpr (A_KC) mask_tail_kcontig<BM, NT>(ra, k0, kend, tid); else mask_tail_rpatch<BM, NT>(ra, k0, kend, tid);
      if constexpr (B_KC) mask_tail_kcontig<BN, NT>(rb, k0, kend, tid); else mask_tail_rpatch<BN, NT>(rb, k0, kend, tid);
    }
    if (cs_on) {
      // both candidates are summed and the wanted one is picked BY VALUE at the end: a run-time choice here makes
      // hipcc merge the two calls into one over selected pointers, which sends ra / rb / cs to scratch
      if constexpr (TA) patch_rowsum<NP>(csA, ra);
      if constexpr (!TB) patch_rowsum<NP>(csB, rb);
    }
  };

#define DPOT_SPLIT_A(P, BUF)                                                                                 \
  do {                                                                                                       \
    if constexpr ((P) < NP) {                                                                                \
      if constexpr (A_KC) split_piece_kcontig<BM, (P) < NP ? (P) : 0, NPL, NT>(reinterpret_cast<__bf16*>(BUF), ra, tid);         \
      else split_piece_rpatch<BM, (P) < NP ? (P) : 0, NPL, NT>(reinterpret_cast<__bf16*>(BUF), ra, tid);                         \
    }                                                                                                        \
  } while (0)
#ifdef DPOT_X_NOSPLITB   /* experiment: upper bound of what pre-split B planes would give */
#define DPOT_X_DOB false
#else
#define DPOT_X_DOB true
#endif
#define DPOT_SPLIT_B(P, BUF)                                                                                 \
  do {                                                                                                       \
    if constexpr ((P) < NP && DPOT_X_DOB) {                                                                                \
      if constexpr (B_KC) split_piece_kcontig<BN, (P) < NP ? (P) : 0, NPL, NT>(reinterpret_cast<__bf16*>(BUF) + A_ELEMS, rb, tid); \
      else split_piece_rpatch<BN, (P) < NP ? (P) : 0, NPL, NT>(reinterpret_cast<__bf16*>(BUF) + A_ELEMS, rb, tid);               \
    }                                                                                                        \
  } while (0)
  // fragment reads: split plane S of the wave's A rows / B columns, k-step KK (0 or 16)
#define DPOT_RD_A(DST, CUR, S, KK)                                                                           \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) DST[i] = *reinterpret_cast<const bf16x8*>(                  \
      reinterpret_cast<const __bf16*>(CUR) + ((S) * BM + wm * WM + i * 32 + li) * XLD + (KK) + 8 * kh)
#define DPOT_RD_B(DST, CUR, S, KK)                                                                           \
  _Pragma("unroll") for (int j = 0; j < TN; ++j) DST[j] = *reinterpret_cast<const bf16x8*>(                  \
      reinterpret_cast<const __bf16*>(CUR) + A_ELEMS + ((S) * BN + wn * WN + j * 32 + li) * XLD + (KK) + 8 * kh)
  // one partial product over the wave tile: TM*TN MFMAs on different accumulators
#define DPOT_TERM(FA, FB)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i], FB[j], acc[i][j], 0, 0, 0)
  // end of a unit: inside it alternate 1 MFMA : VALU : LDS/global, then pin the unit against its neighbours
#define DPOT_UNIT_END()                                                                                      \
  do {                                                                                                       \
    _Pragma("unroll") for (int g = 0; g < TM * TN; ++g) {                                                    \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
      __builtin_amdgcn_sched_group_barrier(0x002, 24 / (TM * TN), 0);                                        \
      __builtin_amdgcn_sched_group_barrier(0x090, 8 / (TM * TN), 0);                                         \
    }                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)

  // one 32-k slab: 12 units = (k-step 0: terms a3b1 a1b3 a2b2 a2b1 a1b2 a1b1) (k-step 1: the same), smallest first.
  // Units 0-7 also split slab kt+1 into `nxt`; every LDS access of the slab (fragment reads of `cur`: units <= 5,
  // split stores: units <= 7) is over by the barrier after unit 7, so units 8-11 can already fetch the first
  // k-step's fragments of the NEXT slab from `nxt` - no fragment-read latency is exposed behind the barrier.
  bf16x8 a0[TM], a1[TM], a2[TM], b0[TN], b1[TN], b2[TN];     // k-step 0 fragments (split planes 0..2), carried
  auto step = [&](int kt, const unsigned char* cur, unsigned char* nxt) __attribute__((always_inline)) {
    if (kt + 1 < nk) fixup(kbeg + (kt + 1) * BK);
    bf16x8 c0[TM], c1[TM], c2[TM], d0[TN], d1[TN], d2[TN];     // k-step 1
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NPL == 3) {
      DPOT_SPLIT_A(0, nxt); DPOT_TERM(a2, b0); DPOT_UNIT_END();
      DPOT_SPLIT_A(1, nxt); DPOT_TERM(a0, b2); DPOT_UNIT_END();
      DPOT_SPLIT_A(2, nxt); DPOT_RD_A(c2, cur, 2, 16); DPOT_TERM(a1, b1); DPOT_UNIT_END();
      DPOT_SPLIT_A(3, nxt); DPOT_RD_B(d0, cur, 0, 16); DPOT_TERM(a1, b0); DPOT_UNIT_END();
      gloadA(kbeg + (kt + 2) * BK);
      DPOT_SPLIT_B(0, nxt); DPOT_RD_A(c0, cur, 0, 16); DPOT_RD_B(d2, cur, 2, 16); DPOT_TERM(a0, b1); DPOT_UNIT_END();
      DPOT_SPLIT_B(1, nxt); DPOT_RD_A(c1, cur, 1, 16); DPOT_RD_B(d1, cur, 1, 16); DPOT_TERM(a0, b0); DPOT_UNIT_END();
      DPOT_SPLIT_B(2, nxt); DPOT_TERM(c2, d0); DPOT_UNIT_END();
      DPOT_SPLIT_B(3, nxt); DPOT_TERM(c0, d2); DPOT_UNIT_END();
      gloadB(kbeg + (kt + 2) * BK);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      DPOT_RD_A(a2, nxt, 2, 0); DPOT_RD_B(b0, nxt, 0, 0); DPOT_TERM(c1, d1); DPOT_UNIT_END();
      DPOT_RD_A(a0, nxt, 0, 0); DPOT_RD_B(b2, nxt, 2, 0); DPOT_TERM(c1, d0); DPOT_UNIT_END();
      DPOT_RD_A(a1, nxt, 1, 0); DPOT_RD_B(b1, nxt, 1, 0); DPOT_TERM(c0, d1); DPOT_UNIT_END();
      DPOT_TERM(c0, d0); DPOT_UNIT_END();
    } else {
      // plain bf16: one product per k-step; the (cheap) rounding of slab kt+1 rides on the first k-step
      DPOT_RD_A(c0, cur, 0, 16); DPOT_RD_B(d0, cur, 0, 16);
      DPOT_SPLIT_A(0, nxt); DPOT_SPLIT_A(1, nxt); DPOT_SPLIT_A(2, nxt); DPOT_SPLIT_A(3, nxt);
      gloadA(kbeg + (kt + 2) * BK);
      DPOT_TERM(a0, b0);
      DPOT_SPLIT_B(0, nxt); DPOT_SPLIT_B(1, nxt); DPOT_SPLIT_B(2, nxt); DPOT_SPLIT_B(3, nxt);
      gloadB(kbeg + (kt + 2) * BK);
      __syncthreads();
      DPOT_RD_A(a0, nxt, 0, 0); DPOT_RD_B(b0, nxt, 0, 0);
      DPOT_TERM(c0, d0);
    }
    // keep the accumulators in AGPRs across the loop edge (otherwise the allocator parks them in VGPRs and copies
    // all 64 of them into AGPRs and back around every slab)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" : "+a"(acc[i][j]));
  };

  if (nk > 0) {
    gloadA(kbeg);
    gloadB(kbeg);
    fixup(kbeg);
    DPOT_SPLIT_A(0, buf0); DPOT_SPLIT_A(1, buf0); DPOT_SPLIT_A(2, buf0); DPOT_SPLIT_A(3, buf0);
    DPOT_SPLIT_B(0, buf0); DPOT_SPLIT_B(1, buf0); DPOT_SPLIT_B(2, buf0); DPOT_SPLIT_B(3, buf0);
    gloadA(kbeg + BK);
    gloadB(kbeg + BK);
    __syncthreads();
    DPOT_RD_A(a0, buf0, 0, 0); DPOT_RD_B(b0, buf0, 0, 0);
    if constexpr (NPL == 3) {
      DPOT_RD_A(a2, buf0, 2, 0); DPOT_RD_B(b2, buf0, 2, 0); DPOT_RD_A(a1, buf0, 1, 0); DPOT_RD_B(b1, buf0, 1, 0);
    }
    // no conditional step inside the loop: a phi of the accumulators there is legalised through VGPRs (64 AGPR->VGPR
    // and 64 VGPR->AGPR copies per slab)
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
      step(kt, buf0, buf1);
      step(kt + 1, buf1, buf0);
    }
    if (kt < nk) step(kt, buf0, buf1);
    __syncthreads();   // the last step's prefetch reads and MFMAs of the other waves are done before buf0 is re-used
  }
#undef DPOT_SPLIT_A
#undef DPOT_SPLIT_B
#undef DPOT_RD_A
#undef DPOT_RD_B
#undef DPOT_TERM
#undef DPOT_UNIT_END

  // ---- epilogue (operand planes are dead: reuse the LDS as per-wave staging; the loop ended on a barrier)
  if (cs_on) {
    // lanes sharing a row group differ in the low KG bits of tid: butterfly in a fixed order
    constexpr int KG = 32 / NP;
    const bool ofA = p.cs_of == 1;
    float cs[4] = {ofA ? csA.x : csB.x, ofA ? csA.y : csB.y, ofA ? csA.z : csB.z, ofA ? csA.w : csB.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = cs[i];
      v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
      if (KG == 16) v += __shfl_xor(v, 8);
      cs[i] = v;
    }
    if (tid % KG == 0) {
      const int L = p.cs_of == 1 ? p.M : p.N;
      const int g0 = (p.cs_of == 1 ? m0 : n0) + (tid / KG) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int g = g0 + i;
        if (g < L) {
          if (p.splits > 1)
            p.ws[(long long)p.splits * p.batch * p.M * p.N + ((long long)zs * p.batch + zb) * L + g] = cs[i];
          else
            p.cs_out[zb * p.sCs + g] = cs[i];
        }
      }
    }
  }
  float* stage = smem + wave * (32 * EPI_LD);
#define DPOT_EPI_FRAG(I, J)                                                                    \
  do {                                                                                         \
    const int n0f = n0 + wn * WN + (J) * 32;                                                   \
    const int m0f = m0 + wm * WM + (I) * 32;                                                   \
    if (p.splits > 1) {                                                                        \
      float* ws = p.ws + ((long long)zs * p.batch + zb) * p.M * p.N;                           \
      const int n = n0f + li;                                                                  \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
        const int m = m0f + 4 * kh + (r & 3) + 8 * (r >> 2);                                   \
        if (m < p.M && n < p.N) ws[(long long)m * p.N + n] = acc[I][J][r];                     \
      }                                                                                        \
    } else {                                                                                   \
      epi_fragment(p.e, p.evec, zb, m0f, n0f, acc[I][J], stage, lane);                         \
    }                                                                                          \
  } while (0)
  DPOT_EPI_FRAG(0, 0);
  if constexpr (TN > 1) DPOT_EPI_FRAG(0, 1);
  if constexpr (TM > 1) {
    DPOT_EPI_FRAG(1, 0);
    if constexpr (TN > 1) DPOT_EPI_FRAG(1, 1);
  }
#undef DPOT_EPI_FRAG
}

template <int BMN, bool VEC, int NPL, int NW>
static void launch_gemm_split(const dpot_gemm_desc* d, const GemmArgs& p, dim3 grid, hipStream_t s) {
  const int key = (d->transA ? 2 : 0) | (d->transB ? 1 : 0);
  const dim3 blk(64 * NW);
  if (d->tag == 1 && key == 0) {
    hipLaunchKernelGGL((gemm_f32x_kernel<BMN, BMN, false, false, VEC, 1, NPL, NW>), grid, blk, 0, s, p);
    return;
  }
  switch (key) {
    case 0: hipLaunchKernelGGL((gemm_f32x_kernel<BMN, BMN, false, false, VEC, 0, NPL, NW>), grid, blk, 0, s, p); break;
    case 1: hipLaunchKernelGGL((gemm_f32x_kernel<BMN, BMN, false, true, VEC, 0, NPL, NW>), grid, blk, 0, s, p); break;
    case 2: hipLaunchKernelGGL((gemm_f32x_kernel<BMN, BMN, true, false, VEC, 0, NPL, NW>), grid, blk, 0, s, p); break;
    default: hipLaunchKernelGGL((gemm_f32x_kernel<BMN, BMN, true, true, VEC, 0, NPL, NW>), grid, blk, 0, s, p); break;
  }
}

}  // namespace dpot
