// gemm_tn.hip - fp32 weight-gradient GEMM  C[N1, N2] = sum_t A[t, n1] * B[t, n2]   (A = dY, B = X: both stored
// token-major, the contraction runs over the ROWS of both operands; dpot_gemm_f32 with transA = 1, transB = 0).
//
// Small outputs (512 x 512 at DPOT-Tiny), 8192-long contractions, 256 CUs: the work has to be split along the
// tokens, and the kernel must make the most of each pass over its token range.  Structure:
//   * workgroup = one 128 x 128 output tile x one token range (split-K, partial sums to the workspace; the existing
//     fixed-order reduce kernels of gemm.hip finish: bias-gradient column sums, AFNO un-packing, accumulate ...);
//   * 4 compute waves in a 2 x 2 grid, each 64 x 64 = 4 x 4 accumulators of v_mfma_f32_16x16x4_f32, + 2 loader waves
//     that only issue LDS-DMA: a token row of a tile is 512 contiguous bytes, two rows per wave-instruction, the slab of
//     32 tokens lands in LDS exactly as it lies in memory ([tok][128] for each operand);
//   * ROW-INTERLEAVED fragments: the MFMA A operand wants, per lane, one element (row n, token k).  A ds_read_b128 at
//     [tok = 4q + (lane >> 4)][4 * (lane & 15) .. +3] hands every lane FOUR of them - for the four 16-row tiles made of
//     the rows n = 4 i + e (e = 0..3).  Which 16 rows form a tile is a free choice, so 64 rows of an operand cost one
//     LDS instruction per 4-token step: 2 ds_read_b128 per 16 MFMAs.  The interleave also makes the accumulator layout
//     store-friendly: a lane holds 4 consecutive columns of one output row -> 16-byte global stores, no LDS staging;
//   * ring of 4 slabs of 32 tokens (128 KiB), loaders 3 slabs ahead with counted vmcnt, ONE barrier per slab (8
//     MFMA k-steps = 128 MFMAs per wave between barriers), the first fragments of slab t+1 fetched before the barrier
//     (the loaders guarantee slab t+1 at barrier t).  Variants measured at 512 x 512 x 8192 (split 16, incl. the reduce
//     launch; generic kernel of gemm.hip: 57.4 us): 16-token slabs 54.1 us; 8 compute waves (two token groups added
//     through LDS at the end) 55.8; 8-slab ring / 6 ahead 57.2 - neither wave-level parallelism nor DMA depth is the
//     limiter, the per-slab barrier + first-fragment latency is;
//   * bias gradients (column sums of one operand over the tokens) come from the fragments the waves read anyway.
#include "common.h"
#include "gemm_epi.h"
#include <string.h>

namespace dpot {

typedef float tn_f32x4 __attribute__((ext_vector_type(4)));

struct TnArgs {
  const float* A;        // [T, lda]  (columns = output rows n1)
  const float* B;        // [T, ldb]  (columns = output columns n2)
  const float* A2;       // problems zb >= batch1 take their operands from (A2, B2) (same strides): the two weight
  const float* B2;       //   gradients of an AFNO block in one launch (dpot_afno_wgrad2)
  int batch1;
  int cs_of2;            // column-sum operand of the second problem set (cs_of serves the first)
  int csL;               // length of a column-sum partial slot (0: N1 or N2 by cs_of)
  int lda, ldb;
  long long sA, sB;      // batch strides (elements)
  int N1, N2, T, batch;
  int tiles1, tiles2, splits, slabs_per_split;
  float* ws;             // [split][batch][N1][N2] (+ [split][batch][L] column-sum partials behind)
  int cs_of;             // 0: none, 1: column sums of A (length N1), 2: of B (length N2)
  // round 5: the AFNO weight gradient dW = S^H dO as THREE real 128 x 128 products instead of the four blocks of the
  // 256 x 256 real product (bs == 128): with A = [Ar | Ai], B = [Br | Bi] (columns)
  //   P1 = Ar^T Br, P2 = Ai^T Bi, P3 = (Ar + Ai)^T (Bi - Br);   dWr = P1 + P2,  dWi = Ar^T Bi - Ai^T Br = P3 + P1 - P2
  // (the reduce does the recombination).  grid.x = 3: tile 0 / 1 are the ordinary tiles (0,0) / (1,1), tile 2 streams all
  // 256 columns of both operands in 16-token slabs (the same 32 KiB per slab) and forms the two sums on the fragments;
  // partials [split][batch][3][128][128]
  int gauss;
  // three-product form on gemm_tn_kernel (128 channels per block): the sum-product tile is the slowest of the three (four fragment
  // reads and the VALU sums per step, twice the barriers), so it gets SHORTER token ranges: it is cut into `splits` ranges, the
  // tiles P1 / P2 into splits12 <= splits (the partial slots zs >= splits12 of those tiles are neither written nor read).
  // splits12 = splits elsewhere
  int splits12, slabs_per_split12;
};

constexpr int TN_TOK = 32;                 // tokens per slab
constexpr int TN_W = 128;                  // tile width (both operands)
constexpr int TN_SLABF = 2 * TN_TOK * TN_W;  // floats per slab: A [16][128] | B [16][128]
constexpr int TN_RING = 4;
constexpr int TN_AHEAD = 3;                // slabs the loaders run ahead
constexpr int TN_NI = 16;                  // DMA instructions per slab and loader wave (1 KiB each)

template <int N>
__device__ __forceinline__ void tn_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void tn_glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__global__ __launch_bounds__(384) void gemm_tn_kernel(const TnArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[TN_RING * TN_SLABF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int ntiles = p.tiles1 * p.tiles2;
  int tile, zb, zs;
  if (p.gauss) {
    // three-product form: a ONE-dimensional grid of exactly the workgroups that have work - per problem splits12 token ranges of
    // P1, splits12 of P2, `splits` of the sum-product tile (the hardware deals consecutive workgroups to the 8 XCDs in turn: a grid
    // with idle members leaves some XCD with more than its 32 workgroups and the launch with a second round)
    const int W = 2 * p.splits12 + p.splits;
    const int lin = blockIdx.x;
    zb = lin / W;
    const int r = lin - zb * W;
    tile = r < p.splits12 ? 0 : r < 2 * p.splits12 ? 1 : 2;
    zs = r - (tile == 2 ? 2 * p.splits12 : tile * p.splits12);
  } else {
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    zb = blockIdx.z / p.splits;
    zs = blockIdx.z - zb * p.splits;
  }
  int t1 = tile / p.tiles2, t2 = tile - t1 * p.tiles2;
  const bool g3 = p.gauss && tile == 2;                 // the sum-product tile of the three-product form
  if (p.gauss) t1 = t2 = (tile == 1 ? 1 : 0);
  const int nslab_all = p.T / TN_TOK;
  const int sps = p.gauss && !g3 ? p.slabs_per_split12 : p.slabs_per_split;
  int slab0 = zs * sps;
  int nslab = nslab_all - slab0;
  nslab = nslab < sps ? nslab : sps;
  if (nslab < 0) nslab = 0;
  if (g3) {                                             // 16-token slabs: twice as many over the same token range
    slab0 *= 2;
    nslab *= 2;
  }

  auto bar = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (wave >= 4) {
    // ---------------------------------- loader waves ----------------------------------
    // wave L issues the tokens 16L .. 16L+15 of both operands (instruction j = tokens 16L + 2j, + 2j + 1 of one operand);
    // lane -> (token parity lane >> 5, 16 B piece lane & 31)
    const int L = wave - 4;
    const bool second = zb >= p.batch1;
    const int zq = second ? zb - p.batch1 : zb;
    // ordinary tile: an instruction moves two token rows of 128 columns; sum-product tile: ONE token row of all 256 columns
    // (wave L: tokens 8L .. 8L+7 of a 16-token slab) - 8 + 8 instructions per slab and wave either way
    const float* a0 = g3 ? (second ? p.A2 : p.A) + zq * p.sA + (long long)(slab0 * 16 + 8 * L) * p.lda + lane * 4
                         : (second ? p.A2 : p.A) + zq * p.sA + (long long)(slab0 * TN_TOK + 16 * L + (lane >> 5)) * p.lda +
                               t1 * TN_W + (lane & 31) * 4;
    const float* b0 = g3 ? (second ? p.B2 : p.B) + zq * p.sB + (long long)(slab0 * 16 + 8 * L) * p.ldb + lane * 4
                         : (second ? p.B2 : p.B) + zq * p.sB + (long long)(slab0 * TN_TOK + 16 * L + (lane >> 5)) * p.ldb +
                               t2 * TN_W + (lane & 31) * 4;
    const long long sa2 = (g3 ? 1ll : 2ll) * p.lda, sb2 = (g3 ? 1ll : 2ll) * p.ldb;
    const long long saS = (long long)(g3 ? 16 : TN_TOK) * p.lda, sbS = (long long)(g3 ? 16 : TN_TOK) * p.ldb;
    auto issue = [&](int t, int ring) __attribute__((always_inline)) {
      float* dst = lds + ring * TN_SLABF + L * 8 * 256;
      const float* a = a0 + t * saS;
      const float* b = b0 + t * sbS;
#pragma unroll
      for (int j = 0; j < 8; ++j) tn_glds16(a + j * sa2, dst + j * 256);
#pragma unroll
      for (int j = 0; j < 8; ++j) tn_glds16(b + j * sb2, dst + TN_TOK * TN_W + j * 256);
    };
    if (nslab > 0) issue(0, 0);
    if (nslab > 1) issue(1, 1);
    if (nslab > 2) issue(2, 2);
    if (nslab > 2) tn_wait_vm<2 * TN_NI>(); else if (nslab > 1) tn_wait_vm<TN_NI>(); else tn_wait_vm<0>();
    bar();                                              // P: slab 0 landed
    int ring = TN_AHEAD;                                // slot of slab g + 3
#pragma unroll 1
    for (int g = 0; g < nslab; ++g) {
      // slab g + 1 must have landed before B_g (the compute waves prefetch its first fragments before B_(g+1))
      if (g + 2 < nslab) tn_wait_vm<TN_NI>(); else tn_wait_vm<0>();
      bar();                                            // B_g: everyone is done with slab g - 1 -> its slot is free
      if (g + TN_AHEAD < nslab) issue(g + TN_AHEAD, ring);
      ring = ring == TN_RING - 1 ? 0 : ring + 1;
    }
    return;
  }

  // ---------------------------------- compute waves ----------------------------------
  const int wm = wave >> 1, wn = wave & 1;
  const int i16 = lane & 15, kq = lane >> 4;
  tn_f32x4 acc[4][4];
#pragma unroll
  for (int ea = 0; ea < 4; ++ea)
#pragma unroll
    for (int eb = 0; eb < 4; ++eb) acc[ea][eb] = tn_f32x4{0.f, 0.f, 0.f, 0.f};
  tn_f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  const int cs_sel = zb >= p.batch1 ? p.cs_of2 : p.cs_of;
  const bool cs_a = cs_sel == 1 && (p.gauss ? !g3 : t2 == 0) && wn == 0;
  const bool cs_b = cs_sel == 2 && (p.gauss ? !g3 : t1 == 0) && wm == 0;

  // this lane's fragment addresses inside a slab: A [tok = 4q + kq][wm*64 + 4*i16], B likewise with wn
  const int offA = kq * TN_W + wm * 64 + 4 * i16;
  const int offB = TN_TOK * TN_W + kq * TN_W + wn * 64 + 4 * i16;
  constexpr int NQ = TN_TOK / 4;                        // 4-token MFMA steps per slab

  bar();                                                // P
  if (g3) {
    // sum-product tile: slab image A [16 tok][256] | B [16 tok][256]; per 4-token step four fragment reads (Ar, Ai, Br, Bi),
    // a = Ar + Ai, b = Bi - Br on the VALU, 16 MFMAs
    const int oA = kq * 256 + wm * 64 + 4 * i16, oB = TN_TOK * TN_W + kq * 256 + wn * 64 + 4 * i16;
    tn_f32x4 far = {0.f, 0.f, 0.f, 0.f}, fai = far, fbr = far, fbi = far;
    if (nslab > 0) {
      far = *reinterpret_cast<const tn_f32x4*>(lds + oA);
      fai = *reinterpret_cast<const tn_f32x4*>(lds + oA + 128);
      fbr = *reinterpret_cast<const tn_f32x4*>(lds + oB);
      fbi = *reinterpret_cast<const tn_f32x4*>(lds + oB + 128);
    }
    int ring = 0;
#pragma unroll 1
    for (int g = 0; g < nslab; ++g) {
      bar();                                            // B_g
      const float* cur = lds + ring * TN_SLABF;
      const int rn = ring == TN_RING - 1 ? 0 : ring + 1;
      const float* nxt = g + 1 < nslab ? lds + rn * TN_SLABF : cur;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const tn_f32x4 a = far + fai, b = fbi - fbr;
        const float* src = q + 1 < 4 ? cur + (q + 1) * 4 * 256 : nxt;
        far = *reinterpret_cast<const tn_f32x4*>(src + oA);
        fai = *reinterpret_cast<const tn_f32x4*>(src + oA + 128);
        fbr = *reinterpret_cast<const tn_f32x4*>(src + oB);
        fbi = *reinterpret_cast<const tn_f32x4*>(src + oB + 128);
#pragma unroll
        for (int ea = 0; ea < 4; ++ea)
#pragma unroll
          for (int eb = 0; eb < 4; ++eb)
            acc[ea][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ea], b[eb], acc[ea][eb], 0, 0, 0);
      }
      ring = rn;
    }
  } else {
  tn_f32x4 fa = {0.f, 0.f, 0.f, 0.f}, fb = fa;
  if (nslab > 0) {
    fa = *reinterpret_cast<const tn_f32x4*>(lds + offA);
    fb = *reinterpret_cast<const tn_f32x4*>(lds + offB);
  }
  int ring = 0;
#pragma unroll 1
  for (int g = 0; g < nslab; ++g) {
    bar();                                              // B_g
    const float* cur = lds + ring * TN_SLABF;
    const int rn = ring == TN_RING - 1 ? 0 : ring + 1;
    // the step after the last one of this slab is the first one of the next slab (landed before B_g); on the very last
    // slab the address is clamped to this slab (the value is never used)
    const float* nxt = g + 1 < nslab ? lds + rn * TN_SLABF : cur;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const tn_f32x4 a = fa, b = fb;
      const float* src = q + 1 < NQ ? cur + (q + 1) * 4 * TN_W : nxt;
      fa = *reinterpret_cast<const tn_f32x4*>(src + offA);
      fb = *reinterpret_cast<const tn_f32x4*>(src + offB);
      if (cs_a) cs += a;
      if (cs_b) cs += b;
#pragma unroll
      for (int ea = 0; ea < 4; ++ea)
#pragma unroll
        for (int eb = 0; eb < 4; ++eb)
          acc[ea][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ea], b[eb], acc[ea][eb], 0, 0, 0);
    }
    ring = rn;
  }
  }

  // partial tile: element (n1 = 4*(4*kq + r) + ea, n2 = 4*i16 + eb) of the wave's 64 x 64 block
  // (three-product form: [split][batch][3 tiles][128][128])
  float* ws = p.gauss ? p.ws + (((long long)zs * p.batch + zb) * 3 + tile) * (TN_W * TN_W)
                      : p.ws + ((long long)zs * p.batch + zb) * p.N1 * p.N2;
  const int ldw = p.gauss ? TN_W : p.N2;
  const int n1b = (p.gauss ? 0 : t1 * TN_W) + wm * 64, n2b = (p.gauss ? 0 : t2 * TN_W) + wn * 64 + 4 * i16;
#pragma unroll
  for (int ea = 0; ea < 4; ++ea)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n1 = n1b + 4 * (4 * kq + r) + ea;
      *reinterpret_cast<float4*>(ws + (long long)n1 * ldw + n2b) =
          make_float4(acc[ea][0][r], acc[ea][1][r], acc[ea][2][r], acc[ea][3][r]);
    }
  if (cs_a || cs_b) {
    // lanes with equal i16 hold the sums over the tokens = kq (mod 4): fixed-order butterfly over kq
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = cs[e];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      cs[e] = v;
    }
    if (kq == 0) {
      const int L = p.csL ? p.csL : (cs_a ? p.N1 : p.N2);
      const int g0 = cs_a ? t1 * TN_W + wm * 64 : t2 * TN_W + wn * 64;
      const long long prod = p.gauss ? 3ll * TN_W * TN_W : (long long)p.N1 * p.N2;
      float* wc = p.ws + (long long)p.splits * p.batch * prod + ((long long)zs * p.batch + zb) * L + g0 + 4 * i16;
      *reinterpret_cast<float4*>(wc) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// 192 x 192 variant (round 3): the AFNO weight gradients of DPOT-Large (bs = 96 -> N = 2*bs = 192, not a multiple of the
// 128-wide tile above; they ran on the generic kernel of gemm.hip at 68 TFLOP/s, 5 % of the DPOT-L step).  One workgroup
// owns the WHOLE 192 x 192 output of a (channel block, layer) problem x a token range: 3 x 3 compute waves of 64 x 64
// (same row-interleaved fragments, same 16-byte stores) + 2 loader waves; slabs of 32 tokens = [32][192] per operand
// (48 KiB), ring of three; a DMA piece is 1 KiB of the slab's linear image - 192 floats per token is a multiple of the
// 16-byte lane chunk, so a piece never splits a chunk across rows.  Nine waves on four SIMDs leave one SIMD with three:
// the matrix pipes can reach 75 % at best - against 43 % on the generic kernel.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TW = 192;                      // tile width (both operands)
constexpr int TW_SLABF = 2 * TN_TOK * TW;    // floats per slab: A [32][192] | B [32][192]
constexpr int TW_RING = 3;
constexpr int TW_NI = 24;                    // DMA instructions per slab and loader wave (12 of A + 12 of B)

__global__ __launch_bounds__(704) void gemm_tn192_kernel(const TnArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[TW_RING * TW_SLABF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int zb = blockIdx.z / p.splits, zs = blockIdx.z - zb * p.splits;
  const int nslab_all = p.T / TN_TOK;
  const int slab0 = zs * p.slabs_per_split;
  int nslab = nslab_all - slab0;
  nslab = nslab < p.slabs_per_split ? nslab : p.slabs_per_split;
  if (nslab < 0) nslab = 0;

  auto bar = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (wave >= 9) {
    // ---------------------------------- loader waves ----------------------------------
    const int L = wave - 9;
    const bool second = zb >= p.batch1;
    const int zq = second ? zb - p.batch1 : zb;
    const float* abase = (second ? p.A2 : p.A) + zq * p.sA + (long long)slab0 * TN_TOK * p.lda;
    const float* bbase = (second ? p.B2 : p.B) + zq * p.sB + (long long)slab0 * TN_TOK * p.ldb;
    auto issue = [&](int t, int ring) __attribute__((always_inline)) {
      float* dst = lds + ring * TW_SLABF;
      const float* a = abase + (long long)t * TN_TOK * p.lda;
      const float* b = bbase + (long long)t * TN_TOK * p.ldb;
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int f = 256 * (L + 2 * j) + 4 * lane;     // float index inside the operand's [32][192] image
        const int tok = f / TW, col = f - tok * TW;
        tn_glds16(a + (long long)tok * p.lda + col, dst + 256 * (L + 2 * j));
      }
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int f = 256 * (L + 2 * j) + 4 * lane;
        const int tok = f / TW, col = f - tok * TW;
        tn_glds16(b + (long long)tok * p.ldb + col, dst + TN_TOK * TW + 256 * (L + 2 * j));
      }
    };
    if (nslab > 0) issue(0, 0);
    if (nslab > 1) issue(1, 1);
    if (nslab > 1) tn_wait_vm<TW_NI>(); else tn_wait_vm<0>();
    bar();                                              // P: slab 0 landed
    int ring = 2;                                       // slot of slab g + 2
#pragma unroll 1
    for (int g = 0; g < nslab; ++g) {
      tn_wait_vm<0>();                                  // slab g + 1 landed (the compute waves prefetch its first fragments)
      bar();                                            // B_g: everyone is done with slab g - 1 -> its slot is free
      if (g + 2 < nslab) issue(g + 2, ring);
      ring = ring == TW_RING - 1 ? 0 : ring + 1;
    }
    return;
  }

  // ---------------------------------- compute waves ----------------------------------
  const int wm = wave / 3, wn = wave - 3 * wm;
  const int i16 = lane & 15, kq = lane >> 4;
  tn_f32x4 acc[4][4];
#pragma unroll
  for (int ea = 0; ea < 4; ++ea)
#pragma unroll
    for (int eb = 0; eb < 4; ++eb) acc[ea][eb] = tn_f32x4{0.f, 0.f, 0.f, 0.f};
  tn_f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  const int cs_sel = zb >= p.batch1 ? p.cs_of2 : p.cs_of;
  const bool cs_a = cs_sel == 1 && wn == 0;
  const bool cs_b = cs_sel == 2 && wm == 0;
  const int offA = kq * TW + wm * 64 + 4 * i16;
  const int offB = TN_TOK * TW + kq * TW + wn * 64 + 4 * i16;
  constexpr int NQ = TN_TOK / 4;

  bar();                                                // P
  tn_f32x4 fa = {0.f, 0.f, 0.f, 0.f}, fb = fa;
  if (nslab > 0) {
    fa = *reinterpret_cast<const tn_f32x4*>(lds + offA);
    fb = *reinterpret_cast<const tn_f32x4*>(lds + offB);
  }
  int ring = 0;
#pragma unroll 1
  for (int g = 0; g < nslab; ++g) {
    bar();                                              // B_g
    const float* cur = lds + ring * TW_SLABF;
    const int rn = ring == TW_RING - 1 ? 0 : ring + 1;
    const float* nxt = g + 1 < nslab ? lds + rn * TW_SLABF : cur;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const tn_f32x4 a = fa, b = fb;
      const float* src = q + 1 < NQ ? cur + (q + 1) * 4 * TW : nxt;
      fa = *reinterpret_cast<const tn_f32x4*>(src + offA);
      fb = *reinterpret_cast<const tn_f32x4*>(src + offB);
      if (cs_a) cs += a;
      if (cs_b) cs += b;
#pragma unroll
      for (int ea = 0; ea < 4; ++ea)
#pragma unroll
        for (int eb = 0; eb < 4; ++eb)
          acc[ea][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ea], b[eb], acc[ea][eb], 0, 0, 0);
    }
    ring = rn;
  }

  float* ws = p.ws + ((long long)zs * p.batch + zb) * p.N1 * p.N2;
  const int n1b = wm * 64, n2b = wn * 64 + 4 * i16;
#pragma unroll
  for (int ea = 0; ea < 4; ++ea)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n1 = n1b + 4 * (4 * kq + r) + ea;
      *reinterpret_cast<float4*>(ws + (long long)n1 * p.N2 + n2b) =
          make_float4(acc[ea][0][r], acc[ea][1][r], acc[ea][2][r], acc[ea][3][r]);
    }
  if (cs_a || cs_b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = cs[e];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      cs[e] = v;
    }
    if (kq == 0) {
      const int Lc = p.csL ? p.csL : (cs_a ? p.N1 : p.N2);
      const int g0 = cs_a ? wm * 64 : wn * 64;
      float* wc = p.ws + (long long)p.splits * p.batch * p.N1 * p.N2 + ((long long)zs * p.batch + zb) * Lc + g0 + 4 * i16;
      *reinterpret_cast<float4*>(wc) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// 96-channel three-product variant (round 5): the AFNO weight gradients of DPOT-Large in the Gauss form of TnArgs::gauss
//     P1 = Ar^T Br,  P2 = Ai^T Bi,  P3 = (Ar + Ai)^T (Bi - Br)          (three 96 x 96 products instead of 192 x 192)
// on TWELVE compute waves - product p on waves 4p .. 4p+3 in a 2 x 2 grid of 48 x 48 (3 x 3 accumulators) - + 2 loader
// waves: 14 waves put 3 compute waves on every SIMD (the 192 x 192 kernel above has 3 / 2 / 2 / 2: its matrix pipes can
// reach 75 % at best), and the launch executes 3/4 of the FLOPs.  Same slabs ([32 tok][192] per operand, ring of three),
// same loaders, same split-K.  Fragments: tile e of a wave = 16 consecutive channels, one ds_read_b32 per tile and 4-token
// step (consecutive lanes, consecutive floats); the P3 waves read both halves of both operands and form the sums on the
// VALU.  Partials [split][problem][3][96][96] (+ column sums of dO from the P1 / P2 waves) - afno_wgrad2_reduce_body's
// three-product branch recombines.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TG_BS = 96;
constexpr int TGS = 208;                     // LDS row stride (floats): 192 + 16 - the four token groups of a ds_read_b32 (rows
                                             // kq, kq+1, ..) start 16 banks apart; with 192 (= 0 mod 64 banks) they collide 4-way
constexpr int TG_SLABF = 2 * TN_TOK * TGS;   // floats per slab: A [32][208] | B [32][208]  (3 slabs = 156 KiB)
constexpr int TG_NI = 32;                    // DMA instructions per slab and loader wave: one token row each (48 lanes x 16 B)

__global__ __launch_bounds__(896) void gemm_tn96g_kernel(const TnArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[TW_RING * TG_SLABF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int zb = blockIdx.z / p.splits, zs = blockIdx.z - zb * p.splits;
  const int nslab_all = p.T / TN_TOK;
  const int slab0 = zs * p.slabs_per_split;
  int nslab = nslab_all - slab0;
  nslab = nslab < p.slabs_per_split ? nslab : p.slabs_per_split;
  if (nslab < 0) nslab = 0;

  auto bar = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (wave >= 12) {
    // ---------------------------------- loader waves (as in gemm_tn192_kernel) ----------------------------------
    const int L = wave - 12;
    const bool second = zb >= p.batch1;
    const int zq = second ? zb - p.batch1 : zb;
    const float* abase = (second ? p.A2 : p.A) + zq * p.sA + (long long)slab0 * TN_TOK * p.lda;
    const float* bbase = (second ? p.B2 : p.B) + zq * p.sB + (long long)slab0 * TN_TOK * p.ldb;
    auto issue = [&](int t, int ring) __attribute__((always_inline)) {
      float* dst = lds + ring * TG_SLABF;
      const float* a = abase + (long long)t * TN_TOK * p.lda + 4 * lane;
      const float* b = bbase + (long long)t * TN_TOK * p.ldb + 4 * lane;
      if (lane < 48) {                                  // a token row = 192 floats = 48 lanes x 16 B, padded to TGS in LDS
#pragma unroll
        for (int j = 0; j < 16; ++j) tn_glds16(a + (long long)(L + 2 * j) * p.lda, dst + (L + 2 * j) * TGS);
#pragma unroll
        for (int j = 0; j < 16; ++j) tn_glds16(b + (long long)(L + 2 * j) * p.ldb, dst + (TN_TOK + L + 2 * j) * TGS);
      }
    };
    if (nslab > 0) issue(0, 0);
    if (nslab > 1) issue(1, 1);
    if (nslab > 1) tn_wait_vm<TG_NI>(); else tn_wait_vm<0>();
    bar();                                              // P
    int ring = 2;
#pragma unroll 1
    for (int g = 0; g < nslab; ++g) {
      tn_wait_vm<0>();
      bar();                                            // B_g
      if (g + 2 < nslab) issue(g + 2, ring);
      ring = ring == TW_RING - 1 ? 0 : ring + 1;
    }
    return;
  }

  // ---------------------------------- compute waves ----------------------------------
  const int prod = wave >> 2;                           // 0: Ar^T Br, 1: Ai^T Bi, 2: (Ar + Ai)^T (Bi - Br)
  const int wm = (wave >> 1) & 1, wn = wave & 1;
  const int i16 = lane & 15, kq = lane >> 4;
  tn_f32x4 acc[3][3];
#pragma unroll
  for (int ea = 0; ea < 3; ++ea)
#pragma unroll
    for (int eb = 0; eb < 3; ++eb) acc[ea][eb] = tn_f32x4{0.f, 0.f, 0.f, 0.f};
  float cs[3] = {0.f, 0.f, 0.f};
  const int cs_sel = zb >= p.batch1 ? p.cs_of2 : p.cs_of;
  const bool cs_a = cs_sel == 1 && prod < 2 && wn == 0;
  const bool cs_b = cs_sel == 2 && prod < 2 && wm == 0;
  const int half = prod == 1 ? TG_BS : 0;               // P2 reads the imaginary halves; P3 reads both (half = 0, + TG_BS)
  const int offA = kq * TGS + half + wm * 48 + i16;
  const int offB = TN_TOK * TGS + kq * TGS + half + wn * 48 + i16;
  constexpr int NQ = TN_TOK / 4;

  bar();                                                // P
  if (prod < 2) {
    float fa[3] = {0.f, 0.f, 0.f}, fb[3] = {0.f, 0.f, 0.f};
    if (nslab > 0) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        fa[e] = lds[offA + 16 * e];
        fb[e] = lds[offB + 16 * e];
      }
    }
    int ring = 0;
#pragma unroll 1
    for (int g = 0; g < nslab; ++g) {
      bar();                                            // B_g
      const float* cur = lds + ring * TG_SLABF;
      const int rn = ring == TW_RING - 1 ? 0 : ring + 1;
      const float* nxt = g + 1 < nslab ? lds + rn * TG_SLABF : cur;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        float a[3], b[3];
        const float* src = q + 1 < NQ ? cur + (q + 1) * 4 * TGS : nxt;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          a[e] = fa[e];
          b[e] = fb[e];
          fa[e] = src[offA + 16 * e];
          fb[e] = src[offB + 16 * e];
        }
        if (cs_a) { cs[0] += a[0]; cs[1] += a[1]; cs[2] += a[2]; }
        if (cs_b) { cs[0] += b[0]; cs[1] += b[1]; cs[2] += b[2]; }
#pragma unroll
        for (int ea = 0; ea < 3; ++ea)
#pragma unroll
          for (int eb = 0; eb < 3; ++eb)
            acc[ea][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ea], b[eb], acc[ea][eb], 0, 0, 0);
      }
      ring = rn;
    }
  } else {
    float far[3] = {0.f, 0.f, 0.f}, fai[3] = {0.f, 0.f, 0.f}, fbr[3] = {0.f, 0.f, 0.f}, fbi[3] = {0.f, 0.f, 0.f};
    if (nslab > 0) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        far[e] = lds[offA + 16 * e];
        fai[e] = lds[offA + TG_BS + 16 * e];
        fbr[e] = lds[offB + 16 * e];
        fbi[e] = lds[offB + TG_BS + 16 * e];
      }
    }
    int ring = 0;
#pragma unroll 1
    for (int g = 0; g < nslab; ++g) {
      bar();                                            // B_g
      const float* cur = lds + ring * TG_SLABF;
      const int rn = ring == TW_RING - 1 ? 0 : ring + 1;
      const float* nxt = g + 1 < nslab ? lds + rn * TG_SLABF : cur;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        float a[3], b[3];
        const float* src = q + 1 < NQ ? cur + (q + 1) * 4 * TGS : nxt;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          a[e] = far[e] + fai[e];
          b[e] = fbi[e] - fbr[e];
          far[e] = src[offA + 16 * e];
          fai[e] = src[offA + TG_BS + 16 * e];
          fbr[e] = src[offB + 16 * e];
          fbi[e] = src[offB + TG_BS + 16 * e];
        }
#pragma unroll
        for (int ea = 0; ea < 3; ++ea)
#pragma unroll
          for (int eb = 0; eb < 3; ++eb)
            acc[ea][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ea], b[eb], acc[ea][eb], 0, 0, 0);
      }
      ring = rn;
    }
  }

  // partial tile: element (row = wm*48 + 16*ea + 4*kq + r, col = wn*48 + 16*eb + i16) of product `prod`
  float* ws = p.ws + (((long long)zs * p.batch + zb) * 3 + prod) * (TG_BS * TG_BS);
#pragma unroll
  for (int ea = 0; ea < 3; ++ea)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* row = ws + (long long)(wm * 48 + 16 * ea + 4 * kq + r) * TG_BS + wn * 48 + i16;
#pragma unroll
      for (int eb = 0; eb < 3; ++eb) row[16 * eb] = acc[ea][eb][r];
    }
  if (cs_a || cs_b) {
    // lanes with equal i16 hold the sums over the tokens = kq (mod 4): fixed-order butterfly over kq
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      float v = cs[e];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      cs[e] = v;
    }
    if (kq == 0) {
      const int Lc = p.csL ? p.csL : (cs_a ? p.N1 : p.N2);
      const int g0 = half + (cs_a ? wm : wn) * 48 + i16;
      float* wc = p.ws + (long long)p.splits * p.batch * (3ll * TG_BS * TG_BS) + ((long long)zs * p.batch + zb) * Lc + g0;
#pragma unroll
      for (int e = 0; e < 3; ++e) wc[16 * e] = cs[e];
    }
  }
}

}  // namespace dpot

using namespace dpot;

// split count the TN kernel wants for a shape (0: the shape is not eligible)
extern "C" int dpot_gemm_tn_splitk(int M, int N, int K, int batch) {
  static const int enabled = tune("panel", 1);
  // batched descriptors (the AFNO weight gradients: 4 x [256 x 256], 4608 tokens) stay on the generic kernel: measured
  // 37.4 us here against 36.2 us there
  if (!enabled || M <= 0 || N <= 0 || K <= 0 || batch != 1 || M % TN_W || N % TN_W || K % TN_TOK) return 0;
  const long long tiles = (long long)(M / TN_W) * (N / TN_W) * batch;
  const int nslab = K / TN_TOK;
  long long s = (256 + tiles / 2) / tiles;            // one workgroup per CU
  const long long smax = nslab / 4;                   // >= 4 slabs (128 tokens) per split
  if (s > smax) s = smax;
  if (s < 2) s = 2;                                   // the kernel always goes through the workspace + reduce
  if (s > nslab) return 0;
  const long long sps = (nslab + s - 1) / s;          // slabs per split; drop the splits that would come out empty
  s = (nslab + sps - 1) / sps;                        // (e.g. 256 slabs / 21 splits -> 13 per split -> 20 splits)
  if (s < 2) return 0;
  return (int)s;
}

// called by dpot_gemm_f32 (gemm.hip) for transA && !transB, native fp32, split-K descriptors; returns -1 when the
// descriptor is not eligible (the generic kernel runs instead), else the launch status.  The partial sums land in
// d->workspace in the layout the generic kernel uses, so the caller's reduce launch is unchanged.
int dpot_gemm_tn_try(const dpot_gemm_desc* d, hipStream_t s) {
  static const int enabled = tune("panel", 1);
  if (!enabled || !d->transA || d->transB || d->splitk <= 1 || !d->workspace || d->batch != 1) return -1;
  if (d->M % TN_W || d->N % TN_W || d->K % TN_TOK || d->lda % 4 || d->ldb % 4 || d->strideA % 4 || d->strideB % 4 ||
      !aligned16(d->A) || !aligned16(d->B) || !aligned16(d->workspace) || d->N % 4)
    return -1;
  const int nslab = d->K / TN_TOK;
  if (d->splitk > nslab || (long long)d->batch * d->splitk > 65535) return -1;
  TnArgs p;
  p.A = d->A; p.B = d->B; p.lda = d->lda; p.ldb = d->ldb; p.sA = d->strideA; p.sB = d->strideB;
  p.A2 = nullptr; p.B2 = nullptr; p.batch1 = d->batch; p.cs_of2 = d->colsum_of; p.csL = 0;
  p.N1 = d->M; p.N2 = d->N; p.T = d->K; p.batch = d->batch;
  p.tiles1 = d->M / TN_W; p.tiles2 = d->N / TN_W;
  p.splits = d->splitk;
  p.slabs_per_split = (nslab + d->splitk - 1) / d->splitk;
  if ((long long)p.slabs_per_split * (d->splitk - 1) >= nslab) return -1;      // an empty split would leave its partial unwritten
  p.ws = d->workspace;
  p.cs_of = d->colsum_of;
  p.gauss = 0; p.splits12 = p.splits; p.slabs_per_split12 = p.slabs_per_split;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(p.tiles1 * p.tiles2), 1, (unsigned)(d->batch * d->splitk)), dim3(384),
                     0, s, p);
  return check_launch("gemm_tn_kernel");
}


// ---------------------------------------------------------------------------------------------------------------------
// Both weight gradients of an AFNO block's complex MLP in ONE launch: per channel block k
//     dWbig1[k] = S[:, k]^T dO1pre[:, k],    dWbig2[k] = O1[:, k]^T dO2[:, k]        (N x N, N = 2*bs, K = Mm tokens)
// = 2*nb independent problems of the kernel above.  One launch halves the split factor each problem needs to fill the
// chip (18 instead of 9 slabs per workgroup, half the partial-sum traffic) against two launches of the generic kernel:
// 2 x 36.5 us -> one launch + one reduce.  The reduce un-packs  dWr = D[0:bs,0:bs] + D[bs:,bs:],
// dWi = D[0:bs,bs:] - D[bs:,0:bs]  and the bias gradients (column sums of dO) for both layers, fixed order.
// ---------------------------------------------------------------------------------------------------------------------
namespace dpot {
// sum_s w[s * stride], s < splits: always the same association ((0+1)+(2+3))+((4+5)+(6+7)) per group of eight, groups in
// order - deterministic - with eight independent loads in flight (a dependent chain of `splits` L2 round trips made the
// reduce kernels latency bound: 16 MB in 12 us)
__device__ __forceinline__ float tn_sum_splits(const float* __restrict__ w, long long stride, int splits) {
  float v = 0.f;
  int s = 0;
  for (; s + 8 <= splits; s += 8) {
    const float a0 = w[(long long)(s + 0) * stride], a1 = w[(long long)(s + 1) * stride];
    const float a2 = w[(long long)(s + 2) * stride], a3 = w[(long long)(s + 3) * stride];
    const float a4 = w[(long long)(s + 4) * stride], a5 = w[(long long)(s + 5) * stride];
    const float a6 = w[(long long)(s + 6) * stride], a7 = w[(long long)(s + 7) * stride];
    v += ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  }
  for (; s + 2 <= splits; s += 2) {
    const float a0 = w[(long long)s * stride], a1 = w[(long long)(s + 1) * stride];
    v += a0 + a1;
  }
  if (s < splits) v += w[(long long)s * stride];
  return v;
}

// (bid, nblk: this workgroup's index / the number of workgroups working on the reduction - the stand-alone kernel passes
// blockIdx.x / gridDim.x, block_finalize_kernel a slice of its grid)
__device__ __forceinline__ void afno_wgrad2_reduce_body(int bid, int nblk, const float* __restrict__ ws, int splits, int nb,
                                                        int bs, float* __restrict__ dw1, float* __restrict__ db1,
                                                        float* __restrict__ dw2, float* __restrict__ db2, int gauss) {
  const int n2 = 2 * bs;
  const long long MN = gauss ? 3ll * bs * bs : (long long)n2 * n2, total = MN * 2 * nb;
  const long long nw = (long long)nb * bs * bs;
  for (long long idx = bid * 256ll + threadIdx.x; idx < 2 * nw; idx += (long long)nblk * 256) {
    const int layer = idx >= nw;
    const long long q = idx - layer * nw;
    const int o = (int)(q % bs), i = (int)((q / bs) % bs), k = (int)(q / ((long long)bs * bs));
    const float* base = ws + (long long)(layer * nb + k) * MN;
    float* dw = layer ? dw2 : dw1;
    if (gauss) {     // P1 = Ar^T Br, P2 = Ai^T Bi, P3 = (Ar + Ai)^T (Bi - Br): dWr = P1 + P2, dWi = P3 + P1 - P2
      // (gauss = the number of token ranges of P1 / P2 and of the column sums, TnArgs::splits12 - <= splits)
      const long long e = (long long)i * bs + o, t = (long long)bs * bs;
      const float p1 = tn_sum_splits(base + e, total, gauss);
      const float p2 = tn_sum_splits(base + t + e, total, gauss);
      const float p3 = tn_sum_splits(base + 2 * t + e, total, splits);
      dw[q] = p1 + p2;
      dw[nw + q] = p3 + p1 - p2;
      continue;
    }
    const float rr = tn_sum_splits(base + (long long)i * n2 + o, total, splits);
    const float ii = tn_sum_splits(base + (long long)(bs + i) * n2 + bs + o, total, splits);
    const float ri = tn_sum_splits(base + (long long)i * n2 + bs + o, total, splits);
    const float ir = tn_sum_splits(base + (long long)(bs + i) * n2 + o, total, splits);
    dw[q] = rr + ii;
    dw[nw + q] = ri - ir;
  }
  const float* wc = ws + (long long)splits * total;
  const long long nc = (long long)2 * nb * n2;
  for (long long idx = bid * 256ll + threadIdx.x; idx < nc; idx += (long long)nblk * 256) {
    const float v = tn_sum_splits(wc + idx, nc, gauss ? gauss : splits);
    const int layer = idx >= (long long)nb * n2;
    const long long q = idx - (long long)layer * nb * n2;
    const int c = (int)(q % bs), part = (int)((q / bs) & 1), k = (int)(q / n2);
    (layer ? db2 : db1)[((long long)part * nb + k) * bs + c] = v;
  }
}
__global__ __launch_bounds__(256) void afno_wgrad2_reduce_kernel(const float* __restrict__ ws, int splits, int nb, int bs,
                                                                 float* __restrict__ dw1, float* __restrict__ db1,
                                                                 float* __restrict__ dw2, float* __restrict__ db2, int gauss) {
  afno_wgrad2_reduce_body(blockIdx.x, gridDim.x, ws, splits, nb, bs, dw1, db1, dw2, db2, gauss);
}

// three-product form of the AFNO weight gradient (TnArgs::gauss): 128 channels per block, and 96 (DPOT-Large): gemm_tn96g_kernel;
// DPOT_TUNE wgrad_gauss=0: four products (128: four tiles of gemm_tn_kernel; 96: the 192 x 192 kernel)
static int tn_gauss(int bs) {
  static const int enabled = tune("wgrad_gauss", 1);
  return enabled && (bs == TN_W || bs == TG_BS) ? 1 : 0;
}
// token ranges of the tiles P1 / P2 when the sum-product tile is cut into `splitk` (TnArgs::splits12): 5/6 of them at 128 channels
// per block (profiles/r05_tn_skew.txt); the 96-channel kernel computes the three products in one workgroup
static int tn_gauss_s12(int bs, int splitk) {
  constexpr int num = 5, den = 6;
  if (bs != TN_W) return splitk;
  const int s = splitk * num / den;
  return s < 1 ? 1 : s;
}
}  // namespace dpot

extern "C" int dpot_afno_wgrad2_splitk(int Mm, int nb, int bs) {
  static const int enabled = tune("panel", 1);
  const int N = 2 * bs;
  if (!enabled || nb <= 0 || bs <= 0 || (N % TN_W && N != TW) || Mm <= 0 || Mm % TN_TOK) return 0;
  const long long tiles = N == TW ? (long long)2 * nb : tn_gauss(bs) ? 6ll * nb : (long long)2 * nb * (N / TN_W) * (N / TN_W);
  const int nslab = Mm / TN_TOK;
  // three-product form: 3 tiles per problem do not divide 256 - round DOWN (one round of workgroups: 24 tiles x 10 splits = 240
  // at DPOT-Tiny; rounded to 11 the launch needs a second round: 75 against 59 us, profiles/r05_tn_bench_gauss.txt)
  long long s = tn_gauss(bs) && N != TW ? 256 / tiles : (256 + tiles / 2) / tiles;
  const long long smax = nslab / 4;
  if (tn_gauss(bs) && N != TW) {
    // skewed ranges: the largest split count whose 2 * s12 + s workgroups per problem still fit one round of 256
    // (DPOT-Tiny: 8 problems x (10 + 10 + 12), DPOT-S / -M: 16 x (5 + 5 + 6))
    long long best = 1;
    for (long long c = 1; c <= smax && c <= 64; ++c)
      if (2ll * nb * (2 * tn_gauss_s12(bs, (int)c) + c) <= 256) best = c;
    s = best;
  }
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  // no empty split
  while (s > 1 && ((nslab + s - 1) / s) * (s - 1) >= nslab) --s;
  return (int)s;
}

extern "C" int64_t dpot_afno_wgrad2_ws_elems(int nb, int bs, int splitk) {
  const int64_t N = 2 * bs;
  return (int64_t)splitk * 2 * nb * ((tn_gauss(bs) ? 3 * (int64_t)bs * bs : N * N) + N);
}

extern "C" int dpot_afno_wgrad2(const float* S, const float* dO1pre, const float* O1, const float* dO2, int ld, int Mm,
                                int nb, int bs, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                                int splitk, dpot_stream_t stream) {
  DPOT_REQUIRE(S && dO1pre && O1 && dO2 && workspace, "afno_wgrad2: null pointer");
  DPOT_REQUIRE((dw1 && db1 && dw2 && db2) || (!dw1 && !db1 && !dw2 && !db2), "afno_wgrad2: outputs must be all given or all NULL");
  const int N = 2 * bs;
  DPOT_REQUIRE(nb > 0 && bs > 0 && (N % TN_W == 0 || N == TW) && Mm > 0 && Mm % TN_TOK == 0 && ld >= nb * N && ld % 4 == 0,
               "afno_wgrad2: needs 2*bs %% 128 == 0 or 2*bs == 192, Mm %% 32 == 0");
  DPOT_REQUIRE(aligned16(S) && aligned16(dO1pre) && aligned16(O1) && aligned16(dO2) && aligned16(workspace),
               "afno_wgrad2: operands must be 16-byte aligned");
  const int nslab = Mm / TN_TOK;
  DPOT_REQUIRE(splitk >= 1 && splitk <= nslab && (long long)2 * nb * splitk <= 65535, "afno_wgrad2: bad split factor");
  TnArgs p;
  p.A = S; p.B = dO1pre; p.A2 = O1; p.B2 = dO2; p.batch1 = nb; p.cs_of2 = 2; p.csL = 0;
  p.lda = ld; p.ldb = ld; p.sA = N; p.sB = N;
  p.N1 = N; p.N2 = N; p.T = Mm; p.batch = 2 * nb;
  p.tiles1 = N / TN_W; p.tiles2 = N / TN_W;
  p.splits = splitk;
  p.slabs_per_split = (nslab + splitk - 1) / splitk;
  DPOT_REQUIRE((long long)p.slabs_per_split * (splitk - 1) < nslab, "afno_wgrad2: split factor leaves an empty split");
  p.ws = workspace;
  p.cs_of = 2;
  p.gauss = tn_gauss(bs);
  p.splits12 = p.gauss ? tn_gauss_s12(bs, splitk) : splitk;
  p.slabs_per_split12 = (nslab + p.splits12 - 1) / p.splits12;
  hipStream_t s = as_stream(stream);
  if (N == TW) {
    p.tiles1 = p.tiles2 = 1;
    if (p.gauss)
      hipLaunchKernelGGL(gemm_tn96g_kernel, dim3(1, 1, (unsigned)(2 * nb * splitk)), dim3(896), 0, s, p);
    else
      hipLaunchKernelGGL(gemm_tn192_kernel, dim3(1, 1, (unsigned)(2 * nb * splitk)), dim3(704), 0, s, p);
  } else {
    if (p.gauss)
      hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(2 * nb * (2 * p.splits12 + splitk))), dim3(384), 0, s, p);
    else
      hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(p.tiles1 * p.tiles2), 1, (unsigned)(2 * nb * splitk)), dim3(384), 0, s, p);
  }
  int rc = check_launch("gemm_tn_kernel");
  if (rc || !dw1) return rc;            // dw1 == NULL: partials only, dpot_block_finalize reduces them
  long long blocks = (2ll * nb * bs * bs + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(afno_wgrad2_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)workspace, splitk, nb,
                     bs, dw1, db1, dw2, db2, p.gauss ? p.splits12 : 0);
  return check_launch("afno_wgrad2_reduce_kernel");
}


// ---------------------------------------------------------------------------------------------------------------------
// Both weight gradients of a channel MLP (y = fc2(act(fc1(x)))) in one launch:
//     dW2 [E, mh]   = do2^T Hh        + db2 = colsum(do2)      (problem 0: column sums of the A operand)
//     dW1^T [E, mh] = xn2^T dHpre     + db1 = colsum(dHpre)    (problem 1: of the B operand; stored transposed -> dW1 [mh, E])
// Two same-shaped problems -> half the split factor of two separate launches (less partial traffic, longer token ranges).
// ---------------------------------------------------------------------------------------------------------------------
namespace dpot {
__device__ __forceinline__ void mlp_wgrad2_reduce_body(int bid, int nblk, const float* __restrict__ ws, int splits, int E,
                                                       int mh, float* __restrict__ dW2, float* __restrict__ dW1,
                                                       float* __restrict__ db2, float* __restrict__ db1) {
  __shared__ float tile[32][33];
  const long long MN = (long long)E * mh, total = 2 * MN;
  const int L = E > mh ? E : mh;
  // 32 x 32 tiles of the [E, mh] outputs: problem 0 is copied, problem 1 goes out transposed (coalesced both ways)
  const int tn = mh / 32, tcount = (E / 32) * tn;
  for (int tix = bid; tix < 2 * tcount; tix += nblk) {
    const int prob = tix >= tcount, tt = tix - prob * tcount;
    const int r0 = (tt / tn) * 32, c0 = (tt % tn) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + ty + 8 * q;
      const float v = tn_sum_splits(ws + (long long)prob * MN + (long long)r * mh + c0 + tx, total, splits);
      if (prob == 0) dW2[(long long)r * mh + c0 + tx] = v;
      else tile[ty + 8 * q][tx] = v;
    }
    if (prob) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = c0 + ty + 8 * q;                                   // row of dW1 [mh, E]
        dW1[(long long)c * E + r0 + tx] = tile[tx][ty + 8 * q];
      }
      __syncthreads();
    }
  }
  const float* wc = ws + (long long)splits * total;
  for (int idx = bid * 256 + threadIdx.x; idx < E + mh; idx += nblk * 256) {
    const int prob = idx >= E, g = idx - prob * E;
    (prob ? db1 : db2)[g] = tn_sum_splits(wc + (long long)prob * L + g, 2ll * L, splits);
  }
}
__global__ __launch_bounds__(256) void mlp_wgrad2_reduce_kernel(const float* __restrict__ ws, int splits, int E, int mh,
                                                                float* __restrict__ dW2, float* __restrict__ dW1,
                                                                float* __restrict__ db2, float* __restrict__ db1) {
  mlp_wgrad2_reduce_body(blockIdx.x, gridDim.x, ws, splits, E, mh, dW2, dW1, db2, db1);
}

// ---------------------------------------------------------------------------------------------------------------------
// ONE finalising launch per DPOT block (round 4): the fixed-order reductions that end a block's backward - the split-K
// partials of the two AFNO weight gradients (+ un-packing to the parameters' layout, bias gradients), those of the two
// channel-MLP weight gradients (+ bias gradients), and the per-sample partials of both GroupNorms' parameter gradients -
// were three launches of a few microseconds each (DPOT-Tiny: 8.1 + 6.5 + 5.3 us per block as eager kernel time, ~4 us each
// inside the replayed graph).  The grid is cut into three slices; each slice runs the body of the stand-alone kernel
// unchanged, so every sum keeps its order (bit-identical results).  A slice of zero workgroups is skipped.
// ---------------------------------------------------------------------------------------------------------------------
struct FinalizeArgs {
  // slice A: AFNO weight gradients (dpot_afno_wgrad2's reduction)
  const float* a_ws; int a_splits, a_nb, a_bs, a_gauss; float *a_dw1, *a_db1, *a_dw2, *a_db2; int nA;
  // slice M: channel-MLP weight gradients (dpot_mlp_wgrad2's reduction)
  const float* m_ws; int m_splits, m_E, m_mh; float *m_dW2, *m_dW1, *m_db2, *m_db1; int nM;
  // slice G: GroupNorm parameter gradients (dpot_groupnorm_param_grads): g_jobs jobs x ceil(E / 64) workgroups
  const float* g_part[2]; float* g_dgamma[2]; float* g_dbeta[2]; int g_jobs, g_B, g_E, nG;
  // slice C: column sums of up to two partial matrices [rows, N] (the bias gradients of the bf16 channel MLP, whose
  // producers - the pack pass and the GEMM epilogue - leave per-row-tile partial sums): out[n] = sum_r part[r, n]
  const float* c_part[2]; float* c_out[2]; int c_rows[2], c_N[2], c_blocks[2];
};
// a workgroup = 32 columns x 8 row lanes (128-byte row segments); a lane strides the rows by 8 with four accumulators
// (four loads in flight), the eight lanes of a column are combined through LDS in a fixed order.  (One thread per column
// over all rows - 16 workgroups of 256 dependent loads at DPOT-M - made the whole finalising launch slower than the
// launches it replaced: DPOT-M 13.23 -> 13.38 ms, profiles/r04_block_finalize_colsum.txt.)
constexpr int CS_COLS = 32, CS_LANES = 8;
__device__ __forceinline__ void colsum_rows_body(int bx, const float* __restrict__ part, float* __restrict__ out, int rows,
                                                 int N) {
  __shared__ float red[CS_LANES][CS_COLS];
  const int tc = threadIdx.x % CS_COLS, tr = threadIdx.x / CS_COLS;
  const int n = bx * CS_COLS + tc;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (n < N) {
    int r = tr;
    for (; r + 3 * CS_LANES < rows; r += 4 * CS_LANES) {
      s0 += part[(long long)r * N + n];
      s1 += part[(long long)(r + CS_LANES) * N + n];
      s2 += part[(long long)(r + 2 * CS_LANES) * N + n];
      s3 += part[(long long)(r + 3 * CS_LANES) * N + n];
    }
    for (; r < rows; r += CS_LANES) s0 += part[(long long)r * N + n];
  }
  red[tr][tc] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (tr == 0 && n < N) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < CS_LANES; ++k) v += red[k][tc];
    out[n] = v;
  }
}
// = norm.hip groupnorm_param_grad_kernel: block = 64 channels x 4 sample lanes, fixed order
__device__ __forceinline__ void gn_param_grad_body(int bx, const float* __restrict__ part, float* __restrict__ dgamma,
                                                   float* __restrict__ dbeta, int B, int E) {
  __shared__ float red[2][4][64];
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
  const int c = bx * 64 + tc;
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
__global__ __launch_bounds__(256) void block_finalize_kernel(const FinalizeArgs a) {
  const int b = blockIdx.x;
  if (b < a.nA) {
    afno_wgrad2_reduce_body(b, a.nA, a.a_ws, a.a_splits, a.a_nb, a.a_bs, a.a_dw1, a.a_db1, a.a_dw2, a.a_db2, a.a_gauss);
  } else if (b < a.nA + a.nM) {
    mlp_wgrad2_reduce_body(b - a.nA, a.nM, a.m_ws, a.m_splits, a.m_E, a.m_mh, a.m_dW2, a.m_dW1, a.m_db2, a.m_db1);
  } else if (b < a.nA + a.nM + a.nG) {
    const int g = b - a.nA - a.nM, per = (a.g_E + 63) / 64;
    const int job = g / per;
    gn_param_grad_body(g - job * per, a.g_part[job], a.g_dgamma[job], a.g_dbeta[job], a.g_B, a.g_E);
  } else {
    int c = b - a.nA - a.nM - a.nG;
    const int job = c >= a.c_blocks[0] ? 1 : 0;
    c -= job * a.c_blocks[0];
    colsum_rows_body(c, a.c_part[job], a.c_out[job], a.c_rows[job], a.c_N[job]);
  }
}
}  // namespace dpot

extern "C" int dpot_mlp_wgrad2_splitk(int T, int E, int mh) {
  static const int enabled = tune("panel", 1);
  if (!enabled || T <= 0 || E <= 0 || mh <= 0 || E % TN_W || mh % TN_W || T % TN_TOK) return 0;
  const long long tiles = 2ll * (E / TN_W) * (mh / TN_W);
  if (tiles > 256) return 0;                           // big layers fill the chip per launch: nothing to merge for
  const int nslab = T / TN_TOK;
  long long s = (256 + tiles / 2) / tiles;
  const long long smax = nslab / 4;
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  while (s > 1 && ((nslab + s - 1) / s) * (s - 1) >= nslab) --s;
  return (int)s;
}

extern "C" int64_t dpot_mlp_wgrad2_ws_elems(int E, int mh, int splitk) {
  const int64_t L = E > mh ? E : mh;
  return (int64_t)splitk * 2 * ((int64_t)E * mh + L);
}

extern "C" int dpot_mlp_wgrad2(const float* do2, const float* Hh, const float* xn2, const float* dHpre, int T, int E,
                               int mh, float* dW2, float* db2, float* dW1, float* db1, float* workspace, int splitk,
                               dpot_stream_t stream) {
  DPOT_REQUIRE(do2 && Hh && xn2 && dHpre && workspace, "mlp_wgrad2: null pointer");
  DPOT_REQUIRE((dW2 && db2 && dW1 && db1) || (!dW2 && !db2 && !dW1 && !db1), "mlp_wgrad2: outputs must be all given or all NULL");
  DPOT_REQUIRE(T > 0 && E % TN_W == 0 && mh % TN_W == 0 && T % TN_TOK == 0, "mlp_wgrad2: needs E, mh %% 128 == 0, T %% 32 == 0");
  DPOT_REQUIRE(aligned16(do2) && aligned16(Hh) && aligned16(xn2) && aligned16(dHpre) && aligned16(workspace),
               "mlp_wgrad2: operands must be 16-byte aligned");
  const int nslab = T / TN_TOK;
  DPOT_REQUIRE(splitk >= 1 && splitk <= nslab && 2ll * splitk <= 65535, "mlp_wgrad2: bad split factor");
  TnArgs p;
  p.A = do2; p.B = Hh; p.A2 = xn2; p.B2 = dHpre; p.batch1 = 1;
  p.lda = E; p.ldb = mh; p.sA = 0; p.sB = 0;
  p.N1 = E; p.N2 = mh; p.T = T; p.batch = 2;
  p.tiles1 = E / TN_W; p.tiles2 = mh / TN_W;
  p.splits = splitk;
  p.slabs_per_split = (nslab + splitk - 1) / splitk;
  DPOT_REQUIRE((long long)p.slabs_per_split * (splitk - 1) < nslab, "mlp_wgrad2: split factor leaves an empty split");
  p.ws = workspace;
  p.cs_of = 1; p.cs_of2 = 2; p.csL = E > mh ? E : mh;
  p.gauss = 0; p.splits12 = p.splits; p.slabs_per_split12 = p.slabs_per_split;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(p.tiles1 * p.tiles2), 1, (unsigned)(2 * splitk)), dim3(384), 0, s, p);
  int rc = check_launch("gemm_tn_kernel");
  if (rc || !dW2) return rc;            // dW2 == NULL: partials only, dpot_block_finalize reduces them
  int blocks = 2 * (E / 32) * (mh / 32);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mlp_wgrad2_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)workspace, splitk, E, mh, dW2,
                     dW1, db2, db1);
  return check_launch("mlp_wgrad2_reduce_kernel");
}

extern "C" int dpot_block_finalize(const float* afno_ws, int afno_splitk, int nb, int bs, float* dw1, float* db1, float* dw2,
                                   float* db2, const float* mlp_ws, int mlp_splitk, int E, int mh, float* dW2, float* dfb2,
                                   float* dW1, float* dfb1, const float* const* gn_parts, float* const* gn_dgammas,
                                   float* const* gn_dbetas, int gn_jobs, int B, int Egn, const float* const* cs_parts,
                                   float* const* cs_outs, const int* cs_rows, const int* cs_cols, int cs_jobs,
                                   dpot_stream_t stream) {
  FinalizeArgs a{};
  if (cs_jobs > 0) {
    DPOT_REQUIRE(cs_jobs <= 2 && cs_parts && cs_outs && cs_rows && cs_cols, "block_finalize: bad column-sum slice");
    for (int i = 0; i < cs_jobs; ++i) {
      DPOT_REQUIRE(cs_parts[i] && cs_outs[i] && cs_rows[i] > 0 && cs_cols[i] > 0, "block_finalize: bad column-sum job %d", i);
      a.c_part[i] = cs_parts[i]; a.c_out[i] = cs_outs[i]; a.c_rows[i] = cs_rows[i]; a.c_N[i] = cs_cols[i];
      a.c_blocks[i] = (cs_cols[i] + CS_COLS - 1) / CS_COLS;
    }
  }
  if (afno_ws) {
    DPOT_REQUIRE(afno_splitk >= 1 && nb > 0 && bs > 0 && dw1 && db1 && dw2 && db2, "block_finalize: bad AFNO slice");
    a.a_ws = afno_ws; a.a_splits = afno_splitk; a.a_nb = nb; a.a_bs = bs; a.a_gauss = tn_gauss(bs) ? tn_gauss_s12(bs, afno_splitk) : 0; a.a_dw1 = dw1; a.a_db1 = db1; a.a_dw2 = dw2; a.a_db2 = db2;
    long long blocks = (2ll * nb * bs * bs + 255) / 256;            // as dpot_afno_wgrad2
    a.nA = (int)(blocks > 4096 ? 4096 : blocks);
  }
  if (mlp_ws) {
    DPOT_REQUIRE(mlp_splitk >= 1 && E > 0 && mh > 0 && E % 32 == 0 && mh % 32 == 0 && dW2 && dfb2 && dW1 && dfb1,
                 "block_finalize: bad channel-MLP slice");
    a.m_ws = mlp_ws; a.m_splits = mlp_splitk; a.m_E = E; a.m_mh = mh; a.m_dW2 = dW2; a.m_dW1 = dW1; a.m_db2 = dfb2; a.m_db1 = dfb1;
    int blocks = 2 * (E / 32) * (mh / 32);                          // as dpot_mlp_wgrad2
    a.nM = blocks > 2048 ? 2048 : blocks;
  }
  if (gn_jobs > 0) {
    DPOT_REQUIRE(gn_jobs <= 2 && gn_parts && gn_dgammas && gn_dbetas && B > 0 && Egn > 0, "block_finalize: bad GroupNorm slice");
    for (int i = 0; i < gn_jobs; ++i) {
      DPOT_REQUIRE(gn_parts[i] && gn_dgammas[i] && gn_dbetas[i], "block_finalize: null pointer in GroupNorm job %d", i);
      a.g_part[i] = gn_parts[i]; a.g_dgamma[i] = gn_dgammas[i]; a.g_dbeta[i] = gn_dbetas[i];
    }
    a.g_jobs = gn_jobs; a.g_B = B; a.g_E = Egn; a.nG = gn_jobs * ((Egn + 63) / 64);
  }
  const int total = a.nA + a.nM + a.nG + a.c_blocks[0] + a.c_blocks[1];
  DPOT_REQUIRE(total > 0, "block_finalize: nothing to do");
  hipLaunchKernelGGL(block_finalize_kernel, dim3((unsigned)total), dim3(256), 0, as_stream(stream), a);
  return check_launch("block_finalize_kernel");
}
