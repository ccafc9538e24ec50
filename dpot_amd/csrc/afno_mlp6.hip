// afno_mlp6.hip - the AFNO mixer's block-diagonal complex 2-layer MLP (models/dpot.py:72-94) on the BF16 matrix cores at
// fp32 accuracy ("bf16x6", round 6):
//
//      Y1 = f(X Wa + ba)          f = act (forward)  or  (.) * act'(aux)  (backward data path, no biases)
//      Y2 = Y1 Wb + bb
//
// Same contract as afno_mlp.hip (dpot_afno_mlp2, weight layout 2).  Every operand is split x = x1 + x2 + x3 into three
// bf16 planes (exact to 2^-25 of |x|) and the six plane products whose weight is >= 2^-16 are accumulated in fp32 on
// v_mfma_f32_32x32x16_bf16: the matrix-core roof is 2.5 PF / 6 = 417 TFLOP/s of fp32-accurate work against 157 TFLOP/s of
// the native fp32 MFMA (afno_mlp.hip's three-product kernel reaches 0.59 .. 0.69 of that).  The static weights are split
// ONCE per optimiser step (dpot_afno_pack6), the activations once per row tile.
//
// Structure - everything TRANSPOSED, so that the hidden layer never leaves the registers:
//   * a wave owns 32 rows (tokens = kept modes of the spectrum) and ALL N = 2 bs output channels of one channel block:
//     Y^T[n, token] = sum_k W^T[n, k] X^T[k, token], the WEIGHT fragment is the A operand (32 channels x 16 k), the
//     activation the B operand (16 k x 32 tokens).  The accumulator of channel tile t then holds, in lane (token = l & 31,
//     h = l >> 5), register r = channel 32 t + 8 (r >> 2) + 4 h + (r & 3) - i.e. registers 8 j .. 8 j + 7 of tile t ARE the B
//     operand of the second layer for the 16 k-values {32 t + 16 j + 8 (i >> 2) + 4 h + (i & 3)}: the k order inside an MFMA
//     is free as long as both operands agree, and the weight packs are written in exactly that order (order 2 below).
//     No LDS round trip, no cross-lane traffic between the layers; bias / activation / saves act on 16-byte groups of 4
//     consecutive channels of a row.
//   * the first layer's B operand comes straight from global memory: lane (token, g) reads 64 contiguous bytes per 32-k
//     super-slab (k = 32 S + 16 g + 0 .. 15, prefetched one super-slab ahead) and uses them as two 8-k halves (order 1).
//   * the weight fragments of a 16-k sub-slab (N / 32 tiles x 3 planes x 1 KiB) travel global -> LDS by LDS-DMA into a ring
//     of three slots shared by the four waves of a workgroup (128 rows); one barrier per sub-slab; fragment reads are
//     lane-linear ds_read_b128 (conflict-free).  Per sub-slab and wave: N / 32 x 3 fragment reads feed N / 32 x 6 MFMAs.
//   * N = 192 (DPOT-L): accumulators 96 (hidden) + 96, 226-232 VGPRs, two workgroups per CU.  N = 256 (DPOT-Ti / -S / -M, opt-in
//     DPOT_TUNE mixer6=2): 128 + 128 accumulators, one workgroup per CU at one wave per SIMD (186 VGPRs + 128 AGPRs; the two-pass
//     form that fits 256 registers spills 70 of them and is no faster) - no gain over the fp32 kernels there: 144 / 288
//     workgroups of 128 rows on 256 CUs.
#include <type_traits>

#include "common.h"
#include "gemm_bf16p_common.h"

namespace dpot {

struct AfnoMlp6Args {
  const float* X;             // [M, ldx]
  const unsigned char* Wa;    // [nb][N/16 sub-slabs][N/32 tiles][3 planes][64 lanes][8 bf16], k order 1
  const unsigned char* Wb;    // same, k order 2
  const float* ba;            // [nb][N] or NULL
  const float* bb;
  const float* aux;           // mode 1: pre-activation of the forward [M, ldo]
  float* pre;                 // optional: X Wa + ba (mode 0) / act(aux) (mode 1)   [M, ldo]
  float* mid;                 // optional: Y1                                       [M, ldo]
  float* Y;                   // Y2                                                 [M, ldo]
  int ldx, ldo;
  int M, nb, act, mode;
};

// x = a + b + c, three bf16 planes (round to nearest even each; the remainders are exact in fp32)
__device__ __forceinline__ void split3(const float* v, bf16x8_t& a, bf16x8_t& b, bf16x8_t& c) {
  u32x4_t ua, ub, uc;
#ifdef M6_ABL_NOSPLIT   // (ablation builds, scripts/variant.sh afno_mlp6 NAME -DM6_ABL_...: NOSPLIT, DMA1, NOSTORE - results are wrong)
  for (int k = 0; k < 4; ++k) { ua[k] = __float_as_uint(v[k]); ub[k] = __float_as_uint(v[k + 4]); uc[k] = ua[k] ^ ub[k]; }
  a = __builtin_bit_cast(bf16x8_t, ua); b = __builtin_bit_cast(bf16x8_t, ub); c = __builtin_bit_cast(bf16x8_t, uc);
  return;
#endif
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float x0 = v[2 * k], x1 = v[2 * k + 1];
    unsigned u = pack2(x0, x1);
    ua[k] = u;
    x0 -= bf_lo(u);
    x1 -= bf_hi(u);
    u = pack2(x0, x1);
    ub[k] = u;
    x0 -= bf_lo(u);
    x1 -= bf_hi(u);
    uc[k] = pack2(x0, x1);
  }
  a = __builtin_bit_cast(bf16x8_t, ua);
  b = __builtin_bit_cast(bf16x8_t, ub);
  c = __builtin_bit_cast(bf16x8_t, uc);
}

// the six plane products of one 32 x 32 x 16 step, smallest terms first
__device__ __forceinline__ void mma6(f32x16& acc, const bf16x8_t* w, const bf16x8_t* x) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], acc, 0, 0, 0);
}

template <int B, int E, class F>
__device__ __forceinline__ void sfor6(F&& f) {   // static for: f(integral_constant<int, i>) for i in [B, E)
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    sfor6<B + 1, E>(f);
  }
}

// the MFMAs of one ring step: NTL tiles x 6 plane products.  The weight fragments of tile m + 1 are read from LDS while the
// six MFMAs of tile m run (two register sets).  The reads and their waits are INLINE ASSEMBLY with counted lgkmcnt: with an
// LDS-DMA in flight the compiler guards every use of a ds_read result with lgkmcnt(0) - each tile then pays the full LDS
// latency of the reads just issued for the next one (measured: 2.4x the MFMA time per step).  The wait statement names the
// fragment registers as in/out operands, so their consumers cannot move above it; the scheduling barriers keep the order.
template <int OFF>
__device__ __forceinline__ void lds_rd3(bf16x8_t* w, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[0]) : "v"(addr), "n"(OFF));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[1]) : "v"(addr), "n"(OFF + 1024));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[2]) : "v"(addr), "n"(OFF + 2048));
}
template <int CNT>
__device__ __forceinline__ void lds_wait3(bf16x8_t* w) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]) : "n"(CNT));
}
template <int NTL>
__device__ __forceinline__ void tiles(unsigned addr, f32x16* acc, const bf16x8_t* x) {
  bf16x8_t w[2][3];
  lds_rd3<0>(w[0], addr);
  sfor6<0, NTL>([&](auto MM) __attribute__((always_inline)) {
    constexpr int m = decltype(MM)::value;
    if constexpr (m + 1 < NTL) {
      lds_rd3<(m + 1) * 3072>(w[(m + 1) & 1], addr);
      lds_wait3<3>(w[m & 1]);
    } else {
      lds_wait3<0>(w[m & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    mma6(acc[m], w[m & 1], x);
    __builtin_amdgcn_sched_barrier(0);
  });
}


// store of one tile (32 tokens x 32 channels) held in the accumulator layout - lane (token, h), value 4 G + k = channel
// 8 G + 4 h + k - as FULL 128-byte lines: through a per-wave LDS slab [32 rows][36 floats] (4 ds_write_b128, then every lane
// reads 16 bytes of row 8 it + (l >> 3)), 8 rows x 128 B per store instruction.  (Stored straight from the accumulator layout
// an instruction writes 32 rows x 32 bytes: the same bytes in four times the requests - measured 24 % / 40 % of the forward /
// backward launch at DPOT-L.)  rowbase = global pointer of (first row of the wave's tile, this tile's first channel); rows
// = valid rows of the wave's tile (32, fewer in the last row tile)
constexpr int M6_STG_LD = 36;
__device__ __forceinline__ void stage_store(float* stg, const float* v, float* rowbase, long long ld, int rows, int lane) {
  const int tk = lane & 31, h = lane >> 5;
#pragma unroll
  for (int G = 0; G < 4; ++G)
    *reinterpret_cast<float4*>(stg + tk * M6_STG_LD + 8 * G + 4 * h) = make_float4(v[4 * G], v[4 * G + 1], v[4 * G + 2], v[4 * G + 3]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int rr = lane >> 3, cc = (lane & 7) * 4;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = 8 * it + rr;
    const float4 q = *reinterpret_cast<const float4*>(stg + row * M6_STG_LD + cc);
    if (row < rows) *reinterpret_cast<float4*>(rowbase + (long long)row * ld + cc) = q;
  }
  __builtin_amdgcn_wave_barrier();
}

constexpr int M6_RING = 3;

// one PAIR of values -> one dword of each of the three planes (the unit of split work that goes into an MFMA shadow)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& a, unsigned& b, unsigned& c) {
  a = pack2(x0, x1);
  x0 -= bf_lo(a);
  x1 -= bf_hi(a);
  b = pack2(x0, x1);
  x0 -= bf_lo(b);
  x1 -= bf_hi(b);
  c = pack2(x0, x1);
}
struct Planes {           // 8 k-values of one lane as three bf16 planes, built one dword (pair) at a time
  u32x4_t u[3];
};
__device__ __forceinline__ void planes_set(Planes& q, int k, float x0, float x1) {
  unsigned a, b, c;
  split_pair(x0, x1, a, b, c);
  q.u[0][k] = a; q.u[1][k] = b; q.u[2][k] = c;
}

// the six plane products of tile job (fragments w, planes x) with the FILLER work of this job interleaved into the MFMA
// shadows: one MFMA, then up to three VALU instructions (the split of the NEXT sub-slab's operand) and, once, the LDS-DMA
// piece of a later ring step - left at the head of the step these serialise with the MFMAs (a wave issues in order)
__device__ __forceinline__ void mma6p(f32x16& acc, const bf16x8_t* w, const Planes& xq) {
  bf16x8_t x[3] = {__builtin_bit_cast(bf16x8_t, xq.u[0]), __builtin_bit_cast(bf16x8_t, xq.u[1]),
                   __builtin_bit_cast(bf16x8_t, xq.u[2])};
  mma6(acc, w, x);
}
__device__ __forceinline__ void interleave6() {
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
  __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // the DMA piece (VMEM)
  __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
  }
}

// NT = N / 32 channel tiles, PASSES = passes of the second layer (NT / PASSES output tiles each), ACTK = compile-time
// activation (-1: run-time switch), WPE = waves per SIMD the register budget is cut for, MODE = 0 forward / 1 backward data
//
// Ring steps: 2 Q of them (Q = N / 16), each NT tile jobs = 6 NT MFMAs per wave behind one barrier, each fed by NT x 3 KiB of
// fragments in a ring slot: first layer - step i = sub-slab i, all NT tiles; second layer - step Q + r = the PASSES
// consecutive sub-slabs r PASSES .. of the job sequence s = pass Q + q, NT / PASSES tiles each.
template <int NT, int PASSES, int ACTK, int WPE, int MODE>
__global__ __launch_bounds__(256, WPE) void afno_mlp6_kernel(const AfnoMlp6Args p) {
  constexpr int N = 32 * NT;
  constexpr int Q = 2 * NT;                       // 16-k sub-slabs per layer
  constexpr int NS = NT;                          // 32-k super-slabs of the first layer
  constexpr int TP = NT / PASSES;                 // output tiles per pass of the second layer
  constexpr int SLABB = NT * 3 * 1024;            // bytes of one ring slot
  constexpr int CH = NT * 3;                      // 1 KiB chunks per ring step
  constexpr int PER = (CH + 3) / 4;               // DMA instructions per wave and step
  constexpr int T = 2 * Q;                        // ring steps
  constexpr int NJ2 = PASSES * Q;                 // sub-slab jobs of the second layer
  static_assert(NT % PASSES == 0 && Q % PASSES == 0, "passes must divide the tiles and the sub-slabs");
  static_assert(PER <= NT, "one DMA piece per tile job");
  __shared__ __attribute__((aligned(16))) unsigned char lds[M6_RING * SLABB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = blockIdx.y;
  const int act = ACTK >= 0 ? ACTK : p.act;
  const int tok = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const bool valid = tok < p.M;
  const int tokc = valid ? tok : p.M - 1;
  const int g = lane >> 5;
  // LDS byte address of this lane's chunk of a fragment (ds_read_b128 operand)
  const unsigned lds_base =
      (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)lds) + (unsigned)lane * 16u;
  const unsigned char* wa = p.Wa + (long long)blk * Q * SLABB;
  const unsigned char* wb = p.Wb + (long long)blk * Q * SLABB;

  // DMA piece n (of PER) of ring step i into its slot: chunk c = wave + 4 n (a wave past the end re-copies its last chunk:
  // uniform instruction count)
  auto dma_piece = [&](auto II, auto NN) __attribute__((always_inline)) {
    constexpr int i = decltype(II)::value, n = decltype(NN)::value;
    if constexpr (i < T && n < PER) {
#ifdef M6_ABL_DMA1
      if (n > 0) return;
#endif
      int c = wave + 4 * n;
      if (c >= CH) c -= 4;
      const unsigned char* src;
      if constexpr (i < Q) {
        src = wa + (long long)i * SLABB + c * 1024;
      } else {
        // chunk c of the step = job (c / (3 TP)) of the step, chunk (c % (3 TP)) of that job's TP tiles
        const int jj = c / (3 * TP), within = c - jj * (3 * TP);
        const int sj = (i - Q) * PASSES + jj;               // job s = pass Q + q
        const int pass = sj / Q, q = sj - pass * Q;
        src = wb + (long long)q * SLABB + (pass * TP * 3 + within) * 1024;
      }
      bglds16(src + lane * 16, lds + (i % M6_RING) * SLABB + c * 1024);
    }
  };

  f32x16 acc1[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;

  // ---- first layer: acc1^T = Wa^T X^T ------------------------------------------------------------------------------------
  // X of super-slab S (16 k of this lane's row: k = 32 S + 16 g + 0 .. 15) sits in xbuf[S & 1], loaded two super-slabs ahead;
  // its planes xq[S & 1][J] (J = the 8-k half = the sub-slab) are split during super-slab S - 1, pair by pair, in the shadows
  // of that super-slab's MFMAs
  const float4* xp = reinterpret_cast<const float4*>(p.X + (long long)tokc * p.ldx + blk * N) + 4 * g;
  float4 xbuf[2][4];
  Planes xq[2][2];
#pragma unroll
  for (int k = 0; k < 4; ++k) xbuf[0][k] = xp[k];
  if constexpr (NS > 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) xbuf[1][k] = xp[8 + k];
  }
  sfor6<0, PER>([&](auto NN) __attribute__((always_inline)) { dma_piece(std::integral_constant<int, 0>{}, NN); });
  sfor6<0, PER>([&](auto NN) __attribute__((always_inline)) { dma_piece(std::integral_constant<int, 1>{}, NN); });
  {
    const float* x0 = reinterpret_cast<const float*>(xbuf[0]);
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
      for (int k = 0; k < 4; ++k) planes_set(xq[0][J], k, x0[8 * J + 2 * k], x0[8 * J + 2 * k + 1]);
  }

  sfor6<0, Q>([&](auto II) __attribute__((always_inline)) {
    constexpr int i = decltype(II)::value;
    constexpr int S = i >> 1, J = i & 1;
    // arrived: the DMA of step i.  Issued after it: [step i - 1] the X loads of super-slab S' + 2 (J' == 0) and DMA(i + 1)
    constexpr bool xprev = i >= 1 && ((i - 1) & 1) == 0 && ((i - 1) >> 1) + 2 < NS;
    constexpr int VMW = PER + (xprev ? 4 : 0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt((VMW & 15) | 0x70 | ((VMW >> 4) << 14));   // vmcnt(VMW) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // (X of super-slab S + 1 - split by this step's fillers - is older than everything the wait left in flight: arrived;
    // the fillers sit between scheduling barriers below the wait, so no use of it can move above)
    if constexpr (J == 0 && S + 2 < NS) {
      // X of super-slab S + 2 into the buffer of super-slab S (free: its planes were split during super-slab S - 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) xbuf[S & 1][k] = xp[8 * (S + 2) + k];
    }
    const float* xnext = reinterpret_cast<const float*>(xbuf[(S + 1) & 1]);
    const unsigned addr = lds_base + (i % M6_RING) * SLABB;
    bf16x8_t w[2][3];
    lds_rd3<0>(w[0], addr);
    sfor6<0, NT>([&](auto MM) __attribute__((always_inline)) {
      constexpr int m = decltype(MM)::value;
      if constexpr (m + 1 < NT) {
        lds_rd3<(m + 1) * 3072>(w[(m + 1) & 1], addr);
        lds_wait3<3>(w[m & 1]);
      } else {
        lds_wait3<0>(w[m & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      dma_piece(std::integral_constant<int, i + 2>{}, MM);
      if constexpr (S + 1 < NS && m < 4)          // pair m of half J of the next super-slab
        planes_set(xq[(S + 1) & 1][J], m, xnext[8 * J + 2 * m], xnext[8 * J + 2 * m + 1]);
      mma6p(acc1[m], w[m & 1], xq[S & 1][J]);
      interleave6();
      __builtin_amdgcn_sched_barrier(0);
    });
  });

  // the first two second-layer steps are in flight; wait for them BEFORE the stores below join the queue (loads complete in
  // order among themselves, not with respect to stores)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0070);
  asm volatile("" ::: "memory");

  // ---- between the layers: bias, saves, activation (in place: acc1 becomes the hidden layer) --------------------------------
  // per tile: the 16-byte loads of the NEXT tile (bias / aux) are issued before the current tile's arithmetic and stores
  const long long orow = (long long)tokc * p.ldo + blk * N + 4 * g;
#ifdef M6_ABL_NOSTORE
  const bool st_ok = p.M < 0;
#else
  const bool st_ok = true;
#endif
  const bool w_pre = p.pre != nullptr && st_ok, w_mid = p.mid != nullptr && st_ok;
  // staged stores: this wave's slab in the ring slot nobody uses right now (the one the last first-layer step read - every
  // wave is past it after the barrier below); global base of the wave's tile rows
  const int row0 = blockIdx.x * 128 + wave * 32;
  const int rows_ok = p.M - row0 < 32 ? p.M - row0 : 32;          // (<= 0: nothing to store)
  const long long obase = (long long)row0 * p.ldo + blk * N;
  float* stg = reinterpret_cast<float*>(lds + ((Q + 2) % M6_RING) * SLABB) + wave * (32 * M6_STG_LD);
  static_assert(4 * 32 * M6_STG_LD * 4 <= SLABB, "the staging slabs of four waves fit a ring slot");
  __builtin_amdgcn_s_barrier();
  {
    const float* lsrc = MODE == 0 ? p.ba + blk * N + 4 * g : p.aux + orow;     // 4 x float4 per tile, 8 floats apart
    const bool has_l = MODE == 1 || p.ba != nullptr;
    float4 ln[4];
#pragma unroll
    for (int G = 0; G < 4; ++G) ln[G] = has_l ? *reinterpret_cast<const float4*>(lsrc + 8 * G) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float4 lc[4];
#pragma unroll
      for (int G = 0; G < 4; ++G) lc[G] = ln[G];
      if (t + 1 < NT) {
#pragma unroll
        for (int G = 0; G < 4; ++G)
          ln[G] = has_l ? *reinterpret_cast<const float4*>(lsrc + 32 * (t + 1) + 8 * G) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float f[16];
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        const float lv[4] = {lc[G].x, lc[G].y, lc[G].z, lc[G].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v = acc1[t][4 * G + k];
          if constexpr (MODE == 0) {
            v += lv[k];
            f[4 * G + k] = v;                                   // the pre-activation
            v = act == DPOT_ACT_GELU ? gelu_fwd(v) : act_fwd(act, v);
          } else {
            float d;
            if (act == DPOT_ACT_GELU) {
              gelu_val_der(lv[k], f[4 * G + k], d);             // f = act(aux): the forward's hidden layer, re-derived
            } else {
              d = act_bwd(act, lv[k]);
              f[4 * G + k] = act_fwd(act, lv[k]);
            }
            v *= d;
          }
          acc1[t][4 * G + k] = v;
        }
      }
      if (w_pre) stage_store(stg, f, p.pre + obase + 32 * t, p.ldo, rows_ok, lane);
      if (w_mid) {
        float hv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) hv[k] = acc1[t][k];
        stage_store(stg, hv, p.mid + obase + 32 * t, p.ldo, rows_ok, lane);
      }
    }
  }

  // ---- second layer: acc2^T = Wb^T H^T --------------------------------------------------------------------------------------
  // sub-slab jobs s = pass Q + q (TP output tiles each); the planes of job s + 1 - registers 8 j .. 8 j + 7 of hidden tile t,
  // q = 2 t + j - are split in the MFMA shadows of job s (two plane sets)
  Planes hq[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) planes_set(hq[0], k, acc1[0][2 * k], acc1[0][2 * k + 1]);
  f32x16 acc2[TP];
  sfor6<0, Q>([&](auto RR) __attribute__((always_inline)) {
    constexpr int r = decltype(RR)::value;       // ring step Q + r
    constexpr int i = Q + r;
    asm volatile("" ::: "memory");
    if constexpr (r >= 2) {                       // (steps 0 and 1: waited for above)
      constexpr int VMW = i + 1 < T ? PER : 0;
      __builtin_amdgcn_s_waitcnt((VMW & 15) | 0x70 | ((VMW >> 4) << 14));
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    sfor6<0, PASSES>([&](auto JJ) __attribute__((always_inline)) {
      constexpr int jj = decltype(JJ)::value;
      constexpr int sj = r * PASSES + jj;         // job
      constexpr int pass = sj / Q, q = sj % Q;
      if constexpr (q == 0) {
#pragma unroll
        for (int m = 0; m < TP; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc2[m][e] = 0.f;
      }
      constexpr int sn = sj + 1;                  // the job whose planes are split here
      constexpr int tn = (sn % Q) >> 1, jn = (sn % Q) & 1;
      const unsigned addr = lds_base + (i % M6_RING) * SLABB + jj * (TP * 3072);
      bf16x8_t w[2][3];
      lds_rd3<0>(w[0], addr);
      sfor6<0, TP>([&](auto MM) __attribute__((always_inline)) {
        constexpr int m = decltype(MM)::value;
        if constexpr (m + 1 < TP) {
          lds_rd3<(m + 1) * 3072>(w[(m + 1) & 1], addr);
          lds_wait3<3>(w[m & 1]);
        } else {
          lds_wait3<0>(w[m & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        dma_piece(std::integral_constant<int, i + 2>{}, std::integral_constant<int, jj * TP + m>{});
        if constexpr (sn < NJ2) {
          // the four pairs of the next job's planes over this job's TP tile slots
          constexpr int k0 = m * 4 / TP, k1 = (m + 1) * 4 / TP;
#pragma unroll
          for (int k = k0; k < k1; ++k)
            planes_set(hq[sn & 1], k, acc1[tn][8 * jn + 2 * k], acc1[tn][8 * jn + 2 * k + 1]);
        }
        mma6p(acc2[m], w[m & 1], hq[sj & 1]);
        interleave6();
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (q == Q - 1) {
        // output tiles pass * TP .. + TP - 1 (the bias of the next tile is fetched ahead); staged through the ring slot that
        // this step read (no DMA targets it until the next step's barrier)
        float* stg2 = reinterpret_cast<float*>(lds + (i % M6_RING) * SLABB) + wave * (32 * M6_STG_LD);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): own fragment reads done
        __builtin_amdgcn_s_barrier();                        // every wave is past the slot's last fragment read
        asm volatile("" ::: "memory");
        const float* bsrc = p.bb ? p.bb + blk * N + 4 * g + 32 * pass * TP : nullptr;
        float4 bn[4];
#pragma unroll
        for (int G = 0; G < 4; ++G) bn[G] = bsrc ? *reinterpret_cast<const float4*>(bsrc + 8 * G) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int m = 0; m < TP; ++m) {
          float4 bc[4];
#pragma unroll
          for (int G = 0; G < 4; ++G) bc[G] = bn[G];
          if (m + 1 < TP) {
#pragma unroll
            for (int G = 0; G < 4; ++G)
              bn[G] = bsrc ? *reinterpret_cast<const float4*>(bsrc + 32 * (m + 1) + 8 * G) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
          float yv[16];
#pragma unroll
          for (int G = 0; G < 4; ++G) {
            yv[4 * G] = acc2[m][4 * G] + bc[G].x; yv[4 * G + 1] = acc2[m][4 * G + 1] + bc[G].y;
            yv[4 * G + 2] = acc2[m][4 * G + 2] + bc[G].z; yv[4 * G + 3] = acc2[m][4 * G + 3] + bc[G].w;
          }
          if (st_ok)
            stage_store(stg2, yv, p.Y + obase + 32 * (pass * TP + m), p.ldo, rows_ok, lane);
        }
      }
    });
  });
}

// wbig [nitems * nb][N][N] (W[k][n]) -> the bf16x6 fragment packs of afno_mlp6_kernel.  Items alternate (first-layer weight,
// second-layer weight): fwd6 = the forward operand (W; k order 1 for a first-layer weight, 2 for a second-layer weight),
// bwd6 = the backward-data operand (W^T; the layers swap places there: order 2 for a first-layer weight, 1 for a second).
//   order 1 (k from the global loads of X):  k(q, g, i) = 32 (q >> 1) + 16 g + 8 (q & 1) + i
//   order 2 (k from the accumulator layout): k(q, g, i) = 32 (q >> 1) + 16 (q & 1) + 8 (i >> 2) + 4 g + (i & 3)
__global__ __launch_bounds__(256) void afno_pack6_kernel(const float* __restrict__ wbig, uint4* __restrict__ fwd6,
                                                         uint4* __restrict__ bwd6, int N, int nb, long long total) {
  const int NT = N >> 5, Q = N >> 4;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int l = (int)(idx & 63);
    long long r = idx >> 6;
    const int m = (int)(r % NT); r /= NT;
    const int q = (int)(r % Q); r /= Q;
    const long long mat = r;
    const int second = (int)((mat / nb) & 1);
    const float* W = wbig + mat * N * N;
    const int n = 32 * m + (l & 31), g = l >> 5;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
      uint4* dst = dir == 0 ? fwd6 : bwd6;
      if (!dst) continue;
      const int order = (second ^ dir) ? 2 : 1;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = order == 1 ? 32 * (q >> 1) + 16 * g + 8 * (q & 1) + i
                                 : 32 * (q >> 1) + 16 * (q & 1) + 8 * (i >> 2) + 4 * g + (i & 3);
        v[i] = dir == 0 ? W[(long long)k * N + n] : W[(long long)n * N + k];
      }
      bf16x8_t a, b, c;
      split3(v, a, b, c);
      uint4* o = dst + ((mat * Q + q) * NT + m) * 192 + l;       // [mat][q][m][plane][lane]
      o[0] = __builtin_bit_cast(uint4, a);
      o[64] = __builtin_bit_cast(uint4, b);
      o[128] = __builtin_bit_cast(uint4, c);
    }
  }
}

template <int NT, int PASSES, int WPE>
static int launch6(const AfnoMlp6Args& p, hipStream_t s) {
  const dim3 grid((unsigned)((p.M + 127) / 128), (unsigned)p.nb), blk(256);
  if (p.act == DPOT_ACT_GELU) {
    if (p.mode == 0) hipLaunchKernelGGL((afno_mlp6_kernel<NT, PASSES, DPOT_ACT_GELU, WPE, 0>), grid, blk, 0, s, p);
    else hipLaunchKernelGGL((afno_mlp6_kernel<NT, PASSES, DPOT_ACT_GELU, WPE, 1>), grid, blk, 0, s, p);
  } else {
    if (p.mode == 0) hipLaunchKernelGGL((afno_mlp6_kernel<NT, PASSES, -1, WPE, 0>), grid, blk, 0, s, p);
    else hipLaunchKernelGGL((afno_mlp6_kernel<NT, PASSES, -1, WPE, 1>), grid, blk, 0, s, p);
  }
  return check_launch("afno_mlp6_kernel");
}

}  // namespace dpot

using namespace dpot;

extern "C" int dpot_afno_mlp6_supported(int nb, int bs) { return nb > 0 && nb <= 65535 && (bs == 128 || bs == 96) ? 1 : 0; }

extern "C" int64_t dpot_afno_pack6_elems(int nb, int bs) { return (int64_t)nb * 4 * bs * bs * 3; }

extern "C" int dpot_afno_pack6(const float* wbig, void* fwd6, void* bwd6, int nitems, int nb, int bs,
                               dpot_stream_t stream) {
  DPOT_REQUIRE(wbig && (fwd6 || bwd6) && nitems > 0 && dpot_afno_mlp6_supported(nb, bs), "afno_pack6: bad argument");
  DPOT_REQUIRE(aligned16(wbig) && aligned16(fwd6) && aligned16(bwd6), "afno_pack6: pointers must be 16-byte aligned");
  const int N = 2 * bs;
  const long long total = (long long)nitems * nb * (N / 16) * (N / 32) * 64;
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(afno_pack6_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), wbig,
                     reinterpret_cast<uint4*>(fwd6), reinterpret_cast<uint4*>(bwd6), N, nb, total);
  return check_launch("afno_pack6_kernel");
}

extern "C" int dpot_afno_mlp6(const float* X, const void* Wa6, const float* ba, const void* Wb6, const float* bb,
                              const float* aux, float* pre, float* mid, float* Y, int M, int nb, int bs, int ldx, int ldo,
                              int act, int mode, dpot_stream_t stream) {
  DPOT_REQUIRE(X && Wa6 && Wb6 && Y && M > 0, "afno_mlp6: bad argument");
  DPOT_REQUIRE(dpot_afno_mlp6_supported(nb, bs), "afno_mlp6: unsupported block size bs=%d (96 or 128)", bs);
  DPOT_REQUIRE(mode == 0 || (mode == 1 && aux != nullptr), "afno_mlp6: mode 1 (backward) needs aux");
  const int N = 2 * bs;
  DPOT_REQUIRE(ldx >= nb * N && ldo >= nb * N && ldx % 4 == 0 && ldo % 4 == 0, "afno_mlp6: bad leading dimension");
  DPOT_REQUIRE(aligned16(X) && aligned16(Wa6) && aligned16(Wb6) && aligned16(Y) && aligned16(ba) && aligned16(bb) &&
                   aligned16(aux) && aligned16(pre) && aligned16(mid),
               "afno_mlp6: pointers must be 16-byte aligned");
  AfnoMlp6Args p;
  p.X = X; p.Wa = static_cast<const unsigned char*>(Wa6); p.Wb = static_cast<const unsigned char*>(Wb6);
  p.ba = ba; p.bb = bb; p.aux = aux; p.pre = pre; p.mid = mid; p.Y = Y;
  p.ldx = ldx; p.ldo = ldo; p.M = M; p.nb = nb; p.act = act; p.mode = mode;
  hipStream_t s = as_stream(stream);
  if (bs == 128) return launch6<8, 1, 1>(p, s);
  return launch6<6, 1, 2>(p, s);
}
