// gemm_bf16bt.hip - the "big tile" bf16 GEMM of round 6: workgroup tile 256 x 256 (or 256 x 192), FOUR waves, each owning a
// 128 x 128 (128 x 96) block of the output = 4 x 4 (4 x 3) accumulators of v_mfma_f32_32x32x16_bf16 - 256 (192) accumulator
// registers, i.e. one wave per SIMD on the unified 512-entry register file.
//
// Why (profiles/r06_gemm_yardstick.txt, r06_yardstick_kernel_names.txt): on the same box hipBLASLt's tuned bf16 GEMM - timed as a
// measurement-only yardstick, never on the product path - reaches 1.17-1.33 PFLOP/s on the channel-MLP products of DPOT-M / -L
// where the 128 x 256 kernels of csrc/gemm_bf16p.hip reach 0.75-1.1, and the kernels it picks are 256 x 256 x 64 macro tiles on four
// waves of 128 x 128.  The arithmetic behind that: per 32-k slab a 128 x 256 tile moves (128 + 256) x 64 B = 24 KiB of operands
// from L2 into the CU for 2.1 MFLOP (85 FLOP/B), a 256 x 256 tile 32 KiB for 4.2 MFLOP (128 FLOP/B): a third less operand
// traffic per FLOP on the L2 -> CU path, and on a part whose matrix clocks are POWER-limited under this load (a register-only
// MFMA loop sustains 1.4-1.8 of the nominal 2.5 PFLOP/s) every byte not moved is clock for the matrix pipes.
//
// Structure (same packed operands - fragment-block-major 1 KiB blocks - and the same fragment epilogues as gemm_bf16p.hip):
//   * a 32-k slab = 16 A pieces + 2 COLT W pieces of 1 KiB, ALL through LDS-DMA (global_load ... lds), 8 (7) pieces per wave
//     and slab; ring of four slabs (128 / 112 KiB of the CU's 160 KiB), three slabs in flight, one counted vmcnt + ONE barrier
//     per slab, loads issued branch-free (past the end the last slab is re-loaded into a slot nobody reads);
//   * there is no second wave on a SIMD to hide latency behind: the fragments of slab g + 1 (8 + 2 x COLT / 2 ds_read_b128 per
//     wave: 16 KiB) are fetched into a SECOND register set under the 32 (24) MFMAs of slab g;
//   * LDS traffic per slab and CU: 32 KiB of DMA writes + 64 KiB of fragment reads against 1024 cycles of matrix-pipe time
//     (128 B/clk LDS peak -> 75 % headroom);
//   * epilogue: the wave's 16 (12) accumulators one after the other through ONE copy of the fragment epilogue (a wave-uniform
//     switch moves accumulator f into the common register set).
// Used for the launches whose 256-row tile grid fills the chip in whole rounds (dpot_gemm_bf16p picks: bf16p_plan); bit-identical
// to the 128-row kernels (same k order per output element, same epilogue code) - tests/test_gpu_ops.py::test_bf16bt_*.
#include "gemm_bf16p_common.h"

namespace dpot {

constexpr int BT_ROWT = 8;                 // 32-row tiles per workgroup (256 rows)
constexpr int BT_P = 3;                    // slabs of look-ahead
constexpr int BT_RING = BT_P + 1;          // slab slots in LDS

template <int B, int E, class F>
__device__ __forceinline__ void bt_sfor(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    bt_sfor<B + 1, E>(f);
  }
}

template <int COLT>
__global__ __launch_bounds__(256, 1) void gemm_bf16bt_kernel(const Bf16pArgs p) {
  static_assert(COLT == 8 || COLT == 6, "256 x 256 or 256 x 192 tiles");
  constexpr int CW = COLT / 2;                                   // 32-column tiles per wave
  constexpr int NPIECE = 2 * BT_ROWT + 2 * COLT;                 // 1 KiB pieces per slab (32 / 28)
  constexpr int NPW = NPIECE / 4;                                // pieces per wave and slab (8 / 7)
  constexpr int SLABB = NPIECE * 1024;
  static_assert(NPIECE % 4 == 0, "pieces must divide among the four waves");
  __shared__ __attribute__((aligned(16))) unsigned char lds[BT_RING * SLABB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int bid0 = blockIdx.x, zs = blockIdx.y;
  const int ks16 = p.K >> 4;
  const int slab0 = zs * p.slabs_per_split;
  int nslab = (p.K >> 5) - slab0;
  nslab = nslab < p.slabs_per_split ? nslab : p.slabs_per_split;

  int tm, tn;
  if (p.super_r > 0) {
    // L2-aware rasterisation: the 32 tiles an XCD runs at one time form a super_r x super_c block of the tile grid, and an XCD
    // walks the super-blocks of one super-row before the next (workgroup b runs on XCD b & 7)
    const int xcd = bid0 & 7, slot = bid0 >> 3;
    const int sb = slot >> 5, in = slot & 31;
    const int srows = p.tilesM / p.super_r, scols = p.tilesN / p.super_c;
    const int gsb = sb * 8 + xcd;
    if (gsb >= srows * scols) return;
    const int srow = gsb % srows, scol = gsb / srows;
    tm = srow * p.super_r + in % p.super_r;
    tn = scol * p.super_c + in / p.super_r;
  } else {
    // XCD-contiguous order: XCD x runs the tiles [x n / 8, (x + 1) n / 8) of the column-major (super_c == 0) or row-major
    // (super_c == 1) enumeration
    const int ntiles = p.tilesM * p.tilesN;
    const int xcd = bid0 & 7, slot = bid0 >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    if (p.super_c == 1) {
      tm = tile / p.tilesN;
      tn = tile - tm * p.tilesN;
    } else {
      tn = tile / p.tilesM;
      tm = tile - tn * p.tilesM;
    }
  }
  const int rt0 = tm * BT_ROWT, ct0 = tn * COLT;
  const int mtiles = (p.M + 31) >> 5;

  // this wave's DMA pieces: q = wave + 4 n; q < 16: A piece (row tile q >> 1, k-half q & 1), else W piece (column tile
  // (q - 16) >> 1, k-half q & 1).  Source = block (tile, 2 slab + k-half) of the packed operand, lane l its bytes 16 l .. + 15
  const unsigned short* src[NPW];
  int dst[NPW];
#pragma unroll
  for (int n = 0; n < NPW; ++n) {
    const int q = wave + 4 * n;
    if (q < 2 * BT_ROWT) {
      int rt = rt0 + (q >> 1);
      rt = rt < mtiles ? rt : mtiles - 1;                        // rows past the matrix: a valid address, results discarded
      src[n] = p.A + ((long long)rt * ks16 + 2 * slab0 + (q & 1)) * 512 + lane * 8;
    } else {
      const int qq = q - 2 * BT_ROWT;
      src[n] = p.W + ((long long)(ct0 + (qq >> 1)) * ks16 + 2 * slab0 + (qq & 1)) * 512 + lane * 8;
    }
    dst[n] = q * 1024;
  }

  f32x16 acc[4][CW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < CW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8_t fa[2][4][2], fw[2][CW][2];

  const int last = nslab - 1;
  auto issue = [&](int t, auto S) __attribute__((always_inline)) {
    constexpr int s = decltype(S)::value;
    t = t < last ? t : last;
#pragma unroll
    for (int n = 0; n < NPW; ++n) bglds16(src[n] + (long long)t * 1024, lds + s * SLABB + dst[n]);
  };
  auto frags = [&](auto S, auto SET) __attribute__((always_inline)) {
    constexpr int s = decltype(S)::value, st = decltype(SET)::value;
    const unsigned char* ba = lds + s * SLABB + (8 * wr) * 1024 + lane * 16;
    const unsigned char* bw = lds + s * SLABB + (2 * BT_ROWT + 2 * CW * wc) * 1024 + lane * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[st][i][ks] = *reinterpret_cast<const bf16x8_t*>(ba + (2 * i + ks) * 1024);
#pragma unroll
      for (int j = 0; j < CW; ++j) fw[st][j][ks] = *reinterpret_cast<const bf16x8_t*>(bw + (2 * j + ks) * 1024);
    }
  };
  auto mm = [&](auto SET) __attribute__((always_inline)) {
    constexpr int st = decltype(SET)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < CW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][i][ks], fw[st][j][ks], acc[i][j], 0, 0, 0);
  };
  // s_waitcnt vmcnt(V) lgkmcnt(0) as the builtin: simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 0 << 8 | vmcnt[5:4] << 14
  constexpr int VMW = NPW * (BT_P - 2);                          // slab g + 1 has landed when <= VMW operations are outstanding
  constexpr int WAITC = (VMW & 15) | 0x70 | ((VMW >> 4) << 14);
  constexpr int VM0 = NPW * (BT_P - 1);                          // prologue: slab 0 has landed
  constexpr int WAIT0 = (VM0 & 15) | 0x70 | ((VM0 >> 4) << 14);
  // step g: slab g + 1 (slot S) complete -> barrier -> loads of slab g + P into the slot of slab g - 1 (SN), fragments of slab
  // g + 1 into the other register set, MFMAs of slab g from set SET
  auto step = [&](int g, auto S, auto SN, auto SET) __attribute__((always_inline)) {
    constexpr int st = decltype(SET)::value;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(WAITC);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(g + BT_P, SN);
    frags(S, std::integral_constant<int, 1 - st>{});
    mm(SET);
  };

  bt_sfor<0, BT_P>([&](auto S) __attribute__((always_inline)) { issue(decltype(S)::value, S); });
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(WAIT0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  int g = 0;
#pragma unroll 1
  for (; g + BT_RING <= nslab; g += BT_RING)
    bt_sfor<0, BT_RING>([&](auto V) __attribute__((always_inline)) {
      constexpr int v = decltype(V)::value;                      // g is a multiple of the ring depth: slab g + v sits in slot v
      step(g + v, std::integral_constant<int, (v + 1) % BT_RING>{}, std::integral_constant<int, (v + BT_RING - 1) % BT_RING>{},
           std::integral_constant<int, v & 1>{});
    });
  bt_sfor<0, BT_RING - 1>([&](auto V) __attribute__((always_inline)) {
    constexpr int v = decltype(V)::value;
    if (g + v < nslab)
      step(g + v, std::integral_constant<int, (v + 1) % BT_RING>{}, std::integral_constant<int, (v + BT_RING - 1) % BT_RING>{},
           std::integral_constant<int, v & 1>{});
  });
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0070);                            // vmcnt(0) lgkmcnt(0): trailing re-loads and prefetches done
  __builtin_amdgcn_s_barrier();                                  // the ring becomes the epilogue's staging area
  asm volatile("" ::: "memory");

  float* stage = reinterpret_cast<float*>(lds) + wave * (32 * EPI_LD);
  const int m0 = (rt0 + 4 * wr) * 32, n0 = (ct0 + CW * wc) * 32;
  if (p.splits > 1) {
    float* ws = p.ws + (long long)zs * p.M * p.N;
    const int li = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const int n = n0 + 32 * j + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 32 * i + 4 * kh + (r & 3) + 8 * (r >> 2);
          if (m < p.M && n < p.N) ws[(long long)m * p.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  const bool packs = p.out_rows || p.out_trans || p.cs_part || p.dact_out || p.dact_in;
  const bool direct = packs && !p.e.pre && !p.e.res && !(p.e.mode == DPOT_EPI_DACT && !p.dact_in);
#pragma unroll 1
  for (int f = 0; f < 4 * CW; ++f) {
    const int fi = f / CW, fj = f - fi * CW;
    if (m0 + 32 * fi >= p.M) break;
    f32x16 af = acc[0][0];
    switch (f) {      // wave-uniform: a branch tree, not a select chain over 16 x 16 registers
#define BT_CASE(F)                                            \
  case F:                                                     \
    if constexpr ((F) < 4 * CW) af = acc[(F) / CW][(F) % CW]; \
    break;
      BT_CASE(1) BT_CASE(2) BT_CASE(3) BT_CASE(4) BT_CASE(5) BT_CASE(6) BT_CASE(7) BT_CASE(8)
      BT_CASE(9) BT_CASE(10) BT_CASE(11) BT_CASE(12) BT_CASE(13) BT_CASE(14) BT_CASE(15)
#undef BT_CASE
      default: break;
    }
    if (!packs) epi_fragment(p.e, 1, 0, m0 + 32 * fi, n0 + 32 * fj, af, stage, lane);
    else if (direct) epi_fragment_direct(p, m0 + 32 * fi, n0 + 32 * fj, af, stage, lane);
    else epi_fragment_pack(p, m0 + 32 * fi, n0 + 32 * fj, af, stage, lane);
  }
}

// host side: called by dpot_gemm_bf16p (csrc/gemm_bf16p.hip) with the argument block already filled in (tilesM / tilesN /
// super_r / super_c for the 256-row tile grid)
int bf16bt_launch(const Bf16pArgs& p, int colt, unsigned grid, hipStream_t stream) {
  if (colt == 6)
    hipLaunchKernelGGL(gemm_bf16bt_kernel<6>, dim3(grid, p.splits), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(gemm_bf16bt_kernel<8>, dim3(grid, p.splits), dim3(256), 0, stream, p);
  return check_launch("gemm_bf16bt_kernel");
}

}  // namespace dpot
