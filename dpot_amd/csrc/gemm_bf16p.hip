// gemm_bf16p.hip - bf16 "panel" GEMM on v_mfma_f32_32x32x16_bf16 for the channel-MLP of the larger DPOT models
// (BASELINE configs[2]: "bf16 channel-MLP on MFMA"; DPOT-M / -L spend 83-89 % of their FLOPs there):
//
//      C[M, N] (fp32) = epilogue( A[M, K] @ W[K, N] )        bf16 operands, fp32 accumulation
//
// Both operands are PRE-PACKED bf16 in fragment-block-major order - the (32 rows x 16 k) block of row tile r / K-slab t is
// 1 KiB contiguous, chunk l of a block = (row l&31, k 8*(l>>5)..+7) = the MFMA A/B operand order - so the GEMM loop is
// LDS-DMA + ds_read_b128 + MFMA only (the round-1 attempt converted inside a register-staged loop and was load-bound: a
// plain-bf16 instantiation of gemm_split.h is no faster than its 6-product form):
//   * weights are packed once per optimiser step (dpot_bf16_pack_jobs, all weights in one launch, W and W^T);
//   * activations are packed by dpot_bf16_pack_rows: one HBM pass (4 B read, 2 B written per element) per GEMM operand,
//     ~10 % of the GEMM's own time at K = 1024;
//   * workgroup tile 128 x 256, 8 compute waves in a 2 x 4 grid (wave tile 64 x 64 = 2 x 2 accumulators of 32x32) + 4
//     loader waves issuing the DMA (24 KiB per 32-k slab), ring of three slab buffers, loaders three slabs ahead with
//     counted vmcnt, ONE barrier per slab, fragments of slab t+1 fetched into a second register set under the MFMAs of
//     slab t (the structure of csrc/gemm_panel.hip / afno_mlp.hip);
//   * the fused epilogue of the fp32 kernels (gemm_epi.h): bias, pre-activation save, activation, act'(aux), residual.
#include "gemm_bf16p_common.h"

namespace dpot {

constexpr int PB_ROWT = 4;      // 32-row tiles per workgroup (128 rows)
constexpr int PB_COLT = 8;      // 32-column tiles per workgroup (256 columns)
#ifndef PB_NLOAD_N
#define PB_NLOAD_N 4
#endif
constexpr int PB_NLOAD = PB_NLOAD_N;                           // loader waves (4 or 8)
constexpr int PB_NPC = 2 * (PB_ROWT + PB_COLT) / PB_NLOAD;      // 1 KiB DMA pieces per loader wave and slab (6 or 3)
constexpr int PB_SLABB = (PB_ROWT + PB_COLT) * 2 * 1024;        // bytes of one 32-k slab: 24 KiB
// Depth of the slab ring (-DPB_RING_N=3..6).  Slab g + R goes into the slot of slab g, which is free at barrier B_g, and has
// to have landed by B_(g+R-1): the loop tolerates R - 1 slab periods of LDS-DMA latency.  Round 3 swept R = 3 .. 6 on the
// theory that the main loop (1350 cycles per slab against 512 of matrix-pipe time) is bound by that latency: no effect
// (fc1 forward at DPOT-M 150.0 / 150.0 / 150.9 / 149.9 us, profiles/r03_bf16p_ring_sweep.txt) - it is not; R stays 3.
#ifndef PB_RING_N
#define PB_RING_N 3
#endif
constexpr int PB_RING = PB_RING_N;

// body of the kernel: problem p, workgroup index bid0 within the problem, split-K index zs.
// COLT = 32-column tiles of the workgroup tile: 8 (128 x 256, 2 x 4 compute waves) or 6 (128 x 192, 2 x 3 compute waves + the
// same 4 loader waves: 20 KiB slabs) - the narrower tile exists for tile COUNTS, not for speed (see the pair launch)
template <int COLT>
__device__ __forceinline__ void gemm_bf16p_body(const Bf16pArgs& p, const int bid0, const int zs) {
  constexpr int NCW = COLT;                                       // compute waves: 2 x (COLT / 2), 64 x 64 each
  constexpr int NPC = 2 * (PB_ROWT + COLT) / PB_NLOAD;            // 1 KiB DMA pieces per loader wave and slab
  constexpr int SLABB = (PB_ROWT + COLT) * 2 * 1024;              // bytes of one 32-k slab
  static_assert(2 * (PB_ROWT + COLT) % PB_NLOAD == 0, "pieces must divide among the loader waves");
  // ring of PB_RING slabs; the epilogue stages through it afterwards (8 waves x 2 x 32 x EPI_LD floats = 72 KiB)
  static_assert(PB_RING >= 3 && PB_RING * SLABB <= 160 * 1024, "slab ring must fit the 160 KiB LDS");
  __shared__ __attribute__((aligned(16))) unsigned char lds[PB_RING * SLABB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks16 = p.K >> 4;                        // 16-k blocks per row tile
  const int slab0 = zs * p.slabs_per_split;         // first 32-k slab of this split (K % 32 == 0)
  int nslab = (p.K >> 5) - slab0;
  nslab = nslab < p.slabs_per_split ? nslab : p.slabs_per_split;

  int tm, tn;
  if (p.super_r > 0) {
    // L2-aware rasterisation.  Workgroup b runs on XCD b % 8, and an XCD's 32 CUs hold 32 consecutive slots b >> 3 at a
    // time: those 32 concurrent tiles are made ONE super_r x super_c block of the tile grid (8 x 4: 8 A row tiles + 4 W
    // column chunks feed 32 tiles through that chiplet's L2, against 32 + 1 for a column of tiles), and an XCD walks the
    // super-blocks of ONE super-row before the next (its A slice stays L2-resident while the W chunks stream)
    const int xcd = bid0 & 7, slot = bid0 >> 3;
    const int sb = slot >> 5, in = slot & 31;                 // super-block number of this XCD, position inside
    const int srows = p.tilesM / p.super_r, scols = p.tilesN / p.super_c;
    const int g = sb * 8 + xcd;                               // global super-block index, super-row fastest ...
    if (g >= srows * scols) return;                           // (grid padded to whole rounds of 8 super-blocks)
    const int srow = g % srows, scol = g / srows;             // ... so XCD x keeps super-row (x mod srows)
    tm = srow * p.super_r + in % p.super_r;
    tn = scol * p.super_c + in / p.super_r;
  } else {
    // XCD-contiguous tile order, column-tile major: workgroups sharing a weight chunk sit on one chiplet's L2
    const int ntiles = p.tilesM * p.tilesN;
    int tile;
    const int bid = bid0, xcd = bid & 7, slot = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    tn = tile / p.tilesM;
    tm = tile - tn * p.tilesM;
  }
  const int rt0 = tm * PB_ROWT, ct0 = tn * COLT;  // first 32-row tile / 32-column tile
  const int mtiles = (p.M + 31) >> 5;

  // s_waitcnt lgkmcnt(0) as the BUILTIN (simm16 0xC07F: vmcnt 63, expcnt 7, lgkmcnt 0), not inline asm: the compiler's
  // own wait-count pass cannot see through inline asm.  With the asm form it did not know that the fragments of the
  // current slab had landed, and put a second `s_waitcnt lgkmcnt(0)` in front of the first MFMA of every slab - AFTER the
  // ds_reads of the next slab had been issued: the MFMAs waited for the full LDS round trip of loads they do not use, and
  // matrix-pipe time and LDS time ADDED up instead of overlapping (ablation profiles/r03_bf16p_ablation.txt: 563 ns per
  // slab = 375 ns without the MFMAs + ~190 ns of MFMAs).
  auto bar = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");                      // (compiler fence: LDS accesses stay on their side)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // ablation builds (scripts/variant.sh gemm_bf16p NAME -DPB_ABL_...): timing experiments, the results are garbage
#ifdef PB_ABL_NOBAR
  auto lbar = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("" ::: "memory");
  };
#else
  auto lbar = bar;                                    // the per-slab barrier of the main loop
#endif

  if (wave >= NCW) {
    // ================================ loader waves: 24 blocks per slab, 6 per wave ================================
    const int L = wave - NCW;
    // block b of a slab: b < 8: A (row tile b>>1, k-half b&1); else W (column tile (b-8)>>1, k-half (b-8)&1)
    const unsigned short* src[NPC];
    int dst[NPC];
#pragma unroll
    for (int n = 0; n < NPC; ++n) {
      const int b = L + PB_NLOAD * n;
      if (b < 2 * PB_ROWT) {
        int rt = rt0 + (b >> 1);
        rt = rt < mtiles ? rt : mtiles - 1;          // clamped: rows past M only feed outputs that are never stored
        src[n] = p.A + ((long long)rt * ks16 + 2 * slab0 + (b & 1)) * 512 + lane * 8;
      } else {
        const int c = b - 2 * PB_ROWT;
        src[n] = p.W + ((long long)(ct0 + (c >> 1)) * ks16 + 2 * slab0 + (c & 1)) * 512 + lane * 8;
      }
      dst[n] = b * 1024;
    }
    auto issue = [&](int t, int ring) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < NPC; ++n) bglds16(src[n] + (long long)t * 1024, lds + ring * SLABB + dst[n]);
    };
    // (Round 3 also tried REGISTER-STAGED loaders - global_load_dwordx4 -> VGPRs -> ds_write_b128, every loader wave a whole
    // slab, three slab periods of latency tolerance - on the reading of the ablation that the LDS-DMA path itself is the
    // limit: 3.5x SLOWER, 256 us against 72 us for the K = 4096 main loop, profiles/r03_bf16p_regstage_rejected.txt.)
    // prologue: slabs 0 .. R-1 in flight; slabs 0 and 1 (the oldest 12 pieces) must have landed at barrier P: the
    // compute waves fetch slab 0 right after P and slab 1 after B_0, and nothing is waited for in between
    int issued = 0;
#pragma unroll
    for (int r = 0; r < PB_RING; ++r)
      if (r < nslab) { issue(r, r); ++issued; }
    bwait_vm_dyn(issued > 2 ? NPC * (issued - 2) : 0);
    bar();                                              // P
    int ring = 0;
#pragma unroll 1
    for (int g = 0; g < nslab; ++g) {
      lbar();                                           // B_g: every wave holds slab g in registers -> its slot is free
      if (g + PB_RING < nslab) issue(g + PB_RING, ring);
      ring = ring == PB_RING - 1 ? 0 : ring + 1;
      // slab g + 2 must have landed before B_(g+1) (the compute waves fetch its fragments there): everything younger
      // may stay in flight
      int younger = nslab - (g + 3);                    // slabs g+3 .. that have been issued
      younger = younger < 0 ? 0 : (younger > PB_RING - 2 ? PB_RING - 2 : younger);
      bwait_vm_dyn(NPC * younger);
    }
    bar();                                              // S
    return;
  }

  // ================================ compute waves ================================
  const int wm = wave / (COLT / 2), wn = wave - wm * (COLT / 2);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragments of one slab: a[i][ks], b[j][ks]  (ks = 16-k half of the 32-k slab)
  auto read_frags = [&](bf16x8_t (&a)[2][2], bf16x8_t (&b)[2][2], int ring) __attribute__((always_inline)) {
    const unsigned char* base = lds + ring * SLABB + lane * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        a[i][ks] = *reinterpret_cast<const bf16x8_t*>(base + ((2 * wm + i) * 2 + ks) * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#ifdef PB_ABL_NOBREAD
        b[j][ks] = a[j][ks];
#else
        b[j][ks] = *reinterpret_cast<const bf16x8_t*>(base + (2 * PB_ROWT + (2 * wn + j) * 2 + ks) * 1024);
#endif
  };
  auto mma = [&](const bf16x8_t (&a)[2][2], const bf16x8_t (&b)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#ifdef PB_ABL_NOMMA
          acc[i][j][0] += (float)a[i][ks][0] + (float)b[j][ks][0];   // keeps the fragment reads alive
#else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], b[j][ks], acc[i][j], 0, 0, 0);
#endif
  };

  bf16x8_t aA[2][2], bA[2][2], aB[2][2], bB[2][2];
  bar();                                               // P
  read_frags(aA, bA, 0);
  {
    // branch-free slab pairs: the fragment reads are unconditional (past the last slab they fetch stale bytes of a valid
    // ring slot that nobody uses) - a conditional read puts a control-flow join in front of the MFMAs, where the compiler
    // no longer knows how many LDS operations are outstanding and waits for all of them
    int r1 = 1 % PB_RING, r2 = 2 % PB_RING;            // ring slots of slabs t+1, t+2
    int t = 0;
#pragma unroll 1
    for (; t + 1 < nslab; t += 2) {
      // (sched_barrier: the fragment reads of the NEXT slab are issued first, then the MFMAs of the current one run under
      // them; left alone, the scheduler sinks the reads towards their uses and the prefetch distance is gone)
      lbar();                                          // B_t
      read_frags(aB, bB, r1);
      __builtin_amdgcn_sched_barrier(0);
      mma(aA, bA);
      __builtin_amdgcn_sched_barrier(0);
      lbar();                                          // B_(t+1)
      read_frags(aA, bA, r2);
      __builtin_amdgcn_sched_barrier(0);
      mma(aB, bB);
      __builtin_amdgcn_sched_barrier(0);
      r1 = r1 + 2 >= PB_RING ? r1 + 2 - PB_RING : r1 + 2;
      r2 = r2 + 2 >= PB_RING ? r2 + 2 - PB_RING : r2 + 2;
    }
    if (t < nslab) {                                   // odd slab count (split-K ranges)
      lbar();                                          // B_t
      mma(aA, bA);
    }
  }
  bar();                                               // S: all DMA landed and read; the ring becomes epilogue staging
#ifdef PB_ABL_NOEPI
  if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(lds)[lane] = acc[1][1][3] + acc[0][1][2] + acc[1][0][1];
  return;
#endif

  float* stage = reinterpret_cast<float*>(lds) + wave * (32 * EPI_LD);
  const int m0 = (rt0 + 2 * wm) * 32, n0 = (ct0 + 2 * wn) * 32;
  if (p.splits > 1) {
    // split-K partial: raw accumulator fragments to the workspace (C/D layout: col = lane&31, row = perm(r, lane>>5));
    // dpot::splitk_reduce_kernel (gemm.hip) sums the splits in a fixed order and applies the epilogue
    float* ws = p.ws + (long long)zs * p.M * p.N;
    const int li = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + 32 * j + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 32 * i + 4 * kh + (r & 3) + 8 * (r >> 2);
          if (m < p.M && n < p.N) ws[(long long)m * p.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  if (p.out_rows || p.out_trans || p.cs_part || p.dact_out || p.dact_in) {   // host-checked: M % 32 == 0
    float* st1 = reinterpret_cast<float*>(lds) + wave * (32 * EPI_LD);
    // ONE copy of the fragment epilogue, looped over the wave's four accumulators (moved into a common register set):
    // inlined four times the epilogue is ~25 k instructions of straight-line code that every wave runs exactly once
    // (host-checked: act' packs only in the direct form)
    const bool direct = !p.e.pre && !p.e.res && !(p.e.mode == DPOT_EPI_DACT && !p.dact_in);
#pragma unroll 1
    for (int f = 0; f < 4; ++f) {
      if (m0 + 32 * (f >> 1) >= p.M) break;
      const f32x16 a = f == 0 ? acc[0][0] : f == 1 ? acc[0][1] : f == 2 ? acc[1][0] : acc[1][1];
      if (direct) epi_fragment_direct(p, m0 + 32 * (f >> 1), n0 + 32 * (f & 1), a, st1, lane);
      else epi_fragment_pack(p, m0 + 32 * (f >> 1), n0 + 32 * (f & 1), a, st1, lane);
    }
    return;
  }
#pragma unroll 1
  for (int f = 0; f < 4; ++f) {
    const f32x16 a = f == 0 ? acc[0][0] : f == 1 ? acc[0][1] : f == 2 ? acc[1][0] : acc[1][1];
    epi_fragment(p.e, 1, 0, m0 + 32 * (f >> 1), n0 + 32 * (f & 1), a, stage, lane);
  }
}

template <int COLT>
__global__ __launch_bounds__(64 * (COLT + PB_NLOAD)) void gemm_bf16p_kernel(const Bf16pArgs p) {
  gemm_bf16p_body<COLT>(p, blockIdx.x, blockIdx.y);
}

// (A 256 x 256-tile variant - 8 self-loading waves of 64 x 128, ring of four 32 KiB slabs, a quarter of the workgroups -
// was built and measured in round 3 on the theory that the main loop is bound by operand latency: it is SLOWER on every
// DPOT-M / -L shape (fc1 forward at DPOT-M 167.6 against 145.6 us, fc2 forward 134.6 against 92.3 us;
// profiles/r03_bf16p_train_bench_tile256_rejected.txt) and was removed.)

// (Rounds 3-5 also carried a "duo" form of this kernel - two 8-wave workgroups per CU for launches with several rounds of tiles;
// the B-direct kernels below took those launches in round 4 and it only ran behind an opt-out switch since: removed in round 6.)
// ---------------------------------------------------------------------------------------------------------------------
// "B-direct" form (round 4).  Round 3's ablations say the main loop of the kernels above is bound by the operand path:
// ~845 CU cycles per 24 KiB slab whatever the clock, i.e. LDS-DMA (global_load ... lds) delivers ~29 B/clk/CU here - the
// guide's own ceiling for that engine is ~37 B/clk/CU, against ~56 B/clk/CU for plain global loads out of L2.  In this
// form only the A panel (the 128 rows of the tile: 8 KiB per 32-k slab) still goes through LDS; the W operand never
// does: the workgroup's 8 waves are laid out 1 x 8 over the COLUMNS (wave tile 128 x 32 = 4 x 1 accumulators), so every
// wave needs 32 columns of W that no other wave of the workgroup touches, and it loads those fragments STRAIGHT into
// registers - a packed (32 columns x 16 k) block is 1 KiB in exactly the MFMA B-operand order, lane l reads bytes
// 16 l .. 16 l + 15 with one global_load_dwordx4, and a wave's column slice is one contiguous stream of K / 16 such blocks.
//   * per slab and wave: ONE LDS-DMA piece of A (8 per slab), TWO 16-byte-per-lane global loads of W, 8 fragment reads of
//     A from LDS, 8 MFMAs; all three memory operations of slab g + 3 are issued at barrier B_g (one counted vmcnt per slab:
//     the counter is in issue order, "at most 9 outstanding" = everything of slab g has arrived);
//   * LDS: ring of four 8 KiB A slabs (32 KiB) - a third of the operand bytes cross LDS (DMA write + fragment reads:
//     8 + 64 KiB per slab instead of 24 + 64);
//   * W fragments of the slabs in flight live in a register ring of 4 x 8 VGPRs (loop unrolled by four: the ring index
//     must be a compile-time constant).
// Same tiles (128 x 256 / 128 x 192), tile order, split-K and epilogues as gemm_bf16p_body.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BD_ASLAB = 2 * PB_ROWT * 1024;   // bytes of A per 32-k slab (4 row tiles x 2 k-halves x 1 KiB)
template <int B, int E, class F>
__device__ __forceinline__ void bd_sfor(F&& f) {   // static for: f(integral_constant<int, i>) for i in [B, E)
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    bd_sfor<B + 1, E>(f);
  }
}

// CPW = 32-column tiles per wave: 1 -> COLT waves of 128 x 32 (one workgroup per CU, <= 256 VGPRs); 2 -> COLT / 2 waves of
// 128 x 64 = 4 x 2 accumulators (128 accumulator registers; FOUR waves at COLT = 8, so two workgroups share a CU at two
// waves per SIMD): each A fragment read from LDS then feeds two MFMAs - 32 instead of 64 KiB of fragment reads per slab, the
// LDS side drops from ~576 to ~320 cycles per slab against 512 of matrix-pipe time - and the workgroup that is in its
// epilogue leaves the matrix pipes to its neighbour.
// BD_P = slabs of look-ahead (the loop is latency bound: period = load latency / look-ahead while that exceeds the 512
// cycles of matrix-pipe time per slab), BD_RING = BD_P + 1 A slots in LDS / W register sets; the slab loop is unrolled by
// the ring depth so that slots and register sets are compile-time constants.
template <int COLT, int CPW, int BD_P>
__device__ __forceinline__ void gemm_bf16p_bd_body(const Bf16pArgs& p, const int bid0, const int zs) {
  constexpr int BD_RING = BD_P + 1;
  static_assert(COLT % CPW == 0, "column tiles must divide among the waves");
  constexpr int NW = COLT / CPW;                                  // waves of the workgroup
  constexpr int STAGEB = NW * 32 * EPI_LD * 4;                    // epilogue staging, one 32 x EPI_LD slab per wave
  constexpr int LDSB = BD_RING * BD_ASLAB > STAGEB ? BD_RING * BD_ASLAB : STAGEB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks16 = p.K >> 4;
  const int slab0 = zs * p.slabs_per_split;
  int nslab = (p.K >> 5) - slab0;
  nslab = nslab < p.slabs_per_split ? nslab : p.slabs_per_split;

  int tm, tn;
  if (p.super_r > 0) {
    // L2-aware rasterisation (as in gemm_bf16p_body): the tiles an XCD runs at one time form super_r x super_c blocks of the
    // tile grid, and an XCD walks the super-blocks of one super-row before the next
    const int xcd = bid0 & 7, slot = bid0 >> 3;
    const int sb = slot >> 5, in = slot & 31;
    const int srows = p.tilesM / p.super_r, scols = p.tilesN / p.super_c;
    const int gsb = sb * 8 + xcd;
    if (gsb >= srows * scols) return;
    const int srow = gsb % srows, scol = gsb / srows;
    tm = srow * p.super_r + in % p.super_r;
    tn = scol * p.super_c + in / p.super_r;
  } else {
    // XCD-contiguous order: XCD x runs the tiles [x n / 8, (x + 1) n / 8) of the COLUMN-major (super_c == 0: an XCD holds few
    // W chunks and many A panels) or ROW-major (super_c == 1: few A panels, many W chunks) enumeration - the host picks the
    // one that makes the eight L2s fetch fewer operand bytes (bf16p_row_major)
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
  const int rt0 = tm * PB_ROWT, ct0 = tn * COLT;
  const int mtiles = (p.M + 31) >> 5;
  // (A deliberate phase skew between the two workgroups of a CU - every second workgroup sleeping 7-14 us before its first
  // slab, three choices of "second" - was measured on the theory that their epilogues coincide: slower by about the sleep in
  // every case, profiles/r04_bf16p_bd_skew_rejected.txt.  The epilogues do not overlap the neighbour's main loop either way.)
  // this wave's DMA piece of an A slab: piece b = wave (b < 8: row tile b >> 1, k-half b & 1); waves 8.. (none at COLT <= 8)
  // and, at COLT = 6, pieces 6 and 7 are taken by waves 0 and 1 as a second piece
  constexpr int NPA = (2 * PB_ROWT + NW - 1) / NW;                // A pieces per wave and slab (1 at 8 waves, 2 at 6)
  const unsigned short* asrc[NPA];
  int adst[NPA];
  bool ahas[NPA];
#pragma unroll
  for (int n = 0; n < NPA; ++n) {
    const int b = wave + NW * n;
    ahas[n] = b < 2 * PB_ROWT;
    const int bb = ahas[n] ? b : 0;
    int rt = rt0 + (bb >> 1);
    rt = rt < mtiles ? rt : mtiles - 1;
    asrc[n] = p.A + ((long long)rt * ks16 + 2 * slab0 + (bb & 1)) * 512 + lane * 8;
    adst[n] = bb * 1024;
  }
  constexpr long long astride = 1024;                             // elements between consecutive slabs of A
  // this wave's W streams: column tiles ct0 + CPW wave + j, blocks 2 (slab0 + t) + ks
  const unsigned short* wsrc = p.W + ((long long)(ct0 + CPW * wave) * ks16 + 2 * slab0) * 512 + lane * 8;
  const long long wcol = (long long)ks16 * 512;                   // elements between neighbouring column tiles

  f32x16 acc[PB_ROWT][CPW];
#pragma unroll
  for (int i = 0; i < PB_ROWT; ++i)
#pragma unroll
    for (int j = 0; j < CPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8_t wreg[BD_RING][CPW][2];

  // memory operations of slab t (ring slot / register set s): per wave NPA DMA pieces (issued by every wave, a piece the wave
  // does not own re-loads piece 0: uniform count) + 2 W loads = NPA + 2 vmcnt events
  // BRANCH-FREE: every slab step issues the operations of slab g + 3 - past the end it re-loads the LAST slab into a ring
  // slot / register set that nobody reads.  (A conditional issue puts a control-flow join in front of the MFMAs, behind
  // which the compiler no longer knows how many memory operations are outstanding and waits for all of them - i.e. for the
  // loads it has just issued: vmcnt(0) in one of every four slabs of the first build.)  The count of outstanding
  // operations is then the same at every barrier: 3 slabs x NOP, and "slab g has arrived" is vmcnt(2 NOP) throughout.
  const int last = nslab - 1;
  auto issue = [&](int t, auto S) __attribute__((always_inline)) {
    constexpr int s = decltype(S)::value;
    t = t < last ? t : last;
#pragma unroll
    for (int n = 0; n < NPA; ++n) bglds16(asrc[n] + (long long)t * astride, lds + s * BD_ASLAB + adst[n]);
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const bf16x8_t* wp = reinterpret_cast<const bf16x8_t*>(wsrc + j * wcol + (long long)t * 1024);
      wreg[s][j][0] = wp[0];                                // default cache policy: the other row tiles' workgroups on
      wreg[s][j][1] = wp[64];                               // this XCD read the same W slice out of L2
    }
  };
  constexpr int NOP = NPA + 2 * CPW;                        // vmcnt events per slab and wave
  constexpr int VMW = NOP * (BD_P - 1);                     // outstanding operations allowed when slab g is needed
  // s_waitcnt vmcnt(VMW) lgkmcnt(0) as the BUILTIN (the compiler's own wait-count pass sees it): simm16 = vmcnt[3:0] |
  // expcnt 7 << 4 | lgkmcnt 0 << 8 | vmcnt[5:4] << 14
  constexpr int WAITC = (VMW & 15) | 0x70 | ((VMW >> 4) << 14);
  // (Two fragment-read schedules were built as variants in round 4 - all eight ds_read_b128 of a slab issued before its first
  // MFMA, and the next slab's A fragments prefetched into a second register set under the current MFMAs (the structure of
  // the LDS-DMA kernels): neither is faster on any DPOT-M / -L shape, the second spills at two column tiles per wave;
  // profiles/r04_bf16p_bd_fragment_variants.txt.  The compiler's own just-in-time pairs stay.)
  auto slab = [&](int g, auto S, auto SN) __attribute__((always_inline)) {
    constexpr int s = decltype(S)::value;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(WAITC);                      // slab g has arrived (DMA piece + W fragments); own LDS reads done
    __builtin_amdgcn_s_barrier();                           // B_g: slab g complete in LDS; slab g - 1 consumed by every wave
    asm volatile("" ::: "memory");
    issue(g + BD_P, SN);                                    // into the slot / register set of slab g - 1
    bf16x8_t a[PB_ROWT][2];
    const unsigned char* base = lds + s * BD_ASLAB + lane * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < PB_ROWT; ++i) a[i][ks] = *reinterpret_cast<const bf16x8_t*>(base + (i * 2 + ks) * 1024);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < PB_ROWT; ++i)
#pragma unroll
        for (int j = 0; j < CPW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], wreg[s][j][ks], acc[i][j], 0, 0, 0);
  };
  bd_sfor<0, BD_P>([&](auto S) __attribute__((always_inline)) { issue(decltype(S)::value, S); });
  int g = 0;
#pragma unroll 1
  for (; g + BD_RING <= nslab; g += BD_RING)
    bd_sfor<0, BD_RING>([&](auto S) __attribute__((always_inline)) {
      constexpr int sv = decltype(S)::value;
      slab(g + sv, S, std::integral_constant<int, (sv + BD_RING - 1) % BD_RING>{});
    });
  bd_sfor<0, BD_RING - 1>([&](auto S) __attribute__((always_inline)) {     // tail: g is a multiple of the ring depth
    constexpr int sv = decltype(S)::value;
    if (g + sv < nslab) slab(g + sv, S, std::integral_constant<int, (sv + BD_RING - 1) % BD_RING>{});
  });
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0070);                       // vmcnt(0) lgkmcnt(0): the trailing re-loads have landed too
  __builtin_amdgcn_s_barrier();                             // S: every wave has read the last slab; the ring becomes staging
  asm volatile("" ::: "memory");

  float* stage = reinterpret_cast<float*>(lds) + wave * (32 * EPI_LD);
  const int m0 = rt0 * 32, n0 = (ct0 + CPW * wave) * 32;
  if (p.splits > 1) {
    float* ws = p.ws + (long long)zs * p.M * p.N;
    const int li = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int i = 0; i < PB_ROWT; ++i)
#pragma unroll
      for (int j = 0; j < CPW; ++j) {
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
  // ONE copy of the fragment epilogue, looped over the wave's accumulators (moved into a common register set)
#pragma unroll 1
  for (int f = 0; f < PB_ROWT * CPW; ++f) {
    const int fi = f / CPW, fj = f - fi * CPW;
    if (m0 + 32 * fi >= p.M) break;
    f32x16 af;
    if constexpr (CPW == 1) {
      af = f == 0 ? acc[0][0] : f == 1 ? acc[1][0] : f == 2 ? acc[2][0] : acc[3][0];
    } else {
      af = f == 0 ? acc[0][0] : f == 1 ? acc[0][1] : f == 2 ? acc[1][0] : f == 3 ? acc[1][1] : f == 4 ? acc[2][0]
         : f == 5 ? acc[2][1] : f == 6 ? acc[3][0] : acc[3][1];
    }
    if (!packs) epi_fragment(p.e, 1, 0, m0 + 32 * fi, n0 + 32 * fj, af, stage, lane);
    else if (direct) epi_fragment_direct(p, m0 + 32 * fi, n0 + 32 * fj, af, stage, lane);
    else epi_fragment_pack(p, m0 + 32 * fi, n0 + 32 * fj, af, stage, lane);
  }
}

// (A PERSISTENT form - grid.x = the resident workgroup slots, every workgroup looping over its tiles, on the theory that a
// workgroup which ENDS holds its slot until its pack stores are acknowledged while a persistent one starts the next tile's
// loads under that drain - was built and measured in round 4: no effect on any launch or step, DPOT-M 14.05 / 14.08 ms,
// DPOT-L 95.5 / 95.6 ms, profiles/r04_bf16p_bd_persistent_rejected.txt; it cost 30 VGPRs and was removed.)
template <int COLT, int CPW, int P>
__global__ __launch_bounds__(64 * COLT / CPW, 2) void gemm_bf16p_bd_kernel(const Bf16pArgs p) {
  static_assert(CPW <= 2, "the epilogue's accumulator select covers 4 x 2 fragments");
  gemm_bf16p_bd_body<COLT, CPW, P>(p, blockIdx.x, blockIdx.y);
}

// two independent problems in ONE launch (the fc1 and fc2 weight gradients of a block: 128 tiles each at DPOT-M - alone
// each needs split-K 2, i.e. 2 x 16 MB of partial sums and a reduce launch, to fill 256 CUs; together they fill them)
struct Bf16pPair {
  Bf16pArgs a[2];
  int n0;          // workgroups of problem 0
};
template <int COLT>
__global__ __launch_bounds__(64 * (COLT + PB_NLOAD)) void gemm_bf16p_pair_kernel(const Bf16pPair pp) {
  const int which = (int)blockIdx.x >= pp.n0 ? 1 : 0;
  gemm_bf16p_body<COLT>(pp.a[which], (int)blockIdx.x - which * pp.n0, blockIdx.y);   // blockIdx.y: common split-K index
}

template <int COLT, int CPW, int P>
__global__ __launch_bounds__(64 * COLT / CPW, 2) void gemm_bf16p_bd_pair_kernel(const Bf16pPair pp) {
  const int which = (int)blockIdx.x >= pp.n0 ? 1 : 0;
  gemm_bf16p_bd_body<COLT, CPW, P>(pp.a[which], (int)blockIdx.x - which * pp.n0, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x6: the fp32-accurate form.  Operands packed as THREE bf16 planes (x = x1 + x2 + x3, 8 significand bits each, see
// gemm_split.h), block (32 rows x 16 k) = 3 KiB = plane-major 3 x 1 KiB; six of the nine plane products are accumulated
// (x1y1, x1y2, x2y1, x2y2, x1y3, x3y1; the dropped ones are < 2^-24 relative).  Same tile / wave grid as above, but
//   * 16-k slabs (36 KiB: 12 A + 24 W blocks), ring of three, one barrier per slab;
//   * no loader waves: the fragments of a slab are 12 x ds_read_b128 = 48 VGPRs, double buffered 96 + 64 accumulators -
//     more than the 168 registers a 12-wave workgroup leaves per wave - so the 8 compute waves (256 registers each) issue
//     the LDS-DMA themselves: 5 one-KiB pieces per wave and slab (4 waves re-load a piece twice to keep the per-wave
//     count, and with it the `vmcnt` immediate, uniform), three slabs ahead;
//   * 24 MFMAs (32x32x16 bf16) per wave and slab against 12 fragment reads: the matrix pipe is the bound
//     (2.5 PF / 6 = 417 TFLOP/s fp32-equivalent).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int X6_SLABB = (PB_ROWT + PB_COLT) * 3 * 1024;     // bytes of one 16-k slab: 36 KiB
constexpr int X6_NI = 5;                                      // DMA instructions per wave and slab

__global__ __launch_bounds__(512) void gemm_bf16x6p_kernel(const Bf16pArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * X6_SLABB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks16 = p.K >> 4;                        // 16-k blocks per row tile
  const int zs = blockIdx.y;
  const int slab0 = zs * p.slabs_per_split;         // in 16-k slabs
  int nslab = ks16 - slab0;
  nslab = nslab < p.slabs_per_split ? nslab : p.slabs_per_split;

  const int ntiles = p.tilesM * p.tilesN;
  int tile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tn = tile / p.tilesM, tm = tile - tn * p.tilesM;
  const int rt0 = tm * PB_ROWT, ct0 = tn * PB_COLT;
  const int mtiles = (p.M + 31) >> 5;

  auto bar = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // this wave's DMA pieces of a slab: piece b = (unit u = b / 3, plane b % 3); units 0-3: A row tiles, 4-11: W col tiles
  const unsigned short* src[X6_NI];
  int dst[X6_NI];
#pragma unroll
  for (int n = 0; n < X6_NI; ++n) {
    int b = wave + 8 * n;
    if (b >= 36) b -= 32;                            // waves 4-7: a second copy of their first piece (uniform count)
    const int u = b / 3, pl = b - 3 * u;
    if (u < PB_ROWT) {
      int rt = rt0 + u;
      rt = rt < mtiles ? rt : mtiles - 1;
      src[n] = p.A + (((long long)rt * ks16 + slab0) * 3 + pl) * 512 + lane * 8;
    } else {
      src[n] = p.W + (((long long)(ct0 + u - PB_ROWT) * ks16 + slab0) * 3 + pl) * 512 + lane * 8;
    }
    dst[n] = b * 1024;
  }
  auto issue = [&](int t, int ring) __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < X6_NI; ++n) bglds16(src[n] + (long long)t * 1536, lds + ring * X6_SLABB + dst[n]);
  };

  const int wm = wave >> 2, wn = wave & 3;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto read_frags = [&](bf16x8_t (&a)[2][3], bf16x8_t (&b)[2][3], int ring) __attribute__((always_inline)) {
    const unsigned char* base = lds + ring * X6_SLABB + lane * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        a[i][pl] = *reinterpret_cast<const bf16x8_t*>(base + ((2 * wm + i) * 3 + pl) * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        b[j][pl] = *reinterpret_cast<const bf16x8_t*>(base + ((PB_ROWT + 2 * wn + j) * 3 + pl) * 1024);
  };
  auto mma = [&](const bf16x8_t (&a)[2][3], const bf16x8_t (&b)[2][3]) __attribute__((always_inline)) {
    // smallest terms first; every accumulator sees the same order on every slab (deterministic)
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PBn[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t]], b[j][PBn[t]], acc[i][j], 0, 0, 0);
  };

  bf16x8_t aA[2][3], bA[2][3], aB[2][3], bB[2][3];
  issue(0, 0);
  if (nslab > 1) issue(1, 1);
  if (nslab > 2) issue(2, 2);
  if (nslab > 2) bwait_vm<2 * X6_NI>(); else if (nslab > 1) bwait_vm<X6_NI>(); else bwait_vm<0>();
  bar();                                               // P: slab 0 landed everywhere
  read_frags(aA, bA, 0);
  {
    int ring = 0;                                      // ring slot of slab t
    auto step = [&](int t, bf16x8_t (&ca)[2][3], bf16x8_t (&cb)[2][3], bf16x8_t (&na)[2][3], bf16x8_t (&nb)[2][3])
        __attribute__((always_inline)) {
      // own DMA still in flight here: slabs t+1 and t+2 (if they exist); slab t+1 must have landed before B_t
      if (t + 2 < nslab) bwait_vm<X6_NI>(); else bwait_vm<0>();
      bar();                                           // B_t: slab t+1 landed; every wave holds slab t in registers
      if (t + 3 < nslab) issue(t + 3, ring);           // slot of slab t is free
      const int rn = ring == 2 ? 0 : ring + 1;
      if (t + 1 < nslab) read_frags(na, nb, rn);
      mma(ca, cb);
      ring = rn;
    };
#pragma unroll 1
    for (int t = 0; t < nslab; t += 2) {
      step(t, aA, bA, aB, bB);
      if (t + 1 < nslab) step(t + 1, aB, bB, aA, bA);
    }
  }
  bar();                                               // S: the ring becomes epilogue staging

  float* stage = reinterpret_cast<float*>(lds) + wave * (32 * EPI_LD);
  const int m0 = (rt0 + 2 * wm) * 32, n0 = (ct0 + 2 * wn) * 32;
  if (p.splits > 1) {
    float* ws = p.ws + (long long)zs * p.M * p.N;
    const int li = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + 32 * j + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + 32 * i + 4 * kh + (r & 3) + 8 * (r >> 2);
          if (m < p.M && n < p.N) ws[(long long)m * p.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  epi_fragment(p.e, 1, 0, m0, n0, acc[0][0], stage, lane);
  epi_fragment(p.e, 1, 0, m0, n0 + 32, acc[0][1], stage, lane);
  epi_fragment(p.e, 1, 0, m0 + 32, n0, acc[1][0], stage, lane);
  epi_fragment(p.e, 1, 0, m0 + 32, n0 + 32, acc[1][1], stage, lane);
}


// 8 consecutive k of one row -> NPL packed planes (16 bytes each); plane p holds bf16(x - sum of the planes before it)
template <int NPL>
__device__ __forceinline__ void pack8(const float (&v)[8], uint4 (&o)[NPL]) {
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = v[e];
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl) {
    unsigned w[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      w[h] = pack2(r[2 * h], r[2 * h + 1]);
      if (pl + 1 < NPL) {
        r[2 * h] -= __uint_as_float(w[h] << 16);
        r[2 * h + 1] -= __uint_as_float(w[h] & 0xffff0000u);
      }
    }
    o[pl] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// activations: src fp32 [R, K] row-major (ld) -> [ceil(R/32)][K/16][NPL planes][64 chunks][8 bf16]; rows past R are zero.
// one thread = one 16-byte chunk (row, 8 consecutive k); a wave covers 4 rows x 128 k (512 B per row: full lines)
// trans: the logical operand is the TRANSPOSE of the stored matrix (src [K, R] row-major): element (row, k) =
// src[k*ld + row] (weight gradients: rows = features, k = tokens); adjacent threads then take adjacent rows, so each of
// a thread's 8 loads is part of a contiguous row segment of the source
template <int NPL>
__global__ __launch_bounds__(256) void bf16_pack_rows_kernel(const float* __restrict__ src, int ld, int R, int K,
                                                             uint4* __restrict__ dst, long long nchunks, int trans) {
  const int kc_per_row = K >> 3;
  const int Rp = (R + 31) & ~31;
  for (long long c = blockIdx.x * 256ll + threadIdx.x; c < nchunks; c += (long long)gridDim.x * 256) {
    int kc, row;
    if (trans) {
      row = (int)(c % Rp);
      kc = (int)(c / Rp);
    } else {
      kc = (int)(c % kc_per_row);
      row = (int)(c / kc_per_row);
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (row < R) {
      if (trans) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[(long long)(8 * kc + e) * ld + row];
      } else {
        const float4 x0 = *reinterpret_cast<const float4*>(src + (long long)row * ld + 8 * kc);
        const float4 x1 = *reinterpret_cast<const float4*>(src + (long long)row * ld + 8 * kc + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
      }
    }
    uint4 o[NPL];
    pack8<NPL>(v, o);
    // block (row tile, 16-k slab kc>>1), chunk (row & 31) + 32 * (kc & 1)
    uint4* d = dst + (((long long)(row >> 5) * (K >> 4) + (kc >> 1)) * NPL) * 64 + (row & 31) + 32 * (kc & 1);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) d[pl * 64] = o[pl];
  }
}

// One pass over an activation matrix src [R, K] (tokens x features) that produces BOTH packed forms the channel MLP
// needs - the row form (GEMM A operand: rows = tokens, k = features) and the transposed form (weight-gradient operand:
// rows = features, k = tokens) - and, optionally, per-block partial column sums (the bias gradient).
// Block = 64 tokens x 256 features, staged through LDS (pitch 260 floats: the 16-byte fragment reads of 32 consecutive
// rows fall on all 64 banks), so that every wave-instruction stores ONE whole 1 KiB fragment block - 64 consecutive
// 16-byte chunks.  (Round 2 kept an 8 x 8 patch per thread and stored its chunks straight from registers: a store
// instruction scattered its 64 chunks over 64 different blocks, 20 us for 67 MB at DPOT-M = 3.3 TB/s.)
// R % 64 == 0, K % 256 == 0.
constexpr int PKB_PITCH = 260;
// optional GroupNorm on the way in (dpot_bf16_pack_both_norm): the packs hold GN(src) = (src - mean) * rstd * gamma + beta with
// the statistics of (sample = row / rows_per_sample, group = column / (K / G)) - the channel MLP's input GroupNorm2(y1) is then
// never written in fp32 (the expression is the one of the GroupNorm apply kernels: bit-identical packs)
struct PackNorm {
  const float* mean;     // [samples, G]; NULL: plain pack
  const float* rstd;
  const float* gamma;    // [K]
  const float* beta;
  int rows_per_sample, G;
};
__global__ __launch_bounds__(256) void bf16_pack_both_kernel(const float* __restrict__ src, int ld, int R, int K,
                                                             uint4* __restrict__ drow, uint4* __restrict__ dtr,
                                                             float* __restrict__ cpart, const PackNorm nrm) {
  __shared__ __attribute__((aligned(16))) float tile[64 * PKB_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * 256, r0 = blockIdx.y * 64;
  float mu = 0.f;
  float4 ga = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nrm.mean) {      // the 64 rows of a block lie in ONE sample (host-checked), a lane's 4 columns in one group
    const int sg = (r0 / nrm.rows_per_sample) * nrm.G + (c0 + 4 * lane) / (K / nrm.G);
    mu = nrm.mean[sg];
    const float rs = nrm.rstd[sg];
    ga = *reinterpret_cast<const float4*>(nrm.gamma + c0 + 4 * lane);
    be = *reinterpret_cast<const float4*>(nrm.beta + c0 + 4 * lane);
    ga.x *= rs; ga.y *= rs; ga.z *= rs; ga.w *= rs;
  }
  // phase 1: 64 rows x 1 KiB, a wave per row (16 rows per wave), straight into the LDS tile; column sums on the way
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = wave + 4 * i;
    float4 v = *reinterpret_cast<const float4*>(src + (long long)(r0 + row) * ld + c0 + 4 * lane);
    if (nrm.mean) v = make_float4(fmaf(v.x - mu, ga.x, be.x), fmaf(v.y - mu, ga.y, be.y), fmaf(v.z - mu, ga.z, be.z),
                                  fmaf(v.w - mu, ga.w, be.w));
    *reinterpret_cast<float4*>(&tile[row * PKB_PITCH + 4 * lane]) = v;
    cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
  }
  __syncthreads();
  if (drow) {   // block (row tile rt, 16-k block kb): chunk l = (row l & 31, features 8 * (l >> 5) .. + 7)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int b = wave + 4 * q, rt = b >> 4, kb = b & 15;
      const float* t = &tile[(32 * rt + (lane & 31)) * PKB_PITCH + 16 * kb + 8 * (lane >> 5)];
      const float4 x0 = *reinterpret_cast<const float4*>(t), x1 = *reinterpret_cast<const float4*>(t + 4);
      drow[((long long)((r0 >> 5) + rt) * (K >> 4) + (c0 >> 4) + kb) * 64 + lane] =
          make_uint4(pack2(x0.x, x0.y), pack2(x0.z, x0.w), pack2(x1.x, x1.y), pack2(x1.z, x1.w));
    }
  }
  if (dtr) {    // block (feature tile ft, 16-token block tb): chunk l = (feature l & 31, tokens 8 * (l >> 5) .. + 7)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int b = wave + 4 * q, ft = b >> 2, tb = b & 3;
      const float* t = &tile[(16 * tb + 8 * (lane >> 5)) * PKB_PITCH + 32 * ft + (lane & 31)];
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = t[i * PKB_PITCH];
      dtr[((long long)((c0 >> 5) + ft) * (R >> 4) + (r0 >> 4) + tb) * 64 + lane] =
          make_uint4(pack2(x[0], x[1]), pack2(x[2], x[3]), pack2(x[4], x[5]), pack2(x[6], x[7]));
    }
  }
  if (cpart) {  // column sums of this block's 64 tokens, fixed order: 16 rows in the thread, then the 4 waves
    __syncthreads();                                       // the tile is dead: its head becomes the reduction scratch
    *reinterpret_cast<float4*>(&tile[wave * 256 + 4 * lane]) = cs;
    __syncthreads();
    const float a = (tile[tid] + tile[256 + tid]) + (tile[512 + tid] + tile[768 + tid]);
    cpart[(long long)blockIdx.y * K + c0 + tid] = a;
  }
}

// weights (job table in device memory): logical Wt [rows, K] (row n, k) = trans ? src[k*ld + n] : src[n*ld + k]
// A wave fills ONE destination fragment block per trip (64 chunks of 16 B = 1 KiB, contiguous): lane l packs row
// 32*rb + (l & 31), k-chunk 2*kb + (l >> 5).  (Mapping threads to consecutive source chunks instead scattered the
// 16-byte stores over 64 blocks per wave and paid two 64-bit divisions per chunk: 325 us for DPOT-M's weights.)
template <int NPL>
__global__ __launch_bounds__(256) void bf16_pack_jobs_kernel(const dpot_pack_job* __restrict__ jobs) {
  const dpot_pack_job job = jobs[blockIdx.y];
  const unsigned kb_per_row = (unsigned)job.K >> 4;                   // 16-k blocks per row block
  const unsigned rbs = ((unsigned)job.rows + 31u) >> 5;
  const unsigned nblk = rbs * kb_per_row;
  const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const bool vec = !job.trans && (job.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(job.src) & 15u) == 0;
  uint4* dst = reinterpret_cast<uint4*>(job.dst);
  for (unsigned blk = blockIdx.x * 4u + wv; blk < nblk; blk += gridDim.x * 4u) {
    const unsigned rb = blk / kb_per_row, kb = blk - rb * kb_per_row;
    const unsigned row_raw = 32u * rb + (lane & 31u), kc = 2u * kb + (lane >> 5);
    const bool live = row_raw < (unsigned)job.rows;
    const unsigned row = live ? row_raw : (unsigned)job.rows - 1u;    // clamped address, zero value (padding rows)
    float v[8];
    if (vec) {
      const float4* sp = reinterpret_cast<const float4*>(job.src + (long long)row * job.ld + 8u * kc);
      const float4 a = sp[0], b = sp[1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned k = 8u * kc + e;
        v[e] = job.trans ? job.src[(long long)k * job.ld + row] : job.src[(long long)row * job.ld + k];
      }
    }
    if (!live) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    uint4 o[NPL];
    pack8<NPL>(v, o);
    uint4* d = dst + (long long)blk * NPL * 64 + lane;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) d[pl * 64] = o[pl];
  }
}


// ---- Adam that also emits the bf16 weight packs (round 6) ----------------------------------------
// The channel-MLP weights are 83-89 % of DPOT-M / -L's parameters, and each needs its two packed bf16 forms (W as the B
// operand of the forward product, W^T as that of the data gradient) refreshed after every optimiser step: as a separate
// launch (bf16_pack_jobs_kernel) that is a second read of every weight Adam has just written.  Here a workgroup owns a
// 64 x 256 tile of ONE weight [R, K] (row-major inside the flat parameter buffer): it runs the Adam update on the tile -
// p, g, m, v read and p, m, v written as whole 1 KiB rows, the arithmetic of adam_kernel (common.h adam_update) - keeps
// the new parameters in LDS and writes both packs from there exactly as bf16_pack_both_kernel does (whole 1 KiB pack
// blocks).  The parameters that are not such weights go through adam_ranges_kernel (below).  Bit-identical parameters
// and packs (tests/test_gpu_train2.py::test_adam_writes_the_weight_packs).
__global__ __launch_bounds__(256) void adam_pack_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ hyper, const float* __restrict__ sumsq,
                                                        float grad_scale, const dpot_adam_pack_job* __restrict__ jobs,
                                                        const int* __restrict__ tile_job) {
  __shared__ __attribute__((aligned(16))) float tile[64 * PKB_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const dpot_adam_pack_job job = jobs[tile_job[blockIdx.x]];
  const int t = (int)blockIdx.x - job.tile0;
  const int ctiles = job.K >> 8;
  const int r0 = (t / ctiles) * 64, c0 = (t % ctiles) * 256;
  const int R = job.R, K = job.K;
  const AdamCoef c = adam_coef(hyper, sumsq, grad_scale);
  const long long base = job.off + (long long)r0 * K + c0 + 4 * lane;
  // phase 1: a wave per row, four rows in flight per wave (16 loads outstanding per lane)
#pragma unroll
  for (int i = 0; i < 16; i += 4) {
    float4 pv[4], gv[4], mv[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long o = base + (long long)(wave + 4 * (i + u)) * K;
      pv[u] = *reinterpret_cast<const float4*>(p + o);
      gv[u] = *reinterpret_cast<const float4*>(g + o);
      mv[u] = *reinterpret_cast<const float4*>(m + o);
      vv[u] = *reinterpret_cast<const float4*>(v + o);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = wave + 4 * (i + u);
      const long long o = base + (long long)row * K;
      adam_update(c, pv[u].x, gv[u].x, mv[u].x, vv[u].x);
      adam_update(c, pv[u].y, gv[u].y, mv[u].y, vv[u].y);
      adam_update(c, pv[u].z, gv[u].z, mv[u].z, vv[u].z);
      adam_update(c, pv[u].w, gv[u].w, mv[u].w, vv[u].w);
      *reinterpret_cast<float4*>(m + o) = mv[u];
      *reinterpret_cast<float4*>(v + o) = vv[u];
      *reinterpret_cast<float4*>(p + o) = pv[u];
      *reinterpret_cast<float4*>(&tile[row * PKB_PITCH + 4 * lane]) = pv[u];
    }
  }
  __syncthreads();
  uint4* drow = reinterpret_cast<uint4*>(job.dst_rows);
  uint4* dtr = reinterpret_cast<uint4*>(job.dst_trans);
  if (drow) {   // block (row tile rt, 16-k block kb): chunk l = (row l & 31, features 8 * (l >> 5) .. + 7)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int b = wave + 4 * q, rt = b >> 4, kb = b & 15;
      const float* tp = &tile[(32 * rt + (lane & 31)) * PKB_PITCH + 16 * kb + 8 * (lane >> 5)];
      const float4 x0 = *reinterpret_cast<const float4*>(tp), x1 = *reinterpret_cast<const float4*>(tp + 4);
      drow[((long long)((r0 >> 5) + rt) * (K >> 4) + (c0 >> 4) + kb) * 64 + lane] =
          make_uint4(pack2(x0.x, x0.y), pack2(x0.z, x0.w), pack2(x1.x, x1.y), pack2(x1.z, x1.w));
    }
  }
  if (dtr) {    // block (feature tile ft, 16-row block tb): chunk l = (feature l & 31, rows 8 * (l >> 5) .. + 7)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int b = wave + 4 * q, ft = b >> 2, tb = b & 3;
      const float* tp = &tile[(16 * tb + 8 * (lane >> 5)) * PKB_PITCH + 32 * ft + (lane & 31)];
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = tp[i * PKB_PITCH];
      dtr[((long long)((c0 >> 5) + ft) * (R >> 4) + (r0 >> 4) + tb) * 64 + lane] =
          make_uint4(pack2(x[0], x[1]), pack2(x[2], x[3]), pack2(x[4], x[5]), pack2(x[6], x[7]));
    }
  }
}

// the rest of the flat buffer: blockIdx.y = one range (<= 2^20 elements, the host splits longer runs), blockIdx.x strides it
__global__ __launch_bounds__(256) void adam_ranges_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          const float* __restrict__ hyper, const float* __restrict__ sumsq,
                                                          float grad_scale, const dpot_adam_range* __restrict__ ranges) {
  const dpot_adam_range rg = ranges[blockIdx.y];
  const AdamCoef c = adam_coef(hyper, sumsq, grad_scale);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < rg.len; i += (long long)gridDim.x * 256) {
    const long long o = rg.start + i;
    float pv = p[o], mv = m[o], vv = v[o];
    adam_update(c, pv, g[o], mv, vv);
    m[o] = mv;
    v[o] = vv;
    p[o] = pv;
  }
}

}  // namespace dpot

using namespace dpot;

extern "C" int dpot_gemm_bf16p_supported(int M, int N, int K) {
  return M > 0 && N > 0 && N % 256 == 0 && K >= 32 && K % 32 == 0 ? 1 : 0;
}

extern "C" int64_t dpot_bf16_packed_elems(int rows, int K, int planes) {
  return (int64_t)((rows + 31) / 32) * 32 * K * (planes == 3 ? 3 : 1);
}

extern "C" int dpot_bf16_pack_rows(const float* src, int ld, int rows, int K, int trans, int planes, void* dst,
                                   dpot_stream_t stream) {
  DPOT_REQUIRE(src && dst && rows > 0 && K > 0 && K % 16 == 0 && aligned16(dst), "bf16_pack_rows: bad argument (K %% 16)");
  DPOT_REQUIRE(planes == 1 || planes == 3, "bf16_pack_rows: planes must be 1 (plain bf16) or 3 (bf16x6 split)");
  DPOT_REQUIRE(trans ? ld >= rows : (ld >= K && ld % 4 == 0 && aligned16(src)),
               "bf16_pack_rows: bad leading dimension / alignment");
  const long long nchunks = (long long)((rows + 31) / 32) * 32 * (K >> 3);
  long long g = (nchunks + 255) / 256;
  if (g > 8192) g = 8192;
  if (planes == 3)
    hipLaunchKernelGGL(bf16_pack_rows_kernel<3>, dim3((unsigned)g), dim3(256), 0, as_stream(stream), src, ld, rows, K,
                       reinterpret_cast<uint4*>(dst), nchunks, trans);
  else
    hipLaunchKernelGGL(bf16_pack_rows_kernel<1>, dim3((unsigned)g), dim3(256), 0, as_stream(stream), src, ld, rows, K,
                       reinterpret_cast<uint4*>(dst), nchunks, trans);
  return check_launch("bf16_pack_rows_kernel");
}

extern "C" int dpot_bf16_pack_both_supported(int rows, int K) { return rows > 0 && K > 0 && rows % 64 == 0 && K % 256 == 0; }

extern "C" int dpot_bf16_pack_both(const float* src, int ld, int rows, int K, void* dst_rows, void* dst_trans,
                                   float* colsum_part, dpot_stream_t stream) {
  DPOT_REQUIRE(src && (dst_rows || dst_trans || colsum_part), "bf16_pack_both: null pointer");
  DPOT_REQUIRE(dpot_bf16_pack_both_supported(rows, K) && ld >= K && ld % 4 == 0 && aligned16(src) &&
                   aligned16(dst_rows) && aligned16(dst_trans),
               "bf16_pack_both: needs rows %% 64 == 0, K %% 256 == 0, 16-byte aligned rows");
  DPOT_REQUIRE(rows / 64 <= 65535, "bf16_pack_both: too many rows");
  hipLaunchKernelGGL(bf16_pack_both_kernel, dim3(K / 256, rows / 64), dim3(256), 0, as_stream(stream), src, ld, rows, K,
                     reinterpret_cast<uint4*>(dst_rows), reinterpret_cast<uint4*>(dst_trans), colsum_part,
                     PackNorm{nullptr, nullptr, nullptr, nullptr, 0, 0});
  return check_launch("bf16_pack_both_kernel");
}

extern "C" int dpot_bf16_pack_both_norm(const float* src, int ld, int rows, int K, const float* mean, const float* rstd,
                                        const float* gamma, const float* beta, int rows_per_sample, int G, void* dst_rows,
                                        void* dst_trans, dpot_stream_t stream) {
  DPOT_REQUIRE(src && mean && rstd && gamma && beta && (dst_rows || dst_trans), "bf16_pack_both_norm: null pointer");
  DPOT_REQUIRE(dpot_bf16_pack_both_supported(rows, K) && ld >= K && ld % 4 == 0 && aligned16(src) && aligned16(gamma) &&
                   aligned16(beta) && aligned16(dst_rows) && aligned16(dst_trans),
               "bf16_pack_both_norm: needs rows %% 64 == 0, K %% 256 == 0, 16-byte aligned rows");
  DPOT_REQUIRE(G > 0 && K % G == 0 && (K / G) % 4 == 0 && rows_per_sample > 0 && rows_per_sample % 64 == 0 &&
                   rows % rows_per_sample == 0 && rows / 64 <= 65535,
               "bf16_pack_both_norm: rows_per_sample must be a multiple of 64 dividing rows, K / G a multiple of 4");
  hipLaunchKernelGGL(bf16_pack_both_kernel, dim3(K / 256, rows / 64), dim3(256), 0, as_stream(stream), src, ld, rows, K,
                     reinterpret_cast<uint4*>(dst_rows), reinterpret_cast<uint4*>(dst_trans), (float*)nullptr,
                     PackNorm{mean, rstd, gamma, beta, rows_per_sample, G});
  return check_launch("bf16_pack_both_kernel");
}

extern "C" int dpot_bf16_pack_jobs(const dpot_pack_job* jobs_dev, int njobs, int max_elems, int planes,
                                   dpot_stream_t stream) {
  DPOT_REQUIRE(jobs_dev && njobs > 0 && njobs <= 65535 && max_elems > 0, "bf16_pack_jobs: bad argument");
  DPOT_REQUIRE(planes == 1 || planes == 3, "bf16_pack_jobs: planes must be 1 or 3");
  long long g = ((long long)max_elems / 8 + 255) / 256;
  if (g > 1024) g = 1024;
  if (planes == 3)
    hipLaunchKernelGGL(bf16_pack_jobs_kernel<3>, dim3((unsigned)g, njobs), dim3(256), 0, as_stream(stream), jobs_dev);
  else
    hipLaunchKernelGGL(bf16_pack_jobs_kernel<1>, dim3((unsigned)g, njobs), dim3(256), 0, as_stream(stream), jobs_dev);
  return check_launch("bf16_pack_jobs_kernel");
}

extern "C" int dpot_adam_pack_supported(int rows, int K) { return rows > 0 && K > 0 && rows % 64 == 0 && K % 256 == 0; }

extern "C" int dpot_adam_step_packs(float* p, const float* g, float* m, float* v, const float* hyper, const float* sumsq,
                                    float grad_scale, const dpot_adam_pack_job* jobs_dev, const int32_t* tile_job_dev,
                                    int ntiles, const dpot_adam_range* ranges_dev, int nranges, int max_range_len,
                                    dpot_stream_t stream) {
  DPOT_REQUIRE(p && g && m && v && hyper, "adam_step_packs: null pointer");
  DPOT_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "adam_step_packs: buffers must be 16-byte aligned");
  DPOT_REQUIRE(ntiles >= 0 && nranges >= 0 && nranges <= 65535 && (ntiles == 0 || (jobs_dev && tile_job_dev)) &&
                   (nranges == 0 || (ranges_dev && max_range_len > 0)),
               "adam_step_packs: bad job / range table");
  if (ntiles > 0) {
    hipLaunchKernelGGL(adam_pack_kernel, dim3((unsigned)ntiles), dim3(256), 0, as_stream(stream), p, g, m, v, hyper, sumsq,
                       grad_scale, jobs_dev, reinterpret_cast<const int*>(tile_job_dev));
    int rc = check_launch("adam_pack_kernel");
    if (rc) return rc;
  }
  if (nranges > 0) {
    int gx = (max_range_len + 256 * 16 - 1) / (256 * 16);          // ~16 elements per thread at the longest range
    gx = gx < 1 ? 1 : gx > 256 ? 256 : gx;
    hipLaunchKernelGGL(adam_ranges_kernel, dim3((unsigned)gx, (unsigned)nranges), dim3(256), 0, as_stream(stream), p, g, m, v,
                       hyper, sumsq, grad_scale, ranges_dev);
    return check_launch("adam_ranges_kernel");
  }
  return 0;
}

namespace dpot {
// defined in gemm.hip: fixed-order reduction of split-K partials + epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int batch,
                                                            const EpiArgs e, float* __restrict__ cs_out,
                                                            long long sCs, int csL);
}

// 1 if the ROW-major XCD-contiguous tile order makes the eight L2s fetch fewer operand bytes than the column-major one:
// XCD x holds a contiguous eighth of the enumeration, i.e. (column-major) ~tilesN / 8 W chunks x all their row panels, or
// (row-major) ~tilesM / 8 A panels x all their column chunks; an operand piece is fetched once per XCD that touches it.
// Round 4: the counters showed every bf16 launch moving 4.0-4.7 TB/s through the fabric, 2.1-2.9x its algorithmic bytes.
static int bf16p_row_major(int tilesM, int tilesN, int rowt, int colt) {
  static const int enabled = tune("bf16p_shape", 1);     // 0: none of the shape heuristics (tests: plain column-major order)
  if (!enabled) return 0;
  auto cost = [&](int outer, int inner, double outer_bytes, double inner_bytes) {
    // enumeration with `outer` slowest: an XCD's n / 8 consecutive tiles span ceil-ish (n / 8) / inner + 1 outer values
    const double n = (double)outer * inner, per = n / 8.0;
    double outers = per / inner; if (outers < 1.0) outers = 1.0;
    const double inners = per < inner ? per : inner;
    return 8.0 * (outers * outer_bytes + inners * inner_bytes);
  };
  const double a = 32.0 * rowt, w = 32.0 * colt;              // bytes per k of an A panel / a W chunk (same K: drop it)
  const double col = cost(tilesN, tilesM, w, a), row = cost(tilesM, tilesN, a, w);
  return row < 0.95 * col ? 1 : 0;
}

// B-direct kernels: 32-column tiles per wave - 1 = eight 128 x 32 waves, one workgroup per CU; 2 = four 128 x 64 waves, two
// workgroups per CU once there are >= 2 tiles per CU (the second workgroup then exists); a single round of <= 256..511 tiles
// keeps eight waves per workgroup (four would be ONE wave per SIMD).  Measured (profiles/r04_bf16p_bd_cpw.txt): DPOT-L at
// batch 16, fc1 forward 408 -> 371 us, fc2 data gradient 378 -> 352; DPOT-M's 256-tile launches 74.8 -> 84.2 us the other way
// round.  (DPOT_TUNE bf16p_shape=0: always eight waves)
static int bd_cpw(long long tiles) {
  static const int shape = tune("bf16p_shape", 1);
  return shape && tiles >= 512 ? 2 : 1;
}

// (slabs of look-ahead of the B-direct kernels: 3.  4 / 5 / 7 were built and measured on the theory that the loop is bound by
// load latency / look-ahead: no effect on any DPOT-S / -M / -L shape, profiles/r04_bf16p_bd_lookahead.txt.)

extern "C" int dpot_gemm_bf16p_splitk(int M, int N, int K) {
  // weight-gradient shapes: few output tiles, long K.  Aim at >= 256 workgroups, keep >= 16 slabs (512 k) per split
  const long long tiles = (long long)((M + 127) / 128) * (N / 256);
  const int nslab = K >> 5;
  if (tiles >= 192 || nslab < 32) return 1;
  long long s = (256 + tiles - 1) / tiles;
  const long long smax = nslab / 16;
  if (s > smax) s = smax;
  if (s > 16) s = 16;
  return s < 1 ? 1 : (int)s;
}

// super-block shape (rows x columns of tiles, product 32) for the L2-aware rasterisation, or 0 x 0: the most square
// shape that divides the tile grid; needs more than one round of workgroups to matter.  OPT-IN (DPOT_BF16P_RASTER=1):
// on back-to-back launches of one GEMM it gains 3-6 % (fc1 forward at DPOT-M 148.9 -> 139.4 us,
// profiles/r03_bf16p_train_bench_raster{0,1}.txt), but inside the DPOT-M train step - operands cold, written by the
// previous kernel - the 48 launches average 123.7 us with it against 121.1 us without
// (profiles/r03_step_census_M_bf16_seq_raster{1,0}.txt): the column-major order it replaces streams each weight chunk
// through one XCD's L2 exactly once, which matters more when nothing is cache-resident.
static void bf16p_pick_super(int tilesM, int tilesN, int splits, int* sr, int* sc, bool bdirect) {
  // default: on (both ranges) for the B-direct kernels - round 4, inside the step: DPOT-M 14.36 -> 14.04 ms, DPOT-L 97.8 -> 95.3 ms
  // (profiles/r04_bf16p_bd_raster.txt; the faster main loop feels the 2-4 fetches of every A panel) - off for the LDS-DMA ones
  static const int shape = tune("bf16p_shape", 1);
  const int enabled = shape && bdirect ? 3 : 0;
  *sr = 0; *sc = 0;
  const long long nt = (long long)tilesM * tilesN;
  // 1: the launches with >= 512 tiles (instead of the two-workgroup kernel); 2: only those with one round of 256..511 tiles
  // (the K = 4096 launches of DPOT-M: 64 x 4 tiles - every A row panel is otherwise pulled into four XCDs' L2s; measured:
  // 87.2 against 87.3 us back to back, DPOT-M step 14.64 against 14.68 ms - the L2 fill path is not what bounds them)
  if (!enabled || splits > 1 || nt < 256 || (enabled == 2 && nt >= 512) || (enabled == 1 && nt < 512)) return;
  static const int cand[6][2] = {{8, 4}, {4, 8}, {16, 2}, {2, 16}, {32, 1}, {1, 32}};
  for (int i = 0; i < 6; ++i)
    if (tilesM % cand[i][0] == 0 && tilesN % cand[i][1] == 0) { *sr = cand[i][0]; *sc = cand[i][1]; return; }
}

// kernel selection of dpot_gemm_bf16p in ONE place (round 5): the launch and dpot_gemm_bf16p_kernel_kind (what
// bench.py reports) read the same plan
struct Bf16pPlan {
  int tilesM, tilesN, splits, slabs_per_split, super_r, super_c, colt, cpw;
  bool use_bd;
  unsigned grid;
};
static Bf16pPlan bf16p_plan(int M, int N, int K, int splitk, int planes, bool packs) {
  Bf16pPlan pl;
  pl.tilesM = (M + 32 * PB_ROWT - 1) / (32 * PB_ROWT);
  pl.tilesN = N / (32 * PB_COLT);
  const int nslab = planes == 3 ? K >> 4 : K >> 5;
  pl.splits = splitk > 1 ? splitk : 1;
  pl.slabs_per_split = (nslab + pl.splits - 1) / pl.splits;
  pl.splits = (nslab + pl.slabs_per_split - 1) / pl.slabs_per_split;       // no empty split
  // B-direct form (DPOT_TUNE bf16p_bd=0: off - the LDS-DMA kernels of rounds 2-3 everywhere).  Shape rule: launches with
  // several rounds of tiles or a long contraction; a single round of tiles with K < 2048 (DPOT-S: 8192 x 1024 x 1024, 256
  // tiles, 32 slabs) keeps the LDS-DMA kernel, whose dedicated loader waves start the pipeline sooner - DPOT-S 5.73 -> 5.94 ms
  // with B-direct everywhere, DPOT-M 14.97 -> 14.24, DPOT-L 102.9 -> 96.2 (profiles/r04_bf16p_bd_step_ab_one_box.txt)
  static const int bd = tune("bf16p_bd", 1);
  const bool bd_shape = (long long)pl.tilesM * pl.tilesN * pl.splits >= 512 || pl.slabs_per_split >= 64;
  const bool bd_first = planes == 1 && bd && bd_shape;
  bf16p_pick_super(pl.tilesM, pl.tilesN, pl.splits, &pl.super_r, &pl.super_c, bd_first);
  // 128 x 192 tiles where they fill the rounds of 256 CUs better (as for the pair launch below; a 192-wide tile costs ~0.83
  // of a 256-wide one): DPOT-L at batch 4 has 32 x 6 = 192 tiles of 128 x 256 in fc2 forward / fc1 data gradient - a
  // quarter of the chip idle - and 32 x 8 = 256 of 128 x 192
  static const int allow192 = tune("bf16p_shape", 1);
  pl.colt = PB_COLT;
  if (allow192 && planes == 1 && N % 192 == 0 && pl.super_r == 0) {
    const long long t8 = (long long)pl.tilesM * pl.tilesN * pl.splits, t6 = (long long)pl.tilesM * (N / 192) * pl.splits;
    if (t8 < 512 && 0.83 * (double)((t6 + 255) / 256) < 0.97 * (double)((t8 + 255) / 256)) pl.colt = 6;
  }
  if (pl.colt == 6) pl.tilesN = N / 192;
  pl.grid = (unsigned)(pl.tilesM * pl.tilesN);
  if (pl.super_r > 0) pl.grid = 256u * (unsigned)((pl.tilesM * pl.tilesN / 32 + 7) / 8);
  pl.use_bd = bd_first;
  if (pl.use_bd && pl.super_r == 0) pl.super_c = bf16p_row_major(pl.tilesM, pl.tilesN, PB_ROWT, pl.colt);
  pl.cpw = pl.colt == 6 ? 1 : bd_cpw((long long)pl.tilesM * pl.tilesN * pl.splits);
  // (Round 6 built a 256 x 256 / 256 x 192 "big tile" form - four waves of 128 x 128 accumulators on the unified 512-entry register
  // file, both operands through LDS-DMA, the tile shape hipBLASLt picks for these products - bit-identical to these kernels and
  // SLOWER on every DPOT-M / -L form: fp32-output launches +7..15 % (one wave per SIMD: nothing covers the barrier of every slab),
  // packed-output launches +30..50 % (four waves per CU in the epilogue); DPOT-M 12.42 -> 14.21 ms, DPOT-L 88.0 -> 98.1 ms.
  // profiles/r06_bt_step_ab.txt, r06_gemm_yardstick_bt.txt; the source is in the history: commit 'big-tile bf16 GEMM measured'.)
  return pl;
}
// which kernel dpot_gemm_bf16p runs for a shape: 0 = LDS-DMA (8 compute + 4 loader waves), (1: the "duo" kernel, removed),
// 2 = B-direct with eight 128 x 32 waves, 3 = B-direct with four 128 x 64 waves (two workgroups per CU), 4 = bf16x6;
// + 8: on 128 x 192 tiles
extern "C" int dpot_gemm_bf16p_kernel_kind(int M, int N, int K, int splitk, int planes, int packed_outputs) {
  if (!dpot_gemm_bf16p_supported(M, N, K)) return -1;
  if (planes == 3) return 4;
  const Bf16pPlan pl = bf16p_plan(M, N, K, splitk, planes, packed_outputs != 0);
  const int kind = pl.use_bd ? (pl.cpw == 2 ? 3 : 2) : 0;
  return kind + (pl.colt == 6 ? 8 : 0);
}

extern "C" int dpot_gemm_bf16p(const void* Apacked, const void* Wpacked, const float* bias, const float* aux, int ldaux,
                               const float* res, int ldres, float* pre, int ldpre, float* C, int ldc, int M, int N, int K,
                               int act, int epi_mode, int planes, int splitk, float* workspace, void* out_rows,
                               void* out_trans, float* colsum_part, void* dact_out, const void* dact_in,
                               dpot_stream_t stream) {
  const bool packs = out_rows || out_trans || colsum_part || dact_out || dact_in;
  DPOT_REQUIRE(!dact_out || epi_mode == DPOT_EPI_ACT, "gemm_bf16p: dact_out needs the activation epilogue");
  DPOT_REQUIRE(!dact_in || epi_mode == DPOT_EPI_DACT, "gemm_bf16p: dact_in needs the act' epilogue");
  DPOT_REQUIRE(aligned16(dact_out) && aligned16(dact_in), "gemm_bf16p: derivative packs must be 16-byte aligned");
  DPOT_REQUIRE(!(dact_out || dact_in) || (!pre && !res && N % 32 == 0),
               "gemm_bf16p: the act' pack (fragment order) cannot be combined with a pre-activation save or a residual");
  DPOT_REQUIRE(epi_mode != DPOT_EPI_DACT || dact_in || aux, "gemm_bf16p: the act' epilogue needs dact_in or aux");
  DPOT_REQUIRE(Apacked && Wpacked && (C || packs), "gemm_bf16p: null operand");
  DPOT_REQUIRE(!packs || (planes == 1 && splitk <= 1 && M % 32 == 0 && aligned16(out_rows) && aligned16(out_trans)),
               "gemm_bf16p: packed outputs need planes == 1, no split-K and M %% 32 == 0");
  DPOT_REQUIRE(planes == 1 || planes == 3, "gemm_bf16p: planes must be 1 (plain bf16) or 3 (bf16x6, fp32-accurate)");
  DPOT_REQUIRE(dpot_gemm_bf16p_supported(M, N, K), "gemm_bf16p: unsupported shape M=%d N=%d K=%d (N %% 256, K %% 32)", M, N, K);
  DPOT_REQUIRE(epi_mode == DPOT_EPI_LINEAR || epi_mode == DPOT_EPI_ACT || (epi_mode == DPOT_EPI_DACT && (aux || dact_in)),
               "gemm_bf16p: bad epilogue mode");
  DPOT_REQUIRE(ldc >= N && ldc % 4 == 0 && (!aux || ldaux % 4 == 0) && (!res || ldres % 4 == 0) && (!pre || ldpre % 4 == 0),
               "gemm_bf16p: leading dimensions must be multiples of 4");
  DPOT_REQUIRE(aligned16(Apacked) && aligned16(Wpacked) && aligned16(bias) && aligned16(aux) && aligned16(res) &&
                   aligned16(pre) && aligned16(C),
               "gemm_bf16p: pointers must be 16-byte aligned");
  Bf16pArgs p;
  p.A = reinterpret_cast<const unsigned short*>(Apacked);
  p.W = reinterpret_cast<const unsigned short*>(Wpacked);
  p.M = M; p.N = N; p.K = K;
  p.tilesM = (M + 32 * PB_ROWT - 1) / (32 * PB_ROWT);
  p.tilesN = N / (32 * PB_COLT);
  EpiArgs& e = p.e;
  e.C = C; e.ldc = ldc; e.sC = 0;
  e.bias = bias; e.sBias = 0;
  e.aux = aux; e.ldaux = ldaux; e.sAux = 0;
  e.pre = pre; e.ldpre = ldpre; e.sPre = 0;
  e.res = res; e.ldres = ldres; e.res_div = 0; e.res_mod = 0; e.sRes = 0;
  e.act = act; e.mode = epi_mode; e.accumulate = 0;
  e.M = M; e.N = N;
  const int nslab = planes == 3 ? K >> 4 : K >> 5;     // the bf16x6 kernel steps K in 16-k slabs
  DPOT_REQUIRE(splitk <= 1 || workspace != nullptr, "gemm_bf16p: split-K needs a workspace of splitk*M*N floats");
  DPOT_REQUIRE((splitk > 1 ? splitk : 1) <= nslab && splitk <= 65535, "gemm_bf16p: too many splits");
  p.ws = workspace;
  p.out_rows = reinterpret_cast<uint4*>(out_rows);
  p.out_trans = reinterpret_cast<uint4*>(out_trans);
  p.cs_part = colsum_part;
  p.dact_out = reinterpret_cast<uint4*>(dact_out);
  p.dact_in = reinterpret_cast<const unsigned short*>(dact_in);
  const Bf16pPlan pl = bf16p_plan(M, N, K, splitk, planes, packs);
  p.splits = pl.splits; p.slabs_per_split = pl.slabs_per_split; p.tilesN = pl.tilesN;
  p.super_r = pl.super_r; p.super_c = pl.super_c;
  const int colt = pl.colt, cpw = pl.cpw;
  const bool use_bd = pl.use_bd;
  const unsigned grid = pl.grid;
  if (planes == 3)
    hipLaunchKernelGGL(gemm_bf16x6p_kernel, dim3((unsigned)(p.tilesM * p.tilesN), p.splits), dim3(512), 0,
                       as_stream(stream), p);
  else if (use_bd) {
#define BD_LAUNCH(CT, CW, LA) hipLaunchKernelGGL((gemm_bf16p_bd_kernel<CT, CW, LA>), dim3(grid, p.splits), dim3(64 * CT / CW), 0, as_stream(stream), p)
    if (colt == 6) BD_LAUNCH(6, 1, 3);
    else if (cpw == 2) BD_LAUNCH(8, 2, 3);
    else BD_LAUNCH(8, 1, 3);
#undef BD_LAUNCH
  }
  else if (colt == 6)
    hipLaunchKernelGGL(gemm_bf16p_kernel<6>, dim3(grid, p.splits), dim3(64 * (6 + PB_NLOAD)), 0, as_stream(stream), p);
  else
    hipLaunchKernelGGL(gemm_bf16p_kernel<PB_COLT>, dim3(grid, p.splits), dim3(64 * (8 + PB_NLOAD)), 0, as_stream(stream), p);
  int rc = check_launch("gemm_bf16p_kernel");
  if (rc != DPOT_OK || p.splits == 1) return rc;
  const long long total = (long long)M * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)workspace, p.splits,
                     1, e, (float*)nullptr, 0ll, 0);
  return check_launch("splitk_reduce_kernel");
}

// C0[M0, N0] = A0 W0^T and C1[M1, N1] = A1 W1^T (plain bf16 operands, common K, linear epilogue, no split-K) in one launch
// split-K factor of the pair launch (common to both problems): 0 = do not pair, 1 = pair without split, s > 1 = pair
// with s splits (workspace: s * (M0*N0 + M1*N1) floats; two fixed-order reduce launches follow)
extern "C" int dpot_gemm_bf16p_pair_splitk(int M0, int N0, int M1, int N1, int K) {
  static const int enabled = tune("bf16p_shape", 1);
  if (!enabled || !dpot_gemm_bf16p_supported(M0, N0, K) || !dpot_gemm_bf16p_supported(M1, N1, K)) return 0;
  const long long t0 = (long long)((M0 + 127) / 128) * (N0 / 256), t1 = (long long)((M1 + 127) / 128) * (N1 / 256);
  if (t0 < 192 && t1 < 192) {
    // each alone would be split: together they need half the split factor (half the partial-sum traffic), or none
    if (t0 + t1 >= 192) return 1;
    const int nslab = K >> 5;
    long long s = (256 + t0 + t1 - 1) / (t0 + t1);
    const long long smax = nslab / 16;                 // >= 16 slabs (512 k) per split
    if (s > smax) s = smax;
    if (s > 16) s = 16;
    if (s < 2) return 0;
    const long long sps = (nslab + s - 1) / s;
    s = (nslab + sps - 1) / sps;                       // no empty split
    return s < 2 ? 0 : (int)s;
  }
  // the joint grid needs fewer 256-workgroup rounds (DPOT-L: 288 + 288 tiles = 2 + 2 rounds apart, 3 together)
  const long long r0 = (t0 + 255) / 256, r1 = (t1 + 255) / 256, r01 = (t0 + t1 + 255) / 256;
  return t0 >= 192 && t1 >= 192 && r0 + r1 > r01 ? 1 : 0;
}
extern "C" int dpot_gemm_bf16p_pair_wanted(int M0, int N0, int M1, int N1, int K) {
  return dpot_gemm_bf16p_pair_splitk(M0, N0, M1, N1, K) > 0 ? 1 : 0;
}

// B-direct pair launch: several rounds of tiles or a long contraction (as bf16p_plan)
static bool pair_use_bd(long long grid, int splits, int sps) {
  static const int bd = tune("bf16p_bd", 1);
  return bd && (grid * splits >= 512 || sps >= 128);
}

extern "C" int dpot_gemm_bf16p_pair(const void* A0, const void* W0, float* C0, int ldc0, int M0, int N0, const void* A1,
                                    const void* W1, float* C1, int ldc1, int M1, int N1, int K, int splitk,
                                    float* workspace, dpot_stream_t stream) {
  DPOT_REQUIRE(A0 && W0 && C0 && A1 && W1 && C1, "gemm_bf16p_pair: null operand");
  DPOT_REQUIRE(dpot_gemm_bf16p_supported(M0, N0, K) && dpot_gemm_bf16p_supported(M1, N1, K),
               "gemm_bf16p_pair: unsupported shape (N %% 256, K %% 32)");
  DPOT_REQUIRE(ldc0 >= N0 && ldc1 >= N1 && ldc0 % 4 == 0 && ldc1 % 4 == 0 &&
                   aligned16(A0) && aligned16(W0) && aligned16(C0) && aligned16(A1) && aligned16(W1) && aligned16(C1) &&
                   aligned16(workspace),
               "gemm_bf16p_pair: bad leading dimension / alignment");
  const int nslab = K >> 5;
  int splits = splitk > 1 ? splitk : 1;
  DPOT_REQUIRE(splits == 1 || workspace != nullptr, "gemm_bf16p_pair: split-K needs a workspace of splitk*(M0*N0+M1*N1) floats");
  DPOT_REQUIRE(splits <= nslab && splits <= 65535, "gemm_bf16p_pair: too many splits");
  const int sps = (nslab + splits - 1) / splits;
  DPOT_REQUIRE((nslab + sps - 1) / sps == splits, "gemm_bf16p_pair: splitk would leave an empty split (use dpot_gemm_bf16p_pair_splitk)");
  Bf16pPair pp;
  const void* As[2] = {A0, A1};
  const void* Ws[2] = {W0, W1};
  float* Cs[2] = {C0, C1};
  const int Ms[2] = {M0, M1}, Ns[2] = {N0, N1}, lds_[2] = {ldc0, ldc1};
  float* wss[2] = {workspace, workspace ? workspace + (size_t)splits * M0 * N0 : nullptr};
  for (int i = 0; i < 2; ++i) {
    Bf16pArgs& p = pp.a[i];
    p.A = reinterpret_cast<const unsigned short*>(As[i]);
    p.W = reinterpret_cast<const unsigned short*>(Ws[i]);
    p.M = Ms[i]; p.N = Ns[i]; p.K = K;
    p.tilesM = (Ms[i] + 32 * PB_ROWT - 1) / (32 * PB_ROWT);
    p.tilesN = Ns[i] / (32 * PB_COLT);
    EpiArgs& e = p.e;
    e.C = Cs[i]; e.ldc = lds_[i]; e.sC = 0;
    e.bias = nullptr; e.sBias = 0;
    e.aux = nullptr; e.ldaux = 0; e.sAux = 0;
    e.pre = nullptr; e.ldpre = 0; e.sPre = 0;
    e.res = nullptr; e.ldres = 0; e.res_div = 0; e.res_mod = 0; e.sRes = 0;
    e.act = 0; e.mode = DPOT_EPI_LINEAR; e.accumulate = 0;
    e.M = Ms[i]; e.N = Ns[i];
    p.splits = splits;
    p.slabs_per_split = sps;
    p.ws = wss[i];
    p.out_rows = nullptr; p.out_trans = nullptr; p.cs_part = nullptr;
    p.dact_out = nullptr; p.dact_in = nullptr; p.super_r = 0; p.super_c = 0;
  }
  // 128 x 192 tiles when they need fewer workgroup-rounds' worth of time: DPOT-L's weight gradients are 1536 x 6144 and
  // 6144 x 1536 - 288 + 288 tiles of 128 x 256 = 2.25 rounds of 256 CUs, i.e. THREE rounds; 384 + 384 tiles of 128 x 192 are
  // exactly three rounds of tiles three quarters the size (a 192-wide tile costs ~0.83 of a 256-wide one: 20 instead of
  // 24 KiB of operands per slab, and the loop is bound by those)
  static const int allow192 = tune("bf16p_shape", 1);
  int colt = PB_COLT;
  if (allow192 && splits == 1 && N0 % 192 == 0 && N1 % 192 == 0) {
    const long long t8 = (long long)pp.a[0].tilesM * pp.a[0].tilesN + (long long)pp.a[1].tilesM * pp.a[1].tilesN;
    const long long t6 = (long long)pp.a[0].tilesM * (N0 / 192) + (long long)pp.a[1].tilesM * (N1 / 192);
    const double c8 = (double)((t8 + 255) / 256), c6 = 0.83 * (double)((t6 + 255) / 256);
    if (c6 < 0.97 * c8) colt = 6;
  }
  if (colt == 6) {
    pp.a[0].tilesN = N0 / 192;
    pp.a[1].tilesN = N1 / 192;
  }
  pp.n0 = pp.a[0].tilesM * pp.a[0].tilesN;
  const unsigned grid = (unsigned)(pp.n0 + pp.a[1].tilesM * pp.a[1].tilesN);
  const bool use_bd = pair_use_bd(grid, splits, sps);
  if (use_bd) {
    for (int i = 0; i < 2; ++i) pp.a[i].super_c = bf16p_row_major(pp.a[i].tilesM, pp.a[i].tilesN, PB_ROWT, colt);
    const int cpw = colt == 6 ? 1 : bd_cpw((long long)grid * splits);
#define BD_LAUNCH(CT, CW, LA) hipLaunchKernelGGL((gemm_bf16p_bd_pair_kernel<CT, CW, LA>), dim3(grid, splits), dim3(64 * CT / CW), 0, as_stream(stream), pp)
    if (colt == 6) BD_LAUNCH(6, 1, 3);
    else if (cpw == 2) BD_LAUNCH(8, 2, 3);
    else BD_LAUNCH(8, 1, 3);
#undef BD_LAUNCH
  }
  else if (colt == 6)
    hipLaunchKernelGGL(gemm_bf16p_pair_kernel<6>, dim3(grid, splits), dim3(64 * (6 + PB_NLOAD)), 0, as_stream(stream), pp);
  else
    hipLaunchKernelGGL(gemm_bf16p_pair_kernel<PB_COLT>, dim3(grid, splits), dim3(64 * (8 + PB_NLOAD)), 0, as_stream(stream), pp);
  int rc = check_launch("gemm_bf16p_pair_kernel");
  if (rc != DPOT_OK || splits == 1) return rc;
  for (int i = 0; i < 2; ++i) {
    const long long total = (long long)Ms[i] * Ns[i];
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)wss[i], splits, 1,
                       pp.a[i].e, (float*)nullptr, 0ll, 0);
    rc = check_launch("splitk_reduce_kernel");
    if (rc) return rc;
  }
  return DPOT_OK;
}
