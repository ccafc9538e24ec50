// afno_mlp.hip - the AFNO mixer's block-diagonal complex 2-layer MLP (models/dpot.py:72-94) as ONE kernel:
//
//      Y1 = f(X Wa + ba)          f = act (forward)  or  (.) * act'(aux)  (backward data path, no biases)
//      Y2 = Y1 Wb + bb
//
// X = the spectrum [M = B*mx*my, 2E] in the "planar per channel block" layout: block k owns columns k*N .. (k+1)*N,
// N = 2*bs = [re(bs) | im(bs)], and the complex weights are the real N x N matrices [[Wr, Wi], [-Wi, Wr]] (dft.hip
// afno_pack) - so per block this is two chained real GEMMs with K = N = 2*bs (256 for DPOT-Ti/S/M, 192 for DPOT-L).
// The backward data path  dS = ((dO2 W2^T) * act'(O1pre)) W1^T  is the same chain with transposed weights.
//
// Why one kernel (round 1 ran two launches of the generic 64x64-tile GEMM, 0.44 of the fp32-MFMA roof):
//   * K is only 256: a 64x64 tile is 8 K-slabs long, so prologue / epilogue / wave quantisation (1152 tiles over 256
//     CUs = 4.5 rounds) cost as much as the MFMAs.  Here a workgroup owns a PANEL of R = 16*RT rows x all N columns
//     and runs both layers back to back: 2*N/16 slabs of MFMA work per prologue, the intermediate Y1 never leaves the
//     CU (it is the A operand of layer 2, kept in LDS), one launch.
//   * R is chosen per problem so that the panels fill the 256 CUs evenly (DPOT-Tiny, B=32: 18432 block-rows / 256 CUs
//     = 72 -> R = 80: 232 workgroups, one per CU, 90 % of them busy to the end); v_mfma_f32_16x16x4_f32 (exact fp32,
//     same rate as 32x32x2) gives the 16-row granularity that needs.
//   * operands reach LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction), always as FULL 128-byte
//     lines read by adjacent lanes (a first version fetched "fragment shaped" 16 rows x 64 B per instruction: 64
//     separate 16-byte L1 accesses each - the kernel was bound by the CU's load path at 2x the MFMA time):
//       - weights are pre-packed in FRAGMENT-BLOCK-MAJOR order (dpot_afno_block_weights): the (16 n x 16 k) block of
//         column tile c / K-slab t is 1 KiB contiguous, chunk l = (n = l&15, k = 4*(l>>4)..+3) = the MFMA operand
//         order, so one DMA instruction streams it and one conflict-free ds_read_b128 per lane reads it back;
//       - X rows (the spectrum, row-major) are fetched 32 k at a time: one instruction = 8 rows x 128 B, lane
//         (r = l>>3, c = l&7) fetching chunk c ^ (r>>1) of its row - the XOR swizzle on the SOURCE address makes the
//         lane-linear LDS image [16 rows][128 B] conflict-free for the fragment reads (row, chunk (4h+q) ^ (row>>1)).
//     Two LDS buffers each, one barrier per 16-k slab: wait own DMA (vmcnt 0) -> barrier -> issue slab t+1 (and every
//     other slab the next X super-slab) -> 4*RT*NT MFMAs of slab t.
//   * 4 waves, wave w owns all R rows x columns [16*NT*w, 16*NT*(w+1)): RT x NT accumulators of 16x16 (80 AGPRs at
//     RT=5, NT=4).  Epilogues bounce each 16-row tile through a per-wave LDS slab so that bias / aux loads and all
//     global stores are 16-byte accesses, 64*NT bytes contiguous per row.
#include <type_traits>

#include "common.h"

namespace dpot {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AfnoMlpArgs {
  const float* X;     // [M, ldx]
  const float* Wa;    // [nb][N/16 column tiles][N/16 slabs][256]: fragment-block-major (dpot_afno_block_weights)
  const float* Wb;
  const float* ba;    // [nb][N] or NULL
  const float* bb;
  const float* aux;   // mode 1: pre-activation of the forward [M, ldo]
  float* pre;         // optional: X Wa + ba            [M, ldo]
  float* mid;         // optional: Y1                   [M, ldo]
  float* Y;           // Y2                             [M, ldo]
  int ldx, ldo;
  int M, nb, panels, act, mode;
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16(const float* g, float* l) {
#ifdef DPOT_ABL_NODMA
  return;
#endif
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// NW + 2 waves: waves 0 .. NW-1 compute (wave w owns all R rows x the two 16-column tiles 2w, 2w+1: RT x 2 accumulators
// of 16x16; NW = N/32 = 8 for 2*bs = 256: TWO compute waves per SIMD, which share the matrix pipe during the slabs at no
// loss and overlap each other's LDS / VALU latencies in the epilogues), the last two waves only issue the LDS-DMA.  A DMA piece costs the issuing wave 60-180 cycles of VMEM issue; inside the compute
// waves' instruction stream that time comes straight out of the MFMA pipe (measured: 26 % of the wave cycles parked,
// MFMA pipe 51 % busy), in a separate wave it runs beside the MFMAs.
//
// One workgroup barrier per 16-k slab; ring of three weight buffers and two X super-slab buffers.  At barrier B_t every
// slab <= t+1 has landed, so a compute wave fetches the fragments of slab t+1 into a second register set WHILE the MFMAs
// of slab t run - no LDS latency and no DMA latency behind a barrier.
//   loader:   [issue slabs 0,1,2, X 0,1]  { B_t; issue slab t+3 (-> ring[t%3]), X super-slab on odd t; vmcnt(newest batch) }
//   compute:  B_0, read frags 0           { (t>0: B_t); read frags t+1; MFMAs of slab t }
// The layer-2 weights continue in the same ring (global slab index NSLAB + u).  The epilogues stage through the ring
// buffer (NSLAB-1)%3: epilogue 1 runs between B_(NSLAB-1) and B_NSLAB, when the loaders fill the two OTHER buffers (slab
// NSLAB+1 -> ring[(NSLAB+1)%3] after B_(NSLAB-2), slab NSLAB+2 -> ring[(NSLAB+2)%3] = ring[(NSLAB-1)%3] only after
// B_NSLAB); epilogue 2 runs after the last fragment read.
template <int RT, int NW, int ACTK>
__global__ __launch_bounds__(64 * (NW + 2)) void afno_mlp2_kernel(const AfnoMlpArgs p) {
  constexpr int NT = 2;             // 16-column tiles per compute wave
  constexpr int N = 32 * NW;        // = K
  constexpr int NCT = 2 * NW;       // 16-column tiles of the panel = K-slabs
  constexpr int NSLAB = NCT;
  constexpr int BFL = NCT * 256;    // floats of one weight slab  (N x 16)
  constexpr int AFL = RT * 512;     // floats of one X super-slab (16*RT rows x 32 k)
  constexpr int NXI = 2 * RT;       // DMA instructions per X super-slab (8 rows x 128 B each)
  constexpr int WCOLS = 16 * NT;    // columns per compute wave
  static_assert(NSLAB % 2 == 0, "slab loop is unrolled by two");
  // ONE shared array (a second LDS object makes hipcc wait vmcnt(0) before every fragment read of a DMA pipeline)
  __shared__ __attribute__((aligned(16))) float lds[3 * BFL + 2 * AFL + RT * NCT * 256];
  float* const Bb = lds;                    // [3][NCT][256]
  float* const Ab = lds + 3 * BFL;          // [2][RT][16 rows][32 k], chunk-swizzled
  float* const Y1 = Ab + 2 * AFL;           // [RT][NCT][256]
  // epilogue staging [4][16][WCOLS] = the ring buffer that is idle while the epilogues run: during epilogue 1 the ring
  // holds the layer-2 slabs 0 and 1 (buffers NSLAB % 3, (NSLAB+1) % 3), the third one last held layer-1 slab NSLAB-1
  float* const stage_all = lds + ((NSLAB - 1) % 3) * BFL;      // [NW][16][WCOLS]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;           // fragment row / k-quad of this lane (MFMA A/B operand layout)

  // XCD-contiguous work-item order: items are (block k, panel) with k major, so a chiplet's L2 holds few blocks' weights
  const int nitems = p.nb * p.panels;
  int item;
  {
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = nitems >> 3, r = nitems & 7;
    item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int kblk = item / p.panels, panel = item - kblk * p.panels;
  const int row0 = panel * (16 * RT);

  auto bar = [&]() __attribute__((always_inline)) {
#ifndef DPOT_ABL_NOBAR   /* timing experiments only (scripts/afno_variants.sh): results are wrong without the barriers */
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
  };

  if (wave >= NW) {
    // ================================ loader waves ================================
    const int L = wave - NW;
    const float* X = p.X + (long long)kblk * N;
    const float* Wa_l = p.Wa + (long long)kblk * N * N + lane * 4;
    const float* Wb_l = p.Wb + (long long)kblk * N * N + lane * 4;
    // weights: block (c, t) is 1 KiB contiguous in global memory, lane l fetches its chunk l
    auto issue_w = [&](const float* __restrict__ Wl, int t, float* dstbuf) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < NCT / 2; ++n) {
        const int c = L + 2 * n;
        glds16(Wl + (c * NSLAB + t) * 256, dstbuf + c * 256);
      }
    };
    // X: instruction q of a super-slab = rows 8q .. 8q+7 of the panel, lane (r = l>>3, c = l&7) fetches 16-byte
    // chunk c ^ ((row_in_tile>>1)&7) of its row
    long long xoff[(NXI + 1) / 2];
#pragma unroll
    for (int n = 0; n < (NXI + 1) / 2; ++n) {
      const int q = L + 2 * n;
      const int rt = (8 * q + (lane >> 3)) & 15;        // row inside its 16-row tile
      int row = row0 + 8 * q + (lane >> 3);
      row = row < p.M ? row : p.M - 1;                  // clamped: rows past M only feed outputs that are never stored
      xoff[n] = (long long)row * p.ldx + 4 * ((lane & 7) ^ ((rt >> 1) & 7));
    }
    auto issue_x = [&](int T, float* dstbuf) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < (NXI + 1) / 2; ++n) {
        const int q = L + 2 * n;
        if (q < NXI) glds16(X + xoff[n] + 32 * T, dstbuf + q * 256);      // wave-uniform predicate
      }
    };
    // The loaders run THREE slabs ahead: after barrier B_g they issue slab g+3 into ring[(g+3)%3] = the buffer of slab
    // g, whose fragments the compute waves already hold in registers (read during iteration g-1).  Before the next
    // barrier they wait for everything EXCEPT that newest batch (counted vmcnt): a batch has two slab periods (~2 us) to
    // land - one period is about the DMA's issue-to-landed latency, and waiting for the newest batch made the loaders
    // the last wave at every barrier (measured: 6.8 of 56 us).
    constexpr int WB = NCT / 2, XB = RT;               // pieces per batch and loader wave: weight slab, X super-slab
    auto issue_slab = [&](int g, float* dst) __attribute__((always_inline)) {       // global slab index g
      if (g < NSLAB) issue_w(Wa_l, g, dst); else issue_w(Wb_l, g - NSLAB, dst);
    };
    issue_x(0, Ab);
    issue_slab(0, Bb);
    issue_slab(1, Bb + BFL);
    if (NSLAB / 2 > 1) issue_x(1, Ab + AFL);           // newest batch of the prologue: X super-slab 1 + slab 2
    issue_slab(2, Bb + 2 * BFL);
    if (NSLAB / 2 > 1) wait_vm<WB + XB>(); else wait_vm<WB>();
    bar();                                              // P: slabs 0, 1 and X super-slab 0 have landed
    int ring = 0;                                       // ring buffer of global slab index g + 3 (= g % 3)
#pragma unroll 1
    for (int g = 0; g < 2 * NSLAB; ++g) {               // g = global slab index (layer 1: 0..NSLAB-1, layer 2: the rest)
      bar();                                            // B_g  (g == NSLAB: the barrier after epilogue 1)
      const int nx = g + 3;
      const int T1 = (g + 3) >> 1;                      // X super-slab first needed by slab g+3 (g odd)
      // after B_(NSLAB-1) the target ring[(NSLAB+2)%3] is epilogue 1's staging area: that slab goes out one barrier late
      const bool has_w = nx < 2 * NSLAB && g != NSLAB - 1;
      const bool has_x = (g & 1) && g + 3 < NSLAB && T1 >= 2;
      if (has_x) issue_x(T1, Ab + (T1 & 1) * AFL);
      if (g == NSLAB) issue_slab(nx - 1, Bb + (ring == 0 ? 2 : ring - 1) * BFL);
      if (has_w) issue_slab(nx, Bb + ring * BFL);
      ring = ring == 2 ? 0 : ring + 1;
      if (has_w && has_x) wait_vm<WB + XB>();
      else if (has_w) wait_vm<WB>();
      else wait_vm<0>();
    }
    bar();                                              // S2: end of layer 2
    return;
  }

  // ================================ compute waves ================================
  f32x4 acc[RT][NT];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto mma = [&](const f32x4 (&af)[RT], const f32x4 (&bf)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
  };
  // weight fragments of global slab g (ring buffer g % 3), this wave's NT column tiles
  auto read_w = [&](f32x4 (&bf)[NT], int ringbuf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
      bf[j] = *reinterpret_cast<const f32x4*>(Bb + ringbuf * BFL + (wave * NT + j) * 256 + lane * 4);
  };
  // X fragments of layer-1 slab t: super-slab t/2, 16-k half t&1, chunk (4h + q) ^ (row>>1) of row fr
  const int xfrag0 = fr * 32 + 4 * ((fq) ^ ((fr >> 1) & 7));
  const int xfrag1 = fr * 32 + 4 * ((4 + fq) ^ ((fr >> 1) & 7));
  auto read_x = [&](f32x4 (&af)[RT], int t) __attribute__((always_inline)) {
    const float* xs = Ab + ((t >> 1) & 1) * AFL + ((t & 1) ? xfrag1 : xfrag0);
#pragma unroll
    for (int i = 0; i < RT; ++i) af[i] = *reinterpret_cast<const f32x4*>(xs + i * 512);
  };
  // layer-2 A fragments from Y1: K-slab u, chunk (k-quad, row rotated by u)
  auto read_y = [&](f32x4 (&af)[RT], int u) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RT; ++i)
      af[i] = *reinterpret_cast<const f32x4*>(Y1 + ((i * NCT + u) * 64 + fq * 16 + ((fr + u + 8 * (fq >> 1)) & 15)) * 4);
  };

  // ---- epilogue of one layer: acc (+bias) -> [pre] -> f -> [mid] -> Y1 in LDS (layer 1) / -> Y (layer 2)
  float* const stage = stage_all + wave * (16 * WCOLS);
  auto epilogue = [&](bool first, const float* __restrict__ bias) __attribute__((always_inline)) {
    const int colw = 16 * NT * wave;                   // first column of this wave inside the block
#ifdef DPOT_ABL_NOEPI
    if (p.M > 0) {
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (sacc == 12345.678f) p.Y[lane] = sacc;
      return;
    }
#endif
    // bias of the chunk this lane handles in pass `it` (the same for every row tile): loaded ONCE - a load inside the
    // chunk loop puts a full L2 round trip in front of every 16 bytes of epilogue (measured: 23 of 70 us)
    float4 b4[NT];
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int g = it * 64 + lane;
      const int c4 = g % (4 * NT);
      b4[it] = bias ? *reinterpret_cast<const float4*>(bias + kblk * N + colw + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // one row tile; the tile index is a compile-time constant (a dynamic index into acc would send the accumulators
    // to scratch, and hipcc declines to unroll a loop over a body of this size)
    auto tile = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::value;
      // C/D layout of a 16x16 tile: col = lane&15, row = 4*(lane>>4) + e
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) stage[(4 * fq + e) * WCOLS + 16 * j + fr] = acc[i][j][e];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int it = 0; it < NT; ++it) {
        const int g = it * 64 + lane;                  // 16-byte chunk of the 16 x WCOLS tile, row-major
        const int r = g / (4 * NT), c4 = g - r * (4 * NT);
        const f32x4 t = *reinterpret_cast<const f32x4*>(stage + g * 4);
        float v[4] = {t[0], t[1], t[2], t[3]};
        const int row = row0 + 16 * i + r;
        const int col = colw + 4 * c4;                 // column inside the block
        const bool ok = row < p.M;
        const int rowc = ok ? row : p.M - 1;
        const long long go = (long long)rowc * p.ldo + (long long)kblk * N + col;
        v[0] += b4[it].x; v[1] += b4[it].y; v[2] += b4[it].z; v[3] += b4[it].w;
        if (first) {
          if (p.mode == 0) {
            if (p.pre && ok) *reinterpret_cast<float4*>(p.pre + go) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ACTK == DPOT_ACT_GELU ? gelu_fwd(v[e]) : act_fwd(p.act, v[e]);
          } else {
            const float4 x4 = *reinterpret_cast<const float4*>(p.aux + go);
            const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
            if (p.pre) {
              // the backward also re-derives the ACTIVATED layer-1 output act(aux) (operand of the layer-2 weight gradient):
              // the forward then need not store it (value and derivative share the Gaussian tail evaluation)
              float a[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (ACTK == DPOT_ACT_GELU) {
                  float d;
                  gelu_val_der(xs[e], a[e], d);
                  v[e] *= d;
                } else {
                  a[e] = act_fwd(p.act, xs[e]);
                  v[e] *= act_bwd(p.act, xs[e]);
                }
              }
              if (ok) *reinterpret_cast<float4*>(p.pre + go) = make_float4(a[0], a[1], a[2], a[3]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= ACTK == DPOT_ACT_GELU ? gelu_bwd(xs[e]) : act_bwd(p.act, xs[e]);
            }
          }
          if (p.mid && ok) *reinterpret_cast<float4*>(p.mid + go) = make_float4(v[0], v[1], v[2], v[3]);
          // A operand of layer 2: K-slab s = col/16, chunk (k-quad, row) at a row position rotated by s + 8*(kq/2):
          // the fragment reads (16 rows x k-quads {0,1} or {2,3} per lane group) stay conflict free and the 8 lanes of a
          // row here spread over 4 bank groups instead of 2
          const int s = col >> 4, kq = (col >> 2) & 3;
          *reinterpret_cast<f32x4*>(Y1 + ((i * NCT + s) * 64 + kq * 16 + ((r + s + 8 * (kq >> 1)) & 15)) * 4) =
              (f32x4){v[0], v[1], v[2], v[3]};
        } else {
          if (ok) *reinterpret_cast<float4*>(p.Y + go) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    };
    tile(std::integral_constant<int, 0>{});
    if constexpr (RT > 1) tile(std::integral_constant<int, 1>{});
    if constexpr (RT > 2) tile(std::integral_constant<int, 2>{});
    if constexpr (RT > 3) tile(std::integral_constant<int, 3>{});
    if constexpr (RT > 4) tile(std::integral_constant<int, 4>{});
  };

  f32x4 afA[RT], bfA[NT], afB[RT], bfB[NT];            // two fragment sets: slab t in use, slab t+1 arriving
  // ================= layer 1:  acc = X[panel, :] Wa  (global slabs 0 .. NSLAB-1) =================
  zero_acc();
  bar();                                               // P: slabs 0 and 1 and X super-slab 0 have landed
  read_x(afA, 0);
  read_w(bfA, 0);
  {
    int r1 = 1, r2 = 2;                                // ring buffers of slabs t+1, t+2
#pragma unroll 1
    for (int t = 0; t < NSLAB; t += 2) {
      bar();                                           // B_t
#ifndef DPOT_ABL_NOLDS
      read_x(afB, t + 1);
      read_w(bfB, r1);
#endif
      mma(afA, bfA);
      bar();                                           // B_(t+1)
#ifndef DPOT_ABL_NOLDS
      if (t + 2 < NSLAB) read_x(afA, t + 2);
      read_w(bfA, r2);                                 // t + 2 == NSLAB: the first weight slab of layer 2
      mma(afB, bfB);
#else
      mma(afA, bfA);
#endif
      r1 = r1 == 0 ? 2 : r1 - 1;                       // (r + 2) % 3 == (r - 1) % 3
      r2 = r2 == 0 ? 2 : r2 - 1;
    }
  }
  epilogue(true, p.ba);                                // staging = the ring buffer of slab NSLAB-1 (read at t = NSLAB-2)
  bar();                                               // B_NSLAB (S1): Y1 complete

  // ================= layer 2:  acc = Y1 Wb  (global slabs NSLAB .. 2*NSLAB-1) =================
  zero_acc();
  {
    int r1 = (NSLAB + 1) % 3, r2 = (NSLAB + 2) % 3;
    read_y(afA, 0);                                    // (bfA: fetched during the last slab of layer 1)
#pragma unroll 1
    for (int u = 0; u < NSLAB; u += 2) {
      if (u > 0) bar();                                // B_(NSLAB+u)
#ifndef DPOT_ABL_NOLDS
      read_y(afB, u + 1);
      read_w(bfB, r1);
#endif
      mma(afA, bfA);
      bar();                                           // B_(NSLAB+u+1)
#ifndef DPOT_ABL_NOLDS
      if (u + 2 < NSLAB) {
        read_y(afA, u + 2);
        read_w(bfA, r2);
      }
      mma(afB, bfB);
#else
      mma(afA, bfA);
#endif
      r1 = r1 == 0 ? 2 : r1 - 1;
      r2 = r2 == 0 ? 2 : r2 - 1;
    }
  }
  bar();                                               // S2: every wave is done with Y1 and the ring
  epilogue(false, p.bb);
}

// =====================================================================================================================
// The same two-layer complex MLP with THREE real products per complex product instead of four (Gauss / Karatsuba):
//      (Sr + i Si)(Wr + i Wi):   P1 = Sr Wr,  P2 = Si Wi,  P3 = (Sr + Si)(Wr + Wi)   ->   re = P1 - P2,  im = P3 - P1 - P2
// 25 % fewer MFMAs (and 25 % fewer weight bytes); bs = 128 (DPOT-Ti / S / M), 96 (DPOT-L) or 64.  The backward data path is the same
// complex product with (Wr^T, -Wi^T), so only the packed weights differ.  Same machinery as afno_mlp2_kernel (panels,
// loader waves, LDS-DMA, ring of three slabs, one barrier per 16-k slab, Y1 in LDS) with these differences:
//   * K runs over bs = 128 (8 slabs per layer); a slab holds the Wr and Wi fragments (16 tiles = 16 KiB) and - layer 1 -
//     the Sr and Si pieces of the panel (16 rows x 64 B per DMA instruction, XOR-swizzled on the source address);
//   * compute wave w owns column tile w (16 columns) of P1, P2 AND P3: 3 * RT accumulators, recombined in the epilogue
//     into 16 real + 16 imaginary output columns - no cross-wave traffic;
//   * per slab: G1 = P1 += a_r b_r, G2 = P2 += a_i b_i, a_s = a_r + a_i and b_s = b_r + b_i on the VALU, then the
//     fragments of the next slab are fetched into the freed registers under G3 = P3 += a_s b_s  (60 MFMAs per slab
//     and wave at RT = 5).
// Rounding: the imaginary part is a difference of three fp32 sums instead of a sum of two - error bound ~2x the direct
// form's, measured ~1e-6 relative on DPOT data (tests: same tolerances as the four-product kernel).
// Packed weights (dpot_afno_pack_all, layout 1): [block][slab t][part][col tile c][4*l + e] =
//      W_part[k = 16t + 4(l>>4) + e][n = 16c + (l&15)],   forward: (Wr, Wi);  backward: (Wr^T, -Wi^T).
// =====================================================================================================================
// SPLIT (round 5; bs = 96, RT >= 2): six column tiles on EIGHT compute waves.  With one wave per column tile the six waves land
// 2 + 2 + 1 + 1 on the four SIMDs (a workgroup's waves go to the SIMDs in the order 0, 2, 1, 3) and the matrix pipes top out at
// 6 / 8; here waves 0-3 own all RT row tiles of column tiles 0-3, waves 4 / 5 the first ceil(RT / 2) row tiles of column tiles 4 / 5
// and waves 6 / 7 their remaining row tiles - per SIMD RT + ceil(RT/2), RT + ceil(RT/2), RT + floor(RT/2), RT + floor(RT/2) row
// tiles (8 : 8 : 7 : 7 at RT = 5 instead of 10 : 10 : 5 : 5).  Every wave still owns exactly ONE column tile: same code, its
// row range [r0, r0 + NR) a compile-time size.  The loader waves become waves 8 and 9.
#ifndef AFNO_XSW
#define AFNO_XSW 2      // XOR mask of the X-slab chunk swizzle (3 = the rounds 2-5 image, for A/B builds: scripts/variant.sh)
#endif
template <int RT, int BS, int ACTK, bool SPLIT = false>
__global__ __launch_bounds__(64 * ((SPLIT ? 8 : BS / 16) + 2)) void afno_mlp3_kernel(const AfnoMlpArgs p) {
  constexpr int NW = BS / 16;       // 16-column tiles of one part (8 for bs = 128, 6 for bs = 96) = compute waves unless SPLIT
  constexpr int NCW = SPLIT ? 8 : NW;   // compute waves
  static_assert(!SPLIT || (NW == 6 && RT >= 2), "the split form is for six column tiles");
  constexpr int N = 2 * BS;
  constexpr int NSLAB = BS / 16;    // 16-k slabs per layer
  constexpr int NCT = 2 * NSLAB;    // 16-column tiles of Y1 (both parts)
  constexpr int WSL = 2 * NW * 256; // floats of a weight slab: [part][c][256]
  constexpr int XSL = 2 * RT * 256; // floats of an X slab: [part][row tile][16 rows][16 k]
  constexpr int WB = NW, XB = RT;   // DMA pieces per slab and loader wave
  __shared__ __attribute__((aligned(16))) float lds[3 * WSL + 3 * XSL + RT * NCT * 256 + 256 + (SPLIT ? NCW * 512 : 0)];
  float* const Wr_ = lds;                   // ring [3][part][c][256]
  float* const Xb = lds + 3 * WSL;          // ring [3][part][RT][256]
  float* const Y1 = Xb + 3 * XSL;           // [RT][NCT][256]
  float* const sink = Y1 + RT * NCT * 256;  // 1 KiB nobody reads: target of the aux prefetch DMA
  // epilogue staging [compute waves][16][32]: one weight slab (NW waves); the split form's eight waves get a region of their own
  float* const stage_all = SPLIT ? sink + 256 : lds + ((NSLAB - 1) % 3) * WSL;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  const int nitems = p.nb * p.panels;
  int item;
  {
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = nitems >> 3, r = nitems & 7;
    item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int kblk = item / p.panels, panel = item - kblk * p.panels;
  const int row0 = panel * (16 * RT);

  auto bar = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (wave >= NCW) {
    // ================================ loader waves ================================
    const int L = wave - NCW;
    const float* X = p.X + (long long)kblk * N;
    const float* Wa_l = p.Wa + (long long)kblk * (NSLAB * WSL) + lane * 4;
    const float* Wb_l = p.Wb + (long long)kblk * (NSLAB * WSL) + lane * 4;
    auto issue_w = [&](int g, float* dstbuf) __attribute__((always_inline)) {   // global slab index g (layer 2: g >= NSLAB)
      const float* Wl = g < NSLAB ? Wa_l + g * WSL : Wb_l + (g - NSLAB) * WSL;
#pragma unroll
      for (int n = 0; n < WB; ++n) {
        const int pc = L + 2 * n;
        glds16(Wl + pc * 256, dstbuf + pc * 256);
      }
    };
    // X piece q = (part = q / RT, row tile i = q % RT): 16 rows x 64 B; lane (r = l >> 2, c = l & 3) fetches the 16-byte
    // chunk c ^ ((r >> 2) & 2) of its row.  (Rounds 2-5 XORed with (r >> 2) & 3, derived for CONTIGUOUS 16-lane groups; a
    // ds_read_b128 is served in the groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS), in which that
    // image put two lanes on every 16-byte slot: SQ_LDS_BANK_CONFLICT = 0.49 of the LDS cycles, profiles/r06_pmc_mixer.json.
    // With (r >> 2) & 2 the 16 lanes of every group read 16 different slots.)
    long long xoff[XB];
#pragma unroll
    for (int n = 0; n < XB; ++n) {
      const int q = L + 2 * n;
      const int part = q / RT, i = q - part * RT;
      const int r = lane >> 2;
      int row = row0 + 16 * i + r;
      row = row < p.M ? row : p.M - 1;                  // clamped: rows past M only feed outputs that are never stored
      xoff[n] = (long long)row * p.ldx + part * BS + 4 * ((lane & 3) ^ ((r >> 2) & AFNO_XSW));
    }
    auto issue_x = [&](int t, float* dstbuf) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < XB; ++n) glds16(X + xoff[n] + 16 * t, dstbuf + (L + 2 * n) * 256);
    };
    // three slabs ahead; before a barrier everything except the newest batch has landed (see afno_mlp2_kernel)
    issue_x(0, Xb);
    issue_w(0, Wr_);
    issue_x(1, Xb + XSL);
    issue_w(1, Wr_ + WSL);
    issue_x(2, Xb + 2 * XSL);
    issue_w(2, Wr_ + 2 * WSL);
    wait_vm<WB + XB>();
    bar();                                              // P: slabs 0 and 1 have landed
    int ring = 0;                                       // ring buffer of global slab g + 3 (= g % 3)
#pragma unroll 1
    for (int g = 0; g < 2 * NSLAB; ++g) {
      bar();                                            // B_g  (g == NSLAB: the barrier after epilogue 1)
      const int nx = g + 3;
      // after B_(NSLAB-1) the target ring[(NSLAB+2)%3] is epilogue 1's staging area: that slab goes out one barrier late
      const bool has_w = nx < 2 * NSLAB && g != NSLAB - 1;
      const bool has_x = nx < NSLAB;
      if (has_x) issue_x(nx, Xb + ring * XSL);
      if (g == NSLAB) issue_w(nx - 1, Wr_ + (ring == 0 ? 2 : ring - 1) * WSL);
      if (has_w) issue_w(nx, Wr_ + ring * WSL);
      ring = ring == 2 ? 0 : ring + 1;
      if (has_w && has_x) wait_vm<WB + XB>();
      else if (has_w) wait_vm<WB>();
      else wait_vm<0>();
    }
    bar();                                              // S2
    return;
  }

  // ================================ compute waves ================================
  // NR row tiles [r0, r0 + NR) of column tile cw (NR == RT, r0 == 0, cw == wave unless SPLIT)
  auto compute = [&](auto NRc, const int r0, const int cw) __attribute__((always_inline)) {
  constexpr int NR = decltype(NRc)::value;
  f32x4 P1[NR], P2[NR], P3[NR];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NR; ++i) P1[i] = P2[i] = P3[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  // weight fragments of this wave's column tile: (b_r, b_i) of the slab in ring buffer `rb`
  auto read_w = [&](f32x4 (&b)[2], int rb) __attribute__((always_inline)) {
    b[0] = *reinterpret_cast<const f32x4*>(Wr_ + rb * WSL + cw * 256 + lane * 4);
    b[1] = *reinterpret_cast<const f32x4*>(Wr_ + rb * WSL + (NW + cw) * 256 + lane * 4);
  };
  const int xfrag = fr * 16 + 4 * (fq ^ ((fr >> 2) & AFNO_XSW));
  auto read_x = [&](f32x4 (&ar)[NR], f32x4 (&ai)[NR], int rb) __attribute__((always_inline)) {
    const float* xs = Xb + rb * XSL + xfrag;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      ar[i] = *reinterpret_cast<const f32x4*>(xs + (r0 + i) * 256);
      ai[i] = *reinterpret_cast<const f32x4*>(xs + (RT + r0 + i) * 256);
    }
  };
  auto read_y = [&](f32x4 (&ar)[NR], f32x4 (&ai)[NR], int u) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      ar[i] = *reinterpret_cast<const f32x4*>(Y1 + (((r0 + i) * NCT + u) * 64 + fq * 16 + ((fr + u + 8 * (fq >> 1)) & 15)) * 4);
      ai[i] = *reinterpret_cast<const f32x4*>(Y1 + (((r0 + i) * NCT + u + NSLAB) * 64 + fq * 16 + ((fr + u + NSLAB + 8 * (fq >> 1)) & 15)) * 4);
    }
  };
  auto g12 = [&](const f32x4 (&ar)[NR], const f32x4 (&ai)[NR], const f32x4 (&b)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        P1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i][s2], b[0][s2], P1[i], 0, 0, 0);
        P2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[i][s2], b[1][s2], P2[i], 0, 0, 0);
      }
  };
  auto g3 = [&](const f32x4 (&as)[NR], const f32x4& bs) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) P3[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[i][s2], bs[s2], P3[i], 0, 0, 0);
  };

  // ---- epilogue: (P1, P2, P3) -> 16 real + 16 imaginary columns (+bias) -> [pre] -> f -> [mid] -> Y1 (layer 1) / Y
  float* const stage = stage_all + wave * 512;          // [16 rows][32]: columns 0-15 real, 16-31 imaginary
  auto epilogue = [&](bool first, const float* __restrict__ bias) __attribute__((always_inline)) {
    // compiler barrier: without it the bias loads and the epilogue's address arithmetic are hoisted above the slab loop,
    // where every register is taken, and spilled across it (17 MB of scratch traffic per launch in the L2 counters)
    // the epilogue's index arithmetic hangs off an OPAQUE copy of the lane id: everything derived from it is computed
    // here, after the slab loops (hoisted above them - where every register is taken - it was spilled across them)
    const long long gbase = (long long)row0 * p.ldo + (long long)kblk * N;       // uniform
    float* const pre_b = p.pre ? p.pre + gbase : nullptr;
    float* const mid_b = p.mid ? p.mid + gbase : nullptr;
    const float* const aux_b = p.aux ? p.aux + gbase : nullptr;
    float* const Y_b = p.Y + gbase;
    int ln = lane;
    asm volatile("" : "+v"(ln)::"memory");
    const int efr = ln & 15, efq = ln >> 4;
    // chunk g = it*64 + lane of the 16 x 32 tile: row g / 8, staged columns 4*(g % 8) .. +3 -> block column
    // (part = staged col / 16) * BS + 16*wave + staged col % 16
    float4 b4[2];
    int bcol[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c4 = (it * 64 + ln) & 7;
      bcol[it] = (c4 >> 2) * BS + 16 * cw + 4 * (c4 & 3);
      b4[it] = bias ? *reinterpret_cast<const float4*>(bias + kblk * N + bcol[it]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto tile = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::value;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float re = P1[i][e] - P2[i][e];
        const float im = P3[i][e] - P1[i][e] - P2[i][e];
        stage[(4 * efq + e) * 32 + efr] = re;
        stage[(4 * efq + e) * 32 + 16 + efr] = im;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int g = it * 64 + ln;
        const int r = g >> 3;
        const f32x4 t = *reinterpret_cast<const f32x4*>(stage + g * 4);
        float v[4] = {t[0] + b4[it].x, t[1] + b4[it].y, t[2] + b4[it].z, t[3] + b4[it].w};
        const int row = row0 + 16 * (r0 + i) + r;
        const int col = bcol[it];
        const bool ok = row < p.M;
        const int rowc = ok ? row : p.M - 1;
        // 32-bit offset from a workgroup-uniform base (row0, kblk come from blockIdx): no 64-bit per-lane address math
        const int go = (rowc - row0) * p.ldo + col;
        if (first) {
          if (p.mode == 0) {
#ifndef AFNO_NO_NT_PRE   // saved for the backward only: non-temporal, it would only push live tensors out of L2 / Infinity Cache
            if (pre_b && ok) {
              typedef float nt_f4 __attribute__((ext_vector_type(4)));
              const nt_f4 q = {v[0], v[1], v[2], v[3]};
              __builtin_nontemporal_store(q, reinterpret_cast<nt_f4*>(pre_b + go));
            }
#else
            if (pre_b && ok) *reinterpret_cast<float4*>(pre_b + go) = make_float4(v[0], v[1], v[2], v[3]);
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ACTK == DPOT_ACT_GELU ? gelu_fwd(v[e]) : act_fwd(p.act, v[e]);
          } else {
            const float4 x4 = *reinterpret_cast<const float4*>(aux_b + go);
            const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
            if (pre_b) {
              // the backward also re-derives the ACTIVATED layer-1 output act(aux) (operand of the layer-2 weight gradient):
              // the forward then need not store it (value and derivative share the Gaussian tail evaluation)
              float a[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (ACTK == DPOT_ACT_GELU) {
                  float d;
                  gelu_val_der(xs[e], a[e], d);
                  v[e] *= d;
                } else {
                  a[e] = act_fwd(p.act, xs[e]);
                  v[e] *= act_bwd(p.act, xs[e]);
                }
              }
              if (ok) *reinterpret_cast<float4*>(pre_b + go) = make_float4(a[0], a[1], a[2], a[3]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= ACTK == DPOT_ACT_GELU ? gelu_bwd(xs[e]) : act_bwd(p.act, xs[e]);
            }
          }
          if (mid_b && ok) *reinterpret_cast<float4*>(mid_b + go) = make_float4(v[0], v[1], v[2], v[3]);
          const int s = col >> 4, kq = (col >> 2) & 3;
          *reinterpret_cast<f32x4*>(Y1 + (((r0 + i) * NCT + s) * 64 + kq * 16 + ((r + s + 8 * (kq >> 1)) & 15)) * 4) =
              (f32x4){v[0], v[1], v[2], v[3]};
        } else {
          if (ok) *reinterpret_cast<float4*>(Y_b + go) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    };
    tile(std::integral_constant<int, 0>{});
    if constexpr (NR > 1) tile(std::integral_constant<int, 1>{});
    if constexpr (NR > 2) tile(std::integral_constant<int, 2>{});
    if constexpr (NR > 3) tile(std::integral_constant<int, 3>{});
    if constexpr (NR > 4) tile(std::integral_constant<int, 4>{});
  };

  // one slab: G1, G2 on the current fragments, sums on the VALU, next slab's fragments into the freed registers, G3.
  // (B0 is dead once b_s is formed: ONE weight-fragment set.  At RT = 5 the kernel sits at the 168 VGPRs a 10-wave
  // workgroup leaves per wave and the allocator spills ~30 registers per lane AROUND the epilogues - not in the slab
  // loops; forming a_s in place of a_r to save 20 registers made the allocation worse, not better.)
  f32x4 AR[NR], AI[NR], AS[NR], B0[2];
  auto sums = [&](const f32x4 (&b)[2], f32x4& bs) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NR; ++i) AS[i] = AR[i] + AI[i];
    bs = b[0] + b[1];
  };

  // backward data path: epilogue 1 multiplies by act'(aux), and aux (the forward's pre-activation, written a whole
  // forward + half a backward ago) comes from HBM.  Pull this wave's part of it towards L2 now: LDS-DMA into a sink
  // nobody reads (no registers, no wait; the compute waves have no counted vmcnt).
  if (p.mode == 1) {
    // 16 rows x (64 B real + 64 B imaginary) per row tile = 2 instructions: lane -> (row, part, 16-byte chunk)
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int id = h2 * 64 + lane;
        int row = row0 + 16 * (r0 + i) + (id >> 3);
        row = row < p.M ? row : p.M - 1;
        const int part = (id >> 2) & 1, c = id & 3;
        glds16(p.aux + (long long)row * p.ldo + (long long)kblk * N + part * BS + 16 * cw + 4 * c, sink);
      }
  }

  // ================= layer 1 =================
  zero_acc();
  bar();                                               // P
  read_x(AR, AI, 0);
  read_w(B0, 0);
  {
    int r1 = 1, r2 = 2;                                // ring buffers of slabs t+1, t+2
#pragma unroll
    for (int t = 0; t < NSLAB; t += 2) {
      f32x4 bs;
      bar();                                           // B_t: slab t+1 has landed
      g12(AR, AI, B0);
      sums(B0, bs);
      read_x(AR, AI, r1);
      read_w(B0, r1);
      g3(AS, bs);
      bar();                                           // B_(t+1)
      g12(AR, AI, B0);
      sums(B0, bs);
      if (t + 2 < NSLAB) read_x(AR, AI, r2);
      read_w(B0, r2);                                  // t + 2 == NSLAB: the first weight slab of layer 2
      g3(AS, bs);
      r1 = r1 == 0 ? 2 : r1 - 1;
      r2 = r2 == 0 ? 2 : r2 - 1;
    }
  }
  epilogue(true, p.ba);
  bar();                                               // B_NSLAB (S1): Y1 complete

  // ================= layer 2 =================
  zero_acc();
  {
    int r1 = (NSLAB + 1) % 3, r2 = (NSLAB + 2) % 3;
    read_y(AR, AI, 0);                                 // (B0: fetched during the last slab of layer 1)
#pragma unroll
    for (int u = 0; u < NSLAB; u += 2) {
      f32x4 bs;
      if (u > 0) bar();                                // B_(NSLAB+u)
      g12(AR, AI, B0);
      sums(B0, bs);
      read_y(AR, AI, u + 1);
      read_w(B0, r1);
      g3(AS, bs);
      bar();                                           // B_(NSLAB+u+1)
      g12(AR, AI, B0);
      sums(B0, bs);
      if (u + 2 < NSLAB) {
        read_y(AR, AI, u + 2);
        read_w(B0, r2);
      }
      g3(AS, bs);
      r1 = r1 == 0 ? 2 : r1 - 1;
      r2 = r2 == 0 ? 2 : r2 - 1;
    }
  }
  bar();                                               // S2
  epilogue(false, p.bb);
  };
  if constexpr (SPLIT) {
    constexpr int RA = (RT + 1) / 2, RB = RT - RA;          // row tiles of column tiles 4 / 5 on waves 4 / 5 and on waves 6 / 7
    if (wave < 4) compute(std::integral_constant<int, RT>{}, 0, wave);
    else if (wave < 6) compute(std::integral_constant<int, RA>{}, 0, wave);
    else compute(std::integral_constant<int, RB>{}, RA, wave - 2);
  } else {
    compute(std::integral_constant<int, RT>{}, 0, wave);
  }
}

// Wbig [J][N][N] (row-major, as dpot_afno_pack writes it: W[k][n]) -> fragment-block-major copies
//   fwd[j][c][t][4*l + e] = W[k = 16t + 4(l>>4) + e][n = 16c + (l&15)]     (Wt = W^T: the forward multiplies by W)
//   bwd[j][c][t][4*l + e] = W[16c + (l&15)][16t + 4(l>>4) + e]             (Wt = W:   the backward multiplies by W^T)
__global__ __launch_bounds__(256) void afno_block_weights_kernel(const float* __restrict__ W, float* __restrict__ fwd,
                                                                 float* __restrict__ bwd, int N, long long total) {
  const int nct = N / 16;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int e = (int)(idx & 3), l = (int)((idx >> 2) & 63);
    long long blk = idx >> 8;
    const int t = (int)(blk % nct);
    blk /= nct;
    const int c = (int)(blk % nct);
    const long long j = blk / nct;
    const int a = 16 * c + (l & 15), b = 16 * t + 4 * (l >> 4) + e;
    const float* Wj = W + j * (long long)N * N;
    if (fwd) fwd[idx] = Wj[(long long)b * N + a];
    if (bwd) bwd[idx] = Wj[(long long)a * N + b];
  }
}

// Everything the mixer needs from the parameters of ONE AFNO layer, in one pass (job table in DEVICE memory, one job per
// layer): w [2, nb, bs, bs], b [2, nb, bs] (models/dpot.py:45-48) ->
//   wbig [nb][N][N] = [[Wr, Wi], [-Wi, Wr]] (row-major W[k][n]; the generic GEMM fallback and the tests use it),
//   bbig [nb][N] = [br | bi],  fwd / bwd = fragment-block-major W and W^T (see afno_block_weights_kernel)
// layout 1 (three-product kernel, bs = 128): fwd / bwd hold [block][slab t][part][col tile c][4*l + e] =
//   W_part[k = 16t + 4(l>>4) + e][n = 16c + (l&15)],  fwd: (Wr, Wi),  bwd: (Wr^T, -Wi^T)  - half the size of layout 0
__global__ __launch_bounds__(256) void afno_pack_all_kernel(const dpot_afno_pack_job* __restrict__ jobs, int nb, int bs,
                                                            int layout) {
  const dpot_afno_pack_job job = jobs[blockIdx.y];
  const int N = 2 * bs, nct = N / 16;
  if (layout == 1) {
    const int nsl = bs / 16;                              // slabs = column tiles per part
    const long long nP = (long long)nb * 2 * bs * bs;
    const long long plane1 = (long long)nb * bs * bs;
    // one destination quad (4 consecutive k) per thread and trip, 32-bit index arithmetic
    const unsigned nq = (unsigned)(nP >> 2), unsl = (unsigned)nsl;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < nq; q += gridDim.x * 256u) {
      const unsigned l = q & 63u;
      unsigned blk = q >> 6;
      const unsigned b1 = blk / unsl, c = blk - b1 * unsl;
      const unsigned part = b1 & 1u;
      blk = b1 >> 1;
      const unsigned k = blk / unsl, t = blk - k * unsl;
      const unsigned kk = 16u * t + 4u * (l >> 4), n = 16u * c + (l & 15u);
      const float* wp = job.w + part * plane1 + (long long)k * bs * bs;
      if (job.fwd) {
        const float* sp = wp + (long long)kk * bs + n;
        reinterpret_cast<float4*>(job.fwd)[q] = make_float4(sp[0], sp[bs], sp[2 * bs], sp[3 * bs]);
      }
      if (job.bwd) {
        const float4 v = *reinterpret_cast<const float4*>(wp + (long long)n * bs + kk);   // bs % 16 == 0, 16-B aligned
        reinterpret_cast<float4*>(job.bwd)[q] = part ? make_float4(-v.x, -v.y, -v.z, -v.w) : v;
      }
    }
  }
  const long long nW = (job.wbig || layout == 0) ? (long long)nb * N * N : 0;   // layout 1 without the dense copy: nothing to do
  const long long plane = (long long)nb * bs * bs;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < nW; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % N);
    const int r = (int)((idx / N) % N);
    const int k = (int)(idx / ((long long)N * N));
    const int i = r % bs, o = c % bs;
    const bool rim = r >= bs, cim = c >= bs;
    const long long wi = ((long long)k * bs + i) * bs + o;
    float v;
    if (!rim && !cim) v = job.w[wi];
    else if (!rim && cim) v = job.w[plane + wi];
    else if (rim && !cim) v = -job.w[plane + wi];
    else v = job.w[wi];
    if (job.wbig) job.wbig[idx] = v;
    const long long base = (long long)k * N * N;
    if (job.bwd && layout == 0)   // Wt = W: block (row tile r/16, slab c/16), chunk (row r%16, k-quad (c%16)/4)
      job.bwd[base + ((((long long)(r >> 4) * nct + (c >> 4)) * 64 + ((c & 15) >> 2) * 16 + (r & 15)) << 2) + (c & 3)] = v;
    if (job.fwd && layout == 0)   // Wt = W^T: block (row tile c/16, slab r/16), chunk (row c%16, k-quad (r%16)/4)
      job.fwd[base + ((((long long)(c >> 4) * nct + (r >> 4)) * 64 + ((r & 15) >> 2) * 16 + (c & 15)) << 2) + (r & 3)] = v;
  }
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < (long long)nb * N; idx += (long long)gridDim.x * 256) {
    const int ci = (int)(idx % bs), part = (int)((idx / bs) & 1), k = (int)(idx / N);
    job.bbig[idx] = job.b[((long long)part * nb + k) * bs + ci];
  }
}

constexpr int AFNO_NUM_CU = 256;

// rows per panel (multiple of 16, <= 80): fewest rounds of (panels * nb) workgroups over the CUs, one workgroup per CU
static int pick_rt(int M, int nb) {
  long long best_cost = -1;
  int best = 5;
  for (int rt = 5; rt >= 1; --rt) {
    const long long panels = (M + 16 * rt - 1) / (16 * rt);
    const long long rounds = (panels * nb + AFNO_NUM_CU - 1) / AFNO_NUM_CU;
    // one round costs the panel's MFMA time plus a fixed prologue / epilogue share (~12 rows' worth)
    const long long cost = rounds * (16 * rt + 12);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = rt;
    }
  }
  return best;
}

template <int NW, int ACTK>
static int launch_rt(const AfnoMlpArgs& p, int rt, hipStream_t s) {
  const dim3 grid((unsigned)(p.nb * p.panels)), blk(64 * (NW + 2));
  switch (rt) {
    case 1: hipLaunchKernelGGL((afno_mlp2_kernel<1, NW, ACTK>), grid, blk, 0, s, p); break;
    case 2: hipLaunchKernelGGL((afno_mlp2_kernel<2, NW, ACTK>), grid, blk, 0, s, p); break;
    case 3: hipLaunchKernelGGL((afno_mlp2_kernel<3, NW, ACTK>), grid, blk, 0, s, p); break;
    case 4: hipLaunchKernelGGL((afno_mlp2_kernel<4, NW, ACTK>), grid, blk, 0, s, p); break;
    default: hipLaunchKernelGGL((afno_mlp2_kernel<5, NW, ACTK>), grid, blk, 0, s, p); break;
  }
  return check_launch("afno_mlp2_kernel");
}
template <int BS, int ACTK>
static int launch3_rt(const AfnoMlpArgs& p, int rt, hipStream_t s) {
  const dim3 grid((unsigned)(p.nb * p.panels)), blk(64 * (BS / 16 + 2));
  if constexpr (BS == 96) {
    // six column tiles on eight compute waves (afno_mlp3_kernel<.., SPLIT>)
    if (rt >= 2) {
      const dim3 blk8(64 * 10);
      switch (rt) {
        case 2: hipLaunchKernelGGL((afno_mlp3_kernel<2, BS, ACTK, true>), grid, blk8, 0, s, p); break;
        case 3: hipLaunchKernelGGL((afno_mlp3_kernel<3, BS, ACTK, true>), grid, blk8, 0, s, p); break;
        case 4: hipLaunchKernelGGL((afno_mlp3_kernel<4, BS, ACTK, true>), grid, blk8, 0, s, p); break;
        default: hipLaunchKernelGGL((afno_mlp3_kernel<5, BS, ACTK, true>), grid, blk8, 0, s, p); break;
      }
      return check_launch("afno_mlp3_kernel");
    }
  }
  switch (rt) {
    case 1: hipLaunchKernelGGL((afno_mlp3_kernel<1, BS, ACTK>), grid, blk, 0, s, p); break;
    case 2: hipLaunchKernelGGL((afno_mlp3_kernel<2, BS, ACTK>), grid, blk, 0, s, p); break;
    case 3: hipLaunchKernelGGL((afno_mlp3_kernel<3, BS, ACTK>), grid, blk, 0, s, p); break;
    case 4: hipLaunchKernelGGL((afno_mlp3_kernel<4, BS, ACTK>), grid, blk, 0, s, p); break;
    default: hipLaunchKernelGGL((afno_mlp3_kernel<5, BS, ACTK>), grid, blk, 0, s, p); break;
  }
  return check_launch("afno_mlp3_kernel");
}
template <int BS>
static int launch3(const AfnoMlpArgs& p, int rt, hipStream_t s) {
  return p.act == DPOT_ACT_GELU ? launch3_rt<BS, DPOT_ACT_GELU>(p, rt, s) : launch3_rt<BS, -1>(p, rt, s);
}
template <int NW>
static int launch_nw(const AfnoMlpArgs& p, int rt, hipStream_t s) {
  // GELU (the DPOT default) gets its own instantiation; every other activation goes through the run-time switch
  return p.act == DPOT_ACT_GELU ? launch_rt<NW, DPOT_ACT_GELU>(p, rt, s) : launch_rt<NW, -1>(p, rt, s);
}

}  // namespace dpot

using namespace dpot;

extern "C" int dpot_afno_mlp2_supported(int nb, int bs) {
  const int N = 2 * bs;
  return nb > 0 && (N == 64 || N == 128 || N == 192 || N == 256) ? 1 : 0;
}

extern "C" int dpot_afno_block_weights(const float* wbig, float* fwd, float* bwd, int nmat, int N,
                                       dpot_stream_t stream) {
  DPOT_REQUIRE(wbig && (fwd || bwd) && nmat > 0 && N > 0 && N % 16 == 0, "afno_block_weights: bad argument");
  const long long total = (long long)nmat * N * N;
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(afno_block_weights_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), wbig, fwd, bwd, N,
                     total);
  return check_launch("afno_block_weights_kernel");
}

extern "C" int dpot_afno_mlp3_supported(int nb, int bs) { return nb > 0 && (bs == 128 || bs == 96 || bs == 64) ? 1 : 0; }

extern "C" int dpot_afno_pack_all(const dpot_afno_pack_job* jobs_dev, int njobs, int nb, int bs, int layout,
                                  dpot_stream_t stream) {
  DPOT_REQUIRE(jobs_dev && njobs > 0 && njobs <= 65535 && nb > 0 && bs > 0, "afno_pack_all: bad argument");
  DPOT_REQUIRE(layout == 0 || (layout == 1 && dpot_afno_mlp3_supported(nb, bs)), "afno_pack_all: layout 1 needs bs in {64, 96, 128}");
  DPOT_REQUIRE((2 * bs) % 16 == 0, "afno_pack_all: 2*bs must be a multiple of 16 for the blocked copies");
  long long g = ((long long)nb * 4 * bs * bs + 255) / 256;
  if (g > 512) g = 512;
  hipLaunchKernelGGL(afno_pack_all_kernel, dim3((unsigned)g, njobs), dim3(256), 0, as_stream(stream), jobs_dev, nb, bs,
                     layout);
  return check_launch("afno_pack_all_kernel");
}

extern "C" int dpot_afno_mlp2(const float* X, const float* WaT, const float* ba, const float* WbT, const float* bb,
                              const float* aux, float* pre, float* mid, float* Y, int M, int nb, int bs, int ldx,
                              int ldo, int act, int mode, int layout, dpot_stream_t stream) {
  DPOT_REQUIRE(X && WaT && WbT && Y && M > 0, "afno_mlp2: bad argument");
  DPOT_REQUIRE(layout == 0 || (layout == 1 && dpot_afno_mlp3_supported(nb, bs)), "afno_mlp2: weight layout 1 needs bs in {64, 96, 128}");
  DPOT_REQUIRE(dpot_afno_mlp2_supported(nb, bs), "afno_mlp2: unsupported block size bs=%d (2*bs must be 64/128/192/256)", bs);
  DPOT_REQUIRE(mode == 0 || (mode == 1 && aux != nullptr), "afno_mlp2: mode 1 (backward) needs aux");
  const int N = 2 * bs;
  DPOT_REQUIRE(ldx >= nb * N && ldo >= nb * N && ldx % 4 == 0 && ldo % 4 == 0, "afno_mlp2: bad leading dimension");
  DPOT_REQUIRE(aligned16(X) && aligned16(WaT) && aligned16(WbT) && aligned16(Y) && aligned16(ba) && aligned16(bb) &&
                   aligned16(aux) && aligned16(pre) && aligned16(mid),
               "afno_mlp2: pointers must be 16-byte aligned");
  AfnoMlpArgs p;
  p.X = X; p.Wa = WaT; p.Wb = WbT; p.ba = ba; p.bb = bb; p.aux = aux; p.pre = pre; p.mid = mid; p.Y = Y;
  p.ldx = ldx; p.ldo = ldo; p.M = M; p.nb = nb; p.act = act; p.mode = mode;
  const int rt = pick_rt(M, nb);
  p.panels = (M + 16 * rt - 1) / (16 * rt);
  hipStream_t s = as_stream(stream);
  if (layout == 1) return bs == 128 ? launch3<128>(p, rt, s) : bs == 96 ? launch3<96>(p, rt, s) : launch3<64>(p, rt, s);
  switch (N / 64) {
    case 1: return launch_nw<2>(p, rt, s);
    case 2: return launch_nw<4>(p, rt, s);
    case 3: return launch_nw<6>(p, rt, s);
    default: return launch_nw<8>(p, rt, s);
  }
}
