// gemm.hip - fp32 GEMM on the CDNA4 matrix cores for every dense contraction of the DPOT step:
//   AFNO block-diagonal complex MLP (as a real GEMM with Wbig), channel MLP (1x1 convs), patch embed,
//   TimeAggregator, ConvTranspose de-embed, cls head - forward, dgrad and wgrad (split-K).
//
// Design (gfx950):
//   * v_mfma_f32_32x32x2_f32 - exact fp32 (k-ordered fma chain), 157 TF peak, same numerics class as the
//     reference's fp32 CPU path (north_star tolerance rtol 1e-4 leaves no room for plain bf16; gemm_split.h holds
//     the fp32-accurate bf16x6 variant selected by desc.precision)
//   * 256 threads = 4 waves in a 2x2 grid; each wave owns a (BM/2)x(BN/2) sub-tile = (BM/64)x(BN/64)
//     accumulators of 32x32 (16 VGPRs each)
//   * BK = 32 K-slab staged through LDS; global loads for slab t+1 are issued before the MFMAs of slab t
//     (register double buffering).  Loads are UNCONDITIONAL from clamped addresses + a select: a predicated
//     load makes hipcc branch around it and wait vmcnt(0) per load, which serialises the whole prefetch
//   * K-contiguous operands sit in LDS as [row][BK+4] (ds_read_b128 per lane = 4 k-steps, conflict free);
//     row-contiguous operands (transposed views) sit as [BK][rows] and are read with ds_read_b32
//   * epilogue works on whole 16-element accumulator fragments: all aux/residual loads of a fragment are issued
//     together (clamped addresses), only the stores are predicated
//   * blockIdx -> tile mapping is XCD aware: the 8 XCDs get contiguous chunks of tile space so that
//     workgroups sharing an A row-panel / the whole weight matrix hit the same 4 MiB L2
//   * 64x64 tiles by default (4-5 co-resident workgroups per CU hide each other's barriers / prologues / epilogues;
//     measured never slower than 128x128 in the DPOT-Ti/S/M/L steps); 128x128 stays selectable through desc.tile
//   * split-K (wgrad: K = tokens*batch is huge, M x N small) writes partials to a workspace that a second
//     kernel reduces in fixed order - deterministic, no atomics
#include "common.h"
#include "gemm_epi.h"

namespace dpot {

constexpr int BK = 32;
constexpr int KPAD = 4;  // K-contiguous LDS rows are BK+4 floats: odd multiple of 16 B -> b128 reads conflict free

// ---- global -> register tile loaders ------------------------------------------------------------------
// Loads are UNCONDITIONAL: out-of-range rows / k are CLAMPED to a valid address instead of being predicated.
// (A load followed by a select is turned by hipcc into a branch around the load + s_waitcnt vmcnt(0), which
// serialises the whole prefetch.)  Clamped rows only feed accumulator rows/columns the epilogue never stores;
// the k >= kmax part of the last, partial K-slab is zeroed in LDS after the store (zero_tail_*).
// K-contiguous source: element (r, k) at base[r*ld + k]; tile [R][BK]
template <int R, bool VEC>
__device__ __forceinline__ void load_kcontig(float4 (&reg)[R / 32], const float* __restrict__ base, int ld,
                                             int r0, int rmax, int k0, int kmax, int tid) {
#pragma unroll
  for (int i = 0; i < R / 32; ++i) {
    const int f = tid + 256 * i;
    const int r = r0 + (f >> 3);
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
// row-contiguous source: element (r, k) at base[k*ld + r]; tile [BK][R]
template <int R, bool VEC>
__device__ __forceinline__ void load_rcontig(float4 (&reg)[R / 32], const float* __restrict__ base, int ld,
                                             int r0, int rmax, int k0, int kmax, int tid) {
  constexpr int F4_PER_ROW = R / 4;
#pragma unroll
  for (int i = 0; i < R / 32; ++i) {
    const int f = tid + 256 * i;
    const int k = k0 + f / F4_PER_ROW;
    const int r = r0 + ((f % F4_PER_ROW) << 2);
    const float* row = base + (long long)min(k, kmax - 1) * ld;
    float4 v;
    if constexpr (VEC) {
      v = *reinterpret_cast<const float4*>(row + min(r, rmax - 4));
    } else {
      const int rl = rmax - 1;
      v.x = row[min(r + 0, rl)]; v.y = row[min(r + 1, rl)]; v.z = row[min(r + 2, rl)]; v.w = row[min(r + 3, rl)];
    }
    reg[i] = v;
  }
}
// zero the k >= krem part of a staged slab (last, partial K-slab only)
template <int R>
__device__ __forceinline__ void zero_tail_kcontig(float* lds, int krem, int tid) {
  for (int idx = tid; idx < R * BK; idx += 256) {
    const int r = idx / BK, k = idx % BK;
    if (k >= krem) lds[r * (BK + KPAD) + k] = 0.f;
  }
}
template <int R>
__device__ __forceinline__ void zero_tail_rcontig(float* lds, int krem, int tid) {
  for (int idx = tid + krem * R; idx < BK * R; idx += 256) lds[idx] = 0.f;
}
template <int R>
__device__ __forceinline__ void store_kcontig(float* lds, const float4 (&reg)[R / 32], int tid) {
#pragma unroll
  for (int i = 0; i < R / 32; ++i) {
    const int f = tid + 256 * i;
    *reinterpret_cast<float4*>(&lds[(f >> 3) * (BK + KPAD) + ((f & 7) << 2)]) = reg[i];
  }
}
template <int R>
__device__ __forceinline__ void store_rcontig(float* lds, const float4 (&reg)[R / 32], int tid) {
  constexpr int F4_PER_ROW = R / 4;
#pragma unroll
  for (int i = 0; i < R / 32; ++i) {
    const int f = tid + 256 * i;
    *reinterpret_cast<float4*>(&lds[(f / F4_PER_ROW) * R + ((f % F4_PER_ROW) << 2)]) = reg[i];
  }
}

template <int R, bool KCONTIG>
struct TileLds {
  static constexpr int floats = KCONTIG ? R * (BK + KPAD) : BK * R;
};

// fetch the MFMA operand values of one lane for 4 consecutive k-steps (k = kk + 4*kh + s, s = 0..3)
template <int R, bool KCONTIG>
__device__ __forceinline__ float4 frag(const float* lds, int row, int kk, int kh) {
  if constexpr (KCONTIG) {
    return *reinterpret_cast<const float4*>(&lds[row * (BK + KPAD) + kk + 4 * kh]);
  } else {
    const float* p = &lds[(kk + 4 * kh) * R + row];
    return make_float4(p[0], p[R], p[2 * R], p[3 * R]);
  }
}

// TA: A stored [K,M] (row-contiguous tile);  TB: B stored [N,K] (K-contiguous tile)
// VEC: 16-byte global loads (alignment / divisibility checked on the host)
// TAG only changes the kernel's name (so rocprof reports the AFNO mixer launches on their own line)
template <int BM, int BN, bool TA, bool TB, bool VEC, int TAG>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs p) {
  constexpr bool A_KC = !TA;
  constexpr bool B_KC = TB;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int OPER_FLOATS = TileLds<BM, A_KC>::floats + TileLds<BN, B_KC>::floats;
  constexpr int STAGE_FLOATS = 4 * 32 * EPI_LD;
  __shared__ __attribute__((aligned(16))) float smem[OPER_FLOATS > STAGE_FLOATS ? OPER_FLOATS : STAGE_FLOATS];
  float* As = smem;
  float* Bs = smem + TileLds<BM, A_KC>::floats;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;

  // ---- XCD-aware tile mapping: hardware round-robins consecutive workgroup ids over the 8 XCDs; give each
  //      XCD a contiguous range of tile indices (bijective for any tile count)
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
  const bool cs_on = (p.cs_of == 1 && tn_idx == 0) || (p.cs_of == 2 && tm_idx == 0);
  float cs_acc = 0.f;

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

  float4 ra[BM / 32], rb[BN / 32];
  auto gload = [&](int k0) {
    if constexpr (A_KC) load_kcontig<BM, VEC>(ra, A, p.lda, m0, p.M, k0, kend, tid);
    else load_rcontig<BM, VEC>(ra, A, p.lda, m0, p.M, k0, kend, tid);
    if constexpr (B_KC) load_kcontig<BN, VEC>(rb, B, p.ldb, n0, p.N, k0, kend, tid);
    else load_rcontig<BN, VEC>(rb, B, p.ldb, n0, p.N, k0, kend, tid);
  };
  auto sstore = [&](int k0) {
    if constexpr (A_KC) store_kcontig<BM>(As, ra, tid); else store_rcontig<BM>(As, ra, tid);
    if constexpr (B_KC) store_kcontig<BN>(Bs, rb, tid); else store_rcontig<BN>(Bs, rb, tid);
    if (k0 + BK > kend) {  // uniform: only the last slab of a K that is not a multiple of 32
      __syncthreads();
      const int krem = kend - k0;
      if constexpr (A_KC) zero_tail_kcontig<BM>(As, krem, tid); else zero_tail_rcontig<BM>(As, krem, tid);
      if constexpr (B_KC) zero_tail_kcontig<BN>(Bs, krem, tid); else zero_tail_rcontig<BN>(Bs, krem, tid);
    }
  };

  if (nk > 0) {
    gload(kbeg);
    sstore(kbeg);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
      if (cs_on) {   // uniform per workgroup: only the first tile column (row) of the grid sums the operand
        if constexpr (TA) {
          if (p.cs_of == 1)
            for (int k = tid / BM; k < BK; k += 256 / BM) cs_acc += As[k * BM + tid % BM];
        }
        if constexpr (!TB) {
          if (p.cs_of == 2)
            for (int k = tid / BN; k < BK; k += 256 / BN) cs_acc += Bs[k * BN + tid % BN];
        }
      }
#pragma unroll
      for (int kk = 0; kk < BK; kk += 8) {
        float4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = frag<BM, A_KC>(As, wm * WM + i * 32 + li, kk, kh);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = frag<BN, B_KC>(Bs, wn * WN + j * 32 + li, kk, kh);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
          }
      }
      __syncthreads();
      if (kt + 1 < nk) {
        sstore(kbeg + (kt + 1) * BK);
        __syncthreads();
      }
    }
  }

  // ---- epilogue (the operand slabs in LDS are dead: reuse them as per-wave staging)
  __syncthreads();
  if (cs_on) {
    // combine the k-lanes in a fixed order; one value per operand column owned by this tile
    const int R = p.cs_of == 1 ? BM : BN;
    smem[tid] = cs_acc;
    __syncthreads();
    if (tid < R) {
      float s = 0.f;
      for (int l = 0; l < 256 / R; ++l) s += smem[l * R + tid];
      const int L = p.cs_of == 1 ? p.M : p.N;
      const int g = (p.cs_of == 1 ? m0 : n0) + tid;
      if (g < L) {
        if (p.splits > 1)
          p.ws[(long long)p.splits * p.batch * p.M * p.N + ((long long)zs * p.batch + zb) * L + g] = s;
        else
          p.cs_out[zb * p.sCs + g] = s;
      }
    }
    __syncthreads();
  }
  float* stage = smem + wave * (32 * EPI_LD);
  // explicit (i, j) instances: a `#pragma unroll` over loops that contain the non-unrolled epilogue loop is
  // declined by the compiler and would turn acc[i][j] into a scratch array
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

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int batch,
                                                            const EpiArgs e, float* __restrict__ cs_out,
                                                            long long sCs, int csL) {
  const long long MN = (long long)e.M * e.N;
  const long long total = MN * batch;
  if (cs_out) {   // column-sum partials [split][batch][csL] live behind the matrix partials
    const float* wc = ws + (long long)splits * total;
    const long long nc = (long long)batch * csL;
    for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < nc; idx += (long long)gridDim.x * 256) {
      float v = 0.f;
      for (int s = 0; s < splits; ++s) v += wc[(long long)s * nc + idx];
      cs_out[(idx / csL) * sCs + idx % csL] = v;
    }
  }
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    float v = 0.f;
    int s = 0;
    for (; s + 4 <= splits; s += 4) {  // 4 independent loads in flight, summed in a fixed order
      const float a0 = ws[(long long)(s + 0) * total + idx], a1 = ws[(long long)(s + 1) * total + idx];
      const float a2 = ws[(long long)(s + 2) * total + idx], a3 = ws[(long long)(s + 3) * total + idx];
      v += (a0 + a1) + (a2 + a3);
    }
    for (; s < splits; ++s) v += ws[(long long)s * total + idx];
    const int b = (int)(idx / MN);
    const long long rem = idx - (long long)b * MN;
    epi_store(e, b, (int)(rem / e.N), (int)(rem % e.N), v);
  }
}

// split-K reduction of the AFNO mixer wgrad that also undoes the [[Wr, Wi], [-Wi, Wr]] packing:
//   dWr = dWbig[0:bs, 0:bs] + dWbig[bs:, bs:],  dWi = dWbig[0:bs, bs:] - dWbig[bs:, 0:bs]   -> dw[2, nb, bs, bs]
//   db[part, k, c] = colsum partials of the B operand, [nb][2bs] = [re | im]                -> db[2, nb, bs]
__global__ __launch_bounds__(256) void splitk_reduce_afno_kernel(const float* __restrict__ ws, int splits, int nb,
                                                                 int bs, float* __restrict__ dw,
                                                                 float* __restrict__ db) {
  const int n2 = 2 * bs;
  const long long MN = (long long)n2 * n2, total = MN * nb;
  const long long nw = (long long)nb * bs * bs;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < nw; idx += (long long)gridDim.x * 256) {
    const int o = (int)(idx % bs), i = (int)((idx / bs) % bs), k = (int)(idx / ((long long)bs * bs));
    const float* base = ws + (long long)k * MN;
    float rr = 0.f, ii = 0.f, ri = 0.f, ir = 0.f;
    for (int s = 0; s < splits; ++s) {           // fixed order
      const float* W = base + (long long)s * total;
      rr += W[(long long)i * n2 + o];
      ii += W[(long long)(bs + i) * n2 + bs + o];
      ri += W[(long long)i * n2 + bs + o];
      ir += W[(long long)(bs + i) * n2 + o];
    }
    dw[idx] = rr + ii;
    dw[nw + idx] = ri - ir;
  }
  if (db) {
    const float* wc = ws + (long long)splits * total;
    const long long nc = (long long)nb * n2;
    for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < nc; idx += (long long)gridDim.x * 256) {
      float v = 0.f;
      for (int s = 0; s < splits; ++s) v += wc[(long long)s * nc + idx];
      const int c = (int)(idx % bs), part = (int)((idx / bs) & 1), k = (int)(idx / n2);
      db[((long long)part * nb + k) * bs + c] = v;
    }
  }
}

constexpr int NUM_CU = 256;

static int pick_tile(int M, int N, int batch, int forced) {
  if (forced == 64 || forced == 128) return forced;
  if (M <= 64 || N <= 64) return 64;
  // measured in the whole train step of DPOT-Ti/S/M/L (scripts/gpu_configs.py, bench.py): the 64x64 tile (4-5
  // co-resident workgroups per CU hide each other's barriers, prologues and epilogues) is never slower than the
  // 128x128 tile for the native fp32 kernel, so auto picks it always; 128 stays available through desc.tile
  const long long t128 = (long long)cdiv(M, 128) * cdiv(N, 128) * batch;
  (void)t128;
  return 64;
}

// bf16x6 path: the 128x128 tile owns 120 KB of LDS (one workgroup per CU) and halves both the operand-split work
// and the L2 traffic per MFMA of the 64x64 tile, so it wins whenever its workgroups fill the 256 CUs evenly
static int pick_tile_split(int M, int N, int batch, int splits, int forced) {
  if (forced == 64 || forced == 128) return forced;
  if (M <= 64 || N <= 64) return 64;
  const long long t128 = (long long)cdiv(M, 128) * cdiv(N, 128) * batch * splits;
  const long long waves = (t128 + NUM_CU - 1) / NUM_CU;
  return 4 * t128 >= 3 * waves * NUM_CU ? 128 : 64;
}

}  // namespace dpot

#include "gemm_split.h"

using namespace dpot;

extern "C" int dpot_gemm_auto_splitk(int M, int N, int K, int batch) {
  const int t = pick_tile(M, N, batch, 0);
  const long long tiles = (long long)cdiv(M, t) * cdiv(N, t) * batch;
  const int ktiles = cdiv(K, BK);
  constexpr int per_cu = 2, min_slabs = 4;
  if (tiles >= per_cu * NUM_CU || ktiles < 8) return 1;
  long long s = (per_cu * NUM_CU + tiles - 1) / tiles;   // aim at ~512 workgroups (2 per CU; measured in the DPOT
                                                         // step: 4 per CU costs more in workspace traffic + reduction than it gains)
  const long long smax = ktiles / min_slabs;             // keep >= 4 K-slabs per split
  if (s > smax) s = smax;
  if (s > 512) s = 512;
  return s < 1 ? 1 : (int)s;
}

// DPOT_GEMM_AUTO: both kernels are fp32-accurate, so the choice is pure speed.  The bf16x6 kernel has the higher
// roof but a longer pipeline (two slabs of prologue, 120 KB of LDS per workgroup): it pays on the large products
// (measured in the DPOT-Tiny step: 3 GFLOP and up: channel-MLP, embed and de-embed GEMMs), not on the 2.4 GFLOP batched AFNO mixer.
static int resolve_precision(int precision, int M, int N, int K, int batch) {
  if (precision != DPOT_GEMM_AUTO) return precision;
  constexpr double min_gflop = 3.0;
  return 2.0 * M * N * K * batch >= min_gflop * 1e9 ? DPOT_GEMM_BF16X6 : DPOT_GEMM_F32;
}

extern "C" int dpot_gemm_auto_splitk2(int M, int N, int K, int batch, int precision) {
  precision = resolve_precision(precision, M, N, K, batch);
  if (precision == DPOT_GEMM_F32 || M <= 64 || N <= 64) return dpot_gemm_auto_splitk(M, N, K, batch);
  const long long t128 = (long long)cdiv(M, 128) * cdiv(N, 128) * batch;
  const int ktiles = cdiv(K, BK);
  if (4 * t128 >= 3 * NUM_CU || ktiles < 16) return 1;
  long long s = (NUM_CU + t128 / 2) / t128;         // one 128x128 workgroup per CU
  const long long smax = ktiles / 8;                // keep >= 8 K-slabs per split (pipeline fill / drain)
  if (s > smax) s = smax;
  return s < 1 ? 1 : (int)s;
}

extern "C" int64_t dpot_gemm_workspace_bytes(const dpot_gemm_desc* d) {
  if (!d || d->splitk <= 1) return 0;
  const int64_t L = d->colsum_of == 1 ? d->M : d->colsum_of == 2 ? d->N : 0;
  return (int64_t)d->splitk * d->batch * ((int64_t)d->M * d->N + L) * (int64_t)sizeof(float);
}

template <int BMN, bool VEC>
static void launch_gemm(const dpot_gemm_desc* d, const GemmArgs& p, dim3 grid, hipStream_t s) {
  const int key = (d->transA ? 2 : 0) | (d->transB ? 1 : 0);
  if (d->tag == 1 && key == 0) {
    hipLaunchKernelGGL((gemm_f32_kernel<BMN, BMN, false, false, VEC, 1>), grid, dim3(256), 0, s, p);
    return;
  }
  switch (key) {
    case 0: hipLaunchKernelGGL((gemm_f32_kernel<BMN, BMN, false, false, VEC, 0>), grid, dim3(256), 0, s, p); break;
    case 1: hipLaunchKernelGGL((gemm_f32_kernel<BMN, BMN, false, true, VEC, 0>), grid, dim3(256), 0, s, p); break;
    case 2: hipLaunchKernelGGL((gemm_f32_kernel<BMN, BMN, true, false, VEC, 0>), grid, dim3(256), 0, s, p); break;
    default: hipLaunchKernelGGL((gemm_f32_kernel<BMN, BMN, true, true, VEC, 0>), grid, dim3(256), 0, s, p); break;
  }
}

extern "C" int dpot_gemm_f32(const dpot_gemm_desc* d, dpot_stream_t stream) {
  DPOT_REQUIRE(d != nullptr, "gemm: null descriptor");
  DPOT_REQUIRE(d->A && d->B && d->C, "gemm: null operand");
  DPOT_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch > 0, "gemm: bad shape M=%d N=%d K=%d batch=%d", d->M,
               d->N, d->K, d->batch);
  DPOT_REQUIRE(d->lda >= (d->transA ? d->M : d->K), "gemm: lda=%d too small", d->lda);
  DPOT_REQUIRE(d->ldb >= (d->transB ? d->K : d->N), "gemm: ldb=%d too small", d->ldb);
  DPOT_REQUIRE(d->ldc >= d->N, "gemm: ldc=%d too small", d->ldc);
  DPOT_REQUIRE(d->epi_mode != DPOT_EPI_DACT || d->aux != nullptr, "gemm: DACT epilogue needs aux");
  DPOT_REQUIRE(d->epi_mode != DPOT_EPI_AFNO_WGRAD ||
                   (d->splitk > 1 && d->M == d->N && d->M % 2 == 0 && !d->bias && !d->res && !d->preact &&
                    !d->accumulate && (d->colsum_of == 0 || d->colsum_of == 2)),
               "gemm: the AFNO wgrad epilogue needs split-K, M == N == 2*bs and no other epilogue term");
  const int splits = d->splitk > 1 ? d->splitk : 1;
  DPOT_REQUIRE(splits == 1 || d->workspace != nullptr, "gemm: split-K needs a workspace");
  DPOT_REQUIRE(d->colsum_of == 0 || (d->colsum_out != nullptr && ((d->colsum_of == 1 && d->transA) ||
                                                                     (d->colsum_of == 2 && !d->transB))),
               "gemm: fused column sums need colsum_out and a [K,M]-stored A (colsum_of=1) or [K,N]-stored B (=2)");

  GemmArgs p;
  p.A = d->A; p.B = d->B;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb;
  p.sA = d->strideA; p.sB = d->strideB;
  p.batch = d->batch; p.splits = splits;
  const int ktiles = cdiv(d->K, BK);
  p.ktiles_per_split = cdiv(ktiles, splits);
  // vector (16 B) loads need: 16-B aligned base + batch stride, ld % 4 == 0 and the contiguous extent % 4 == 0
  const int contA = d->transA ? d->M : d->K;
  const int contB = d->transB ? d->K : d->N;
  const bool vecA = aligned16(d->A) && (d->lda % 4 == 0) && (contA % 4 == 0) && (d->strideA % 4 == 0);
  const bool vecB = aligned16(d->B) && (d->ldb % 4 == 0) && (contB % 4 == 0) && (d->strideB % 4 == 0);
  const bool vec = vecA && vecB;
  p.ws = d->workspace;
  p.cs_out = d->colsum_out; p.sCs = d->strideColsum; p.cs_of = d->colsum_of;
  EpiArgs& e = p.e;
  e.C = d->C; e.ldc = d->ldc; e.sC = d->strideC;
  e.bias = d->bias; e.sBias = d->strideBias;
  e.aux = d->aux; e.ldaux = d->ldaux; e.sAux = d->strideAux;
  e.pre = d->preact; e.ldpre = d->ldpre; e.sPre = d->stridePre;
  e.res = d->res; e.ldres = d->ldres; e.res_div = d->res_div; e.res_mod = d->res_mod; e.sRes = d->strideRes;
  e.act = d->act; e.mode = d->epi_mode; e.accumulate = d->accumulate;
  e.M = d->M; e.N = d->N;

  auto vec_ok = [](const void* ptr, int ld, long long stride) {
    return ptr == nullptr || (aligned16(ptr) && ld % 4 == 0 && stride % 4 == 0);
  };
  p.evec = (d->N % 4 == 0) && vec_ok(d->C, d->ldc, d->strideC) && vec_ok(d->bias, 4, d->strideBias) &&
           vec_ok(d->aux, d->ldaux, d->strideAux) && vec_ok(d->preact, d->ldpre, d->stridePre) &&
           vec_ok(d->res, d->ldres, d->strideRes);
  const int precision = resolve_precision(d->precision, d->M, d->N, d->K, d->batch);
  DPOT_REQUIRE(precision == DPOT_GEMM_F32 || precision == DPOT_GEMM_BF16X6 || precision == DPOT_GEMM_BF16,
               "gemm: bad precision %d", d->precision);
  const int t = precision != DPOT_GEMM_F32 ? pick_tile_split(d->M, d->N, d->batch, splits, d->tile)
                                              : pick_tile(d->M, d->N, d->batch, d->tile);
  p.tilesM = cdiv(d->M, t); p.tilesN = cdiv(d->N, t);
  const long long ntiles = (long long)p.tilesM * p.tilesN;
  DPOT_REQUIRE(ntiles < (1ll << 31) && (long long)d->batch * splits <= 65535, "gemm: grid too large");
  dim3 grid((unsigned)ntiles, 1, (unsigned)(d->batch * splits));
  hipStream_t s = as_stream(stream);
  // weight gradients (A^T B, both operands token-major) with split-K: the dedicated kernel of gemm_tn.hip leaves its
  // partial sums in the workspace exactly as the generic kernel would, so the reduce launches below serve both
  const int tn_rc = precision == DPOT_GEMM_F32 ? dpot_gemm_tn_try(d, s) : -1;
  if (tn_rc > 0) return tn_rc;
  if (tn_rc == 0) {
    // launched
  } else if (precision == DPOT_GEMM_BF16) {
    // reduced precision on request (BASELINE configs[2] "bf16 channel-MLP on MFMA"): operands rounded to bf16 between
    // the global load and the LDS store, ONE product per k-step on v_mfma_f32_32x32x16_bf16, fp32 accumulation
    if (t == 128) {
      if (vec) launch_gemm_split<128, true, 1, 8>(d, p, grid, s); else launch_gemm_split<128, false, 1, 8>(d, p, grid, s);
    } else {
      if (vec) launch_gemm_split<64, true, 1, 4>(d, p, grid, s); else launch_gemm_split<64, false, 1, 4>(d, p, grid, s);
    }
  } else if (precision == DPOT_GEMM_BF16X6) {
    constexpr int waves128 = 8;
    if (t == 128 && waves128 == 8) {
      if (vec) launch_gemm_split<128, true, 3, 8>(d, p, grid, s); else launch_gemm_split<128, false, 3, 8>(d, p, grid, s);
    } else if (t == 128) {
      if (vec) launch_gemm_split<128, true, 3, 4>(d, p, grid, s); else launch_gemm_split<128, false, 3, 4>(d, p, grid, s);
    } else {
      if (vec) launch_gemm_split<64, true, 3, 4>(d, p, grid, s); else launch_gemm_split<64, false, 3, 4>(d, p, grid, s);
    }
  } else if (t == 128) {
    if (vec) launch_gemm<128, true>(d, p, grid, s); else launch_gemm<128, false>(d, p, grid, s);
  } else {
    if (vec) launch_gemm<64, true>(d, p, grid, s); else launch_gemm<64, false>(d, p, grid, s);
  }
  int rc = check_launch("gemm_f32_kernel");
  if (rc != DPOT_OK) return rc;
  if (splits > 1 && d->epi_mode == DPOT_EPI_AFNO_WGRAD) {
    const int bs = d->M / 2;
    long long blocks = ((long long)d->batch * bs * bs + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_afno_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)d->workspace,
                       splits, d->batch, bs, d->C, d->colsum_of ? d->colsum_out : (float*)nullptr);
    rc = check_launch("splitk_reduce_afno_kernel");
  } else if (splits > 1) {
    const long long total = (long long)d->M * d->N * d->batch;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)d->workspace, splits,
                       d->batch, e, d->colsum_of ? d->colsum_out : (float*)nullptr, (long long)d->strideColsum,
                       d->colsum_of == 1 ? d->M : d->N);
    rc = check_launch("splitk_reduce_kernel");
  }
  return rc;
}
