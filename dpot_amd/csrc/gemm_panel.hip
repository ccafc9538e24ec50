// gemm_panel.hip - fp32 "panel" GEMM for the Linear / 1x1-conv layers of the DPOT step whose weight is a static operand:
//
//      C[M, N] = epilogue(A[M, K] @ W[K, N])       A row-major (K contiguous), W pre-packed once per optimiser step
//
// The forward (x W^T) and the data gradient (dy W) of every channel-MLP / de-embed GEMM have this form; the weight is
// packed in FRAGMENT-BLOCK-MAJOR order by dpot_panel_pack_weights (as for csrc/afno_mlp.hip, whose structure this kernel
// shares - see there for the measurements behind it):
//   * a workgroup owns a panel of 16*RT rows x one 32*NW-column chunk (256 columns at NW = 8) and streams K in 16-deep
//     slabs; v_mfma_f32_16x16x4_f32, RT x 2 accumulators per compute wave;
//   * NW compute waves (two per SIMD at NW = 8: they share the matrix pipe during the slabs and overlap each other's
//     LDS / VALU latencies in the epilogue) + 2 loader waves that only issue LDS-DMA: weight blocks are 1 KiB contiguous
//     pieces, A rows arrive 32 k at a time as full 128-byte lines with an XOR swizzle on the source address;
//   * ring of three weight buffers / two A buffers, loaders three slabs ahead, ONE barrier per slab, counted vmcnt;
//     compute waves fetch the fragments of slab t+1 into a second register set under the MFMAs of slab t;
//   * epilogue through a per-wave LDS slab: bias, pre-activation save, activation or act'(aux) product, residual,
//     16-byte global accesses.
// Against the generic 64x64-tile kernel (gemm.hip: 4-5 co-resident workgroups in lock step, ~76 % matrix-pipe
// utilisation at K = 512) the panels cut the prologue / epilogue share and the L2 -> LDS traffic (32 vs 16 FLOP/B).
#include <type_traits>

#include "common.h"

namespace dpot {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PanelArgs {
  const float* A;      // [M, lda]
  const float* W;      // [N/16 column tiles][K/16 slabs][256]  fragment-block-major
  const float* bias;   // [N] or NULL
  const float* aux;    // DACT: pre-activation [M, ldaux]
  const float* res;    // residual [M, ldres] or NULL
  float* pre;          // optional pre-activation output [M, ldpre]
  float* C;            // [M, ldc]
  int lda, ldaux, ldres, ldpre, ldc;
  int M, N, K, panels, nchunks, act, mode;
};

template <int N>
__device__ __forceinline__ void pwait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pglds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int RT, int NW, int ACTK>
__global__ __launch_bounds__(64 * (NW + 2)) void gemm_panel_kernel(const PanelArgs p) {
  constexpr int NT = 2;             // 16-column tiles per compute wave
  constexpr int NC = 32 * NW;       // columns of a chunk
  constexpr int NCT = 2 * NW;       // 16-column tiles of a chunk
  constexpr int BFL = NCT * 256;    // floats of one weight slab  (NC x 16)
  constexpr int AFL = RT * 512;     // floats of one A super-slab (16*RT rows x 32 k)
  constexpr int NXI = 2 * RT;       // DMA instructions per A super-slab (8 rows x 128 B each)
  constexpr int WCOLS = 16 * NT;    // columns per compute wave
  constexpr int WB = NCT / 2, XB = RT;               // DMA pieces per batch and loader wave
  __shared__ __attribute__((aligned(16))) float lds[3 * BFL + 2 * AFL];
  float* const Bb = lds;                    // [3][NCT][256]
  float* const Ab = lds + 3 * BFL;          // [2][RT][16 rows][32 k], chunk-swizzled
  float* const stage_all = lds;             // epilogue staging (ring buffer 0; everything has landed and been read)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int NSLAB = p.K >> 4;               // K % 32 == 0 (host-checked): NSLAB is even

  // XCD-contiguous order, chunk major: the workgroups that share a weight chunk sit on one chiplet's L2
  const int nitems = p.nchunks * p.panels;
  int item;
  {
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = nitems >> 3, r = nitems & 7;
    item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int chunk = item / p.panels, panel = item - chunk * p.panels;
  const int row0 = panel * (16 * RT);
  const int col0 = chunk * NC;

  auto bar = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (wave >= NW) {
    // ================================ loader waves ================================
    const int L = wave - NW;
    const float* Wl = p.W + (long long)chunk * NCT * NSLAB * 256 + lane * 4;
    auto issue_w = [&](int t, float* dstbuf) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < NCT / 2; ++n) {
        const int c = L + 2 * n;
        pglds16(Wl + ((long long)c * NSLAB + t) * 256, dstbuf + c * 256);
      }
    };
    long long xoff[(NXI + 1) / 2];
#pragma unroll
    for (int n = 0; n < (NXI + 1) / 2; ++n) {
      const int q = L + 2 * n;
      const int rt = (8 * q + (lane >> 3)) & 15;
      int row = row0 + 8 * q + (lane >> 3);
      row = row < p.M ? row : p.M - 1;
      xoff[n] = (long long)row * p.lda + 4 * ((lane & 7) ^ ((rt >> 1) & 7));
    }
    auto issue_x = [&](int T, float* dstbuf) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < (NXI + 1) / 2; ++n) {
        const int q = L + 2 * n;
        if (q < NXI) pglds16(p.A + xoff[n] + 32 * T, dstbuf + q * 256);
      }
    };
    // slabs 0, 1 and A super-slab 0, then the newest batch (A super-slab 1 + slab 2) - see afno_mlp.hip
    issue_x(0, Ab);
    issue_w(0, Bb);
    issue_w(1, Bb + BFL);
    const bool x1 = NSLAB > 2, w2 = NSLAB > 2;
    if (x1) issue_x(1, Ab + AFL);
    if (w2) issue_w(2, Bb + 2 * BFL);
    if (w2) pwait_vm<WB + XB>(); else pwait_vm<0>();
    bar();                                              // P
    int ring = 0;
#pragma unroll 1
    for (int g = 0; g < NSLAB; ++g) {
      bar();                                            // B_g
      const int nx = g + 3;
      const int T1 = (g + 3) >> 1;
      const bool has_w = nx < NSLAB;
      const bool has_x = (g & 1) && has_w && T1 >= 2;
      if (has_x) issue_x(T1, Ab + (T1 & 1) * AFL);
      if (has_w) issue_w(nx, Bb + ring * BFL);
      ring = ring == 2 ? 0 : ring + 1;
      if (has_w && has_x) pwait_vm<WB + XB>();
      else if (has_w) pwait_vm<WB>();
      else pwait_vm<0>();
    }
    bar();                                              // S: end of the K loop
    return;
  }

  // ================================ compute waves ================================
  f32x4 acc[RT][NT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mma = [&](const f32x4 (&af)[RT], const f32x4 (&bf)[NT]) __attribute__((always_inline)) {
#ifdef PANEL_ABL_NOMFMA
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] += af[i] * bf[j][0];
    return;
#endif
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
  };
  auto read_w = [&](f32x4 (&bf)[NT], int ringbuf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
      bf[j] = *reinterpret_cast<const f32x4*>(Bb + ringbuf * BFL + (wave * NT + j) * 256 + lane * 4);
  };
  const int xfrag0 = fr * 32 + 4 * ((fq) ^ ((fr >> 1) & 7));
  const int xfrag1 = fr * 32 + 4 * ((4 + fq) ^ ((fr >> 1) & 7));
  auto read_x = [&](f32x4 (&af)[RT], int t) __attribute__((always_inline)) {
    const float* xs = Ab + ((t >> 1) & 1) * AFL + ((t & 1) ? xfrag1 : xfrag0);
#pragma unroll
    for (int i = 0; i < RT; ++i) af[i] = *reinterpret_cast<const f32x4*>(xs + i * 512);
  };

  f32x4 afA[RT], bfA[NT], afB[RT], bfB[NT];
  bar();                                               // P
  read_x(afA, 0);
  read_w(bfA, 0);
  {
    int r1 = 1, r2 = 2;
#pragma unroll 1
    for (int t = 0; t < NSLAB; t += 2) {
      bar();                                           // B_t
      read_x(afB, t + 1);
      read_w(bfB, r1);
      mma(afA, bfA);
      bar();                                           // B_(t+1)
      if (t + 2 < NSLAB) {
        read_x(afA, t + 2);
        read_w(bfA, r2);
      }
      mma(afB, bfB);
      r1 = r1 == 0 ? 2 : r1 - 1;
      r2 = r2 == 0 ? 2 : r2 - 1;
    }
  }
  bar();                                               // S: every wave is done with the ring; all DMA has landed
#ifdef PANEL_ABL_NOEPI   /* ablation builds (scripts/panel_ablation.sh): results are garbage */
  if (p.M > 0) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) a += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (a == 12345.678f) p.C[0] = a;
    return;
  }
#endif

  // ---- epilogue: acc (+bias) -> [pre] -> act | * act'(aux) -> (+res) -> C
  float* const stage = stage_all + wave * (16 * WCOLS);
  const int colw = col0 + 16 * NT * wave;              // first column of this wave
  float4 b4[NT];
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const int c4 = (it * 64 + lane) % (4 * NT);
    b4[it] = p.bias ? *reinterpret_cast<const float4*>(p.bias + colw + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto tile = [&](auto Ic) __attribute__((always_inline)) {
    constexpr int i = decltype(Ic)::value;
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
      const int g = it * 64 + lane;
      const int r = g / (4 * NT), c4 = g - r * (4 * NT);
      const f32x4 t = *reinterpret_cast<const f32x4*>(stage + g * 4);
      float v[4] = {t[0] + b4[it].x, t[1] + b4[it].y, t[2] + b4[it].z, t[3] + b4[it].w};
      const int row = row0 + 16 * i + r;
      const int col = colw + 4 * c4;
      const bool ok = row < p.M;
      const long long rowc = ok ? row : p.M - 1;
#ifdef PANEL_ABL_NOSTORE
      if (p.pre && ok && v[0] == 12345.678f) p.pre[rowc * p.ldpre + col] = v[0];
#elif !defined(PANEL_NO_NT_PRE)   // saved for the backward only: non-temporal, it would only push live tensors out of L2 / Infinity Cache
      if (p.pre && ok) {
        typedef float nt_f4 __attribute__((ext_vector_type(4)));
        const nt_f4 q = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(q, reinterpret_cast<nt_f4*>(p.pre + rowc * p.ldpre + col));
      }
#else
      if (p.pre && ok) *reinterpret_cast<float4*>(p.pre + rowc * p.ldpre + col) = make_float4(v[0], v[1], v[2], v[3]);
#endif
      if (p.mode == DPOT_EPI_ACT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ACTK == DPOT_ACT_GELU ? gelu_fwd(v[e]) : act_fwd(p.act, v[e]);
      } else if (p.mode == DPOT_EPI_DACT) {
        const float4 x4 = *reinterpret_cast<const float4*>(p.aux + rowc * p.ldaux + col);
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= ACTK == DPOT_ACT_GELU ? gelu_bwd(xs[e]) : act_bwd(p.act, xs[e]);
      }
      if (p.res) {
        const float4 r4 = *reinterpret_cast<const float4*>(p.res + rowc * p.ldres + col);
        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
      }
#ifdef PANEL_ABL_NOSTORE
      if (ok && v[0] == 12345.678f)
#else
      if (ok)
#endif
        *reinterpret_cast<float4*>(p.C + rowc * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __builtin_amdgcn_wave_barrier();
  };
  tile(std::integral_constant<int, 0>{});
  if constexpr (RT > 1) tile(std::integral_constant<int, 1>{});
  if constexpr (RT > 2) tile(std::integral_constant<int, 2>{});
  if constexpr (RT > 3) tile(std::integral_constant<int, 3>{});
  if constexpr (RT > 4) tile(std::integral_constant<int, 4>{});
}

// pack jobs (DEVICE table): W given as rows x K (row r, k) = trans ? src[k*ld + r] : src[r*ld + k]
// -> dst[(ct*NSLAB + t)*256 + 4*l + e] = element (row 16ct + (l&15), k 16t + 4(l>>4) + e)
// one destination quad (4 consecutive k of one row) per thread and trip, 32-bit index arithmetic (the element-wise
// version spent its time in two 64-bit divisions per element)
__global__ __launch_bounds__(256) void panel_pack_kernel(const dpot_pack_job* __restrict__ jobs) {
  const dpot_pack_job job = jobs[blockIdx.y];
  const unsigned nslab = (unsigned)job.K >> 4;
  const unsigned quads = (unsigned)(((long long)job.rows * job.K) >> 2);
  const bool vec = !job.trans && (job.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(job.src) & 15u) == 0;
  for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < quads; q += gridDim.x * 256u) {
    const unsigned l = q & 63u, blk = q >> 6;
    const unsigned ct = blk / nslab, t = blk - ct * nslab;
    const unsigned r = 16u * ct + (l & 15u), k = 16u * t + 4u * (l >> 4);
    float4 v;
    if (vec) {
      v = *reinterpret_cast<const float4*>(job.src + (long long)r * job.ld + k);
    } else if (job.trans) {
      const float* sp = job.src + (long long)k * job.ld + r;
      v = make_float4(sp[0], sp[job.ld], sp[2ll * job.ld], sp[3ll * job.ld]);
    } else {
      const float* sp = job.src + (long long)r * job.ld + k;
      v = make_float4(sp[0], sp[1], sp[2], sp[3]);
    }
    reinterpret_cast<float4*>(job.dst)[q] = v;
  }
}

constexpr int PANEL_NUM_CU = 256;

static int panel_chunk_cols(int N) {
  if (N % 256 == 0) return 256;
  if (N % 192 == 0) return 192;
  if (N % 128 == 0) return 128;
  if (N % 64 == 0) return 64;
  return 0;
}

// rows per panel.  Measured (scripts/panel_bench.py, profiles/r02_panel_gemm_bench.txt): 64-row panels (RT = 4) win on
// every DPOT shape - M = 8192 then gives exactly one workgroup per CU and column chunk; shorter panels that would fit two
// workgroups per CU (20 waves) are slower, taller ones re-balance worse.  Other heights only when they save a round.
static int panel_pick_rt(int M, int nchunks, int nc) {
  (void)nc;
  double best_cost = -1;
  int best = 4;
  const int order[5] = {4, 5, 3, 2, 1};
  for (int o = 0; o < 5; ++o) {
    const int rt = order[o];
    const long long panels = (M + 16 * rt - 1) / (16 * rt);
    const long long rounds = (panels * nchunks + PANEL_NUM_CU - 1) / PANEL_NUM_CU;
    const double cost = rounds * (16.0 * rt + 10.0) * (rt == 4 ? 1.0 : 1.08);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = rt;
    }
  }
  return best;
}

template <int NW, int ACTK>
static int panel_launch_rt(const PanelArgs& p, int rt, hipStream_t s) {
  const dim3 grid((unsigned)(p.nchunks * p.panels)), blk(64 * (NW + 2));
  switch (rt) {
    case 1: hipLaunchKernelGGL((gemm_panel_kernel<1, NW, ACTK>), grid, blk, 0, s, p); break;
    case 2: hipLaunchKernelGGL((gemm_panel_kernel<2, NW, ACTK>), grid, blk, 0, s, p); break;
    case 3: hipLaunchKernelGGL((gemm_panel_kernel<3, NW, ACTK>), grid, blk, 0, s, p); break;
    case 4: hipLaunchKernelGGL((gemm_panel_kernel<4, NW, ACTK>), grid, blk, 0, s, p); break;
    default: hipLaunchKernelGGL((gemm_panel_kernel<5, NW, ACTK>), grid, blk, 0, s, p); break;
  }
  return check_launch("gemm_panel_kernel");
}
template <int NW>
static int panel_launch_nw(const PanelArgs& p, int rt, hipStream_t s) {
  const bool gelu = p.act == DPOT_ACT_GELU || p.mode == DPOT_EPI_LINEAR;
  return gelu ? panel_launch_rt<NW, DPOT_ACT_GELU>(p, rt, s) : panel_launch_rt<NW, -1>(p, rt, s);
}

}  // namespace dpot

using namespace dpot;

extern "C" int dpot_gemm_panel_supported(int M, int N, int K) {
  return M > 0 && K >= 32 && K % 32 == 0 && panel_chunk_cols(N) != 0 ? 1 : 0;
}

extern "C" int dpot_panel_pack_weights(const dpot_pack_job* jobs_dev, int njobs, int max_elems, dpot_stream_t stream) {
  DPOT_REQUIRE(jobs_dev && njobs > 0 && njobs <= 65535 && max_elems > 0, "panel_pack_weights: bad argument");
  long long g = ((long long)max_elems / 4 + 255) / 256;
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(panel_pack_kernel, dim3((unsigned)g, njobs), dim3(256), 0, as_stream(stream), jobs_dev);
  return check_launch("panel_pack_kernel");
}

extern "C" int dpot_gemm_panel(const float* A, int lda, const float* Wpacked, const float* bias, const float* aux,
                               int ldaux, const float* res, int ldres, float* pre, int ldpre, float* C, int ldc, int M,
                               int N, int K, int act, int epi_mode, dpot_stream_t stream) {
  DPOT_REQUIRE(A && Wpacked && C, "gemm_panel: null operand");
  DPOT_REQUIRE(dpot_gemm_panel_supported(M, N, K), "gemm_panel: unsupported shape M=%d N=%d K=%d (K %% 32, N %% 64)", M, N, K);
  DPOT_REQUIRE(epi_mode == DPOT_EPI_LINEAR || epi_mode == DPOT_EPI_ACT || (epi_mode == DPOT_EPI_DACT && aux),
               "gemm_panel: bad epilogue mode");
  DPOT_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0 && ldc % 4 == 0 && (!aux || ldaux % 4 == 0) &&
                   (!res || ldres % 4 == 0) && (!pre || ldpre % 4 == 0),
               "gemm_panel: leading dimensions must be multiples of 4");
  DPOT_REQUIRE(aligned16(A) && aligned16(Wpacked) && aligned16(bias) && aligned16(aux) && aligned16(res) &&
                   aligned16(pre) && aligned16(C),
               "gemm_panel: pointers must be 16-byte aligned");
  PanelArgs p;
  p.A = A; p.W = Wpacked; p.bias = bias; p.aux = aux; p.res = res; p.pre = pre; p.C = C;
  p.lda = lda; p.ldaux = ldaux; p.ldres = ldres; p.ldpre = ldpre; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.act = act; p.mode = epi_mode;
  const int nc = panel_chunk_cols(N);
  p.nchunks = N / nc;
  const int rt = panel_pick_rt(M, p.nchunks, nc);
  p.panels = (M + 16 * rt - 1) / (16 * rt);
  hipStream_t s = as_stream(stream);
  switch (nc / 64) {
    case 1: return panel_launch_nw<2>(p, rt, s);
    case 2: return panel_launch_nw<4>(p, rt, s);
    case 3: return panel_launch_nw<6>(p, rt, s);
    default: return panel_launch_nw<8>(p, rt, s);
  }
}
