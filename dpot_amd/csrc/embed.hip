// embed.hip - the P x P / stride-P patch convolution of DPOT's PatchEmbed (models/dpot.py:198-202, first conv of
// `proj`) as an IMPLICIT GEMM over the input field, forward and weight gradient:
//
//     Hpre[(b,px,py,t), n] = sum_{c<4,i,j} x[b, px*8+i, py*8+j, t, c] * W0[n, c, i, j]  +  bt[(px,py,t), n]
//     Hh = act(Hpre)
//     dW0[n, c, i, j]     = sum_{b,px,py,t} dHpre[(b,px,py,t), n] * x[b, px*8+i, py*8+j, t, c]           (c < 4)
//
// Round 1 materialised the patch matrix A0[B*tok*T, (C+3)*64] (147 MB at DPOT-Tiny B = 32: written once, read by the
// forward GEMM and again by the weight-gradient GEMM - 0.44 GB of HBM traffic around 3 GFLOP).  Here the rows are gathered
// from x itself: for a fixed kernel row i, the pixels of NP = 4 neighbouring patches are ONE contiguous run of
// NP*8*T*4 floats in x ([B, X, Y, T, C] with C = 4 innermost), so a group of four patches arrives as 8 contiguous LDS-DMA
// pieces and the MFMA A operand of k-quad (i, j) = the 4 channels of pixel (i, j) is a single ds_read_b32 per lane.
// The three unit-grid channels of the reference's input (x, y, t coordinates) do not depend on the batch: their
// contribution is folded into the bias table bt[(tok,t), n] once per optimiser step by a 2560 x 192 x 36 GEMM on the
// host side (functional.embed_derived), and their weight gradient is that GEMM's transpose on the batch-summed dHpre.
//
//   * forward: persistent workgroups, one group of 4 patches (4*T rows -> ceil(4T/16) row tiles = compute waves, plus
//     two loader waves that issue all LDS-DMA and are the only ones to wait on vmcnt) per iteration,
//     double-buffered groups (the DMA of group g+1 runs under the MFMAs of group g, one barrier per group); the weights
//     sit in LDS in fragment order ([i][j/4][col tile][lane][j%4]: one ds_read_b128 serves 4 k-quads);
//     v_mfma_f32_16x16x4_f32, fused bias-table add + activation, both outputs stored.
//   * weight gradient: the contraction runs over the rows; wave w owns kernel rows i = 2w, 2w+1 (12 accumulator tiles),
//     accumulates over all groups of the workgroup, partials are summed in a fixed order by a second launch.
//
// Algorithmic bytes: x once (21 MB) + Hpre/Hh (2 x 11.8 MB) forward; x + dHpre backward.
#include "common.h"

namespace dpot {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int EM_P = 8;          // patch size
constexpr int EM_C = 4;          // data channels (one float4 per pixel and frame)
constexpr int EM_NP = 4;         // patches per group
constexpr int EM_CT = 3;         // 16-column tiles of the hidden width (hid <= 48)
constexpr int EM_LW = 2;         // loader waves of the forward kernel
constexpr int EM_TMAX = 10;      // frames: group = 8 * NP*8*T*4 floats <= 40 KiB (DPOT: T_in = 10)
constexpr int EM_KD = EM_C * EM_P * EM_P;                    // 256 data columns of the conv weight
constexpr int EM_WFL = EM_P * 2 * EM_CT * 256;               // floats of the fragment-ordered weights (48 KiB)
constexpr int EM_GFL = EM_P * EM_NP * EM_P * EM_TMAX * EM_C; // floats of one group buffer at T = TMAX

struct EmbedArgs {
  const float* x;      // [B, X, Y, T, 4]
  const float* wf;     // forward: fragment-ordered weights [8][2][3][64][4]
  const float* bt;     // forward: [tok*T, hidp] bias + unit-grid contribution
  const float* dh;     // wgrad: dHpre [M0, hidp]
  float* hpre;         // forward outputs [M0, hidp]
  float* hh;
  float* part;         // wgrad: [gridDim.x][48][256] partial sums
  int X, Y, T, h, w, hidp, act, ngroups, wq;   // wq = w / NP
};

__device__ __forceinline__ void em_glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// LDS-DMA of the x pieces of group g into buf: 8 runs (kernel rows i) of `run` floats each, contiguous in x
__device__ __forceinline__ void em_issue_x(const EmbedArgs& a, int g, float* buf, int run, int tid, int nthr) {
  const int pyq = g % a.wq;
  const int t2 = g / a.wq;
  const int px = t2 % a.h, b = t2 / a.h;
  const long long rowf = (long long)a.Y * a.T * EM_C;
  const float* src = a.x + ((long long)b * a.X + px * EM_P) * rowf + (long long)pyq * run;
  const int per = run >> 2;                            // 16-byte chunks per run
  int i = tid / per, o = tid - i * per;                // (run index, chunk within the run), advanced incrementally
  for (int c = tid; c < EM_P * per; c += nthr) {      // wave-uniform trip count up to the ragged last wave-instruction
    // destination: the wave-uniform base of this instruction (the hardware adds lane * 16 bytes); chunk c lands at
    // float 4*c = i*run + 4*o, so a wave-instruction may straddle two runs
    em_glds16(src + i * rowf + 4 * o, buf + 4 * (c - (tid & 63)));
    o += nthr;
    while (o >= per) { o -= per; ++i; }
  }
}

// LDS-DMA of `nchunk` consecutive 16-byte chunks src -> dst (dst wave-uniform per instruction, lane-linear)
__device__ __forceinline__ void em_issue_lin(const float* src, float* dst, int nchunk, int tid, int nthr) {
  for (int c = tid; c < nchunk; c += nthr) em_glds16(src + 4 * c, dst + 4 * (c - (tid & 63)));
}

__device__ __forceinline__ void em_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool GELU>
__global__ __launch_bounds__(64 * (3 + EM_LW)) void embed_fwd_kernel(const EmbedArgs a) {
  constexpr int BTF = EM_NP * EM_TMAX * 48;            // floats of one bias-table tile buffer
  __shared__ __attribute__((aligned(16))) float lds[2 * EM_GFL + 2 * BTF + EM_WFL];
  float* const btl = lds + 2 * EM_GFL;
  float* const wl = btl + 2 * BTF;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int T = a.T, T4 = T * EM_C, hidp = a.hidp;
  const int run = EM_NP * EM_P * T4;                   // floats of one kernel row of a group
  const int rows = EM_NP * T;
  const int btn = rows * hidp;                         // floats of a group's bias-table / output tile (contiguous rows)
  for (int c = tid; c < EM_WFL / 4; c += nthr)
    reinterpret_cast<float4*>(wl)[c] = reinterpret_cast<const float4*>(a.wf)[c];

  // this lane's A row: r = (patch p, frame t); rows past 4T are padding (clamped, never stored)
  const int r = 16 * wave + (lane & 15), kq = lane >> 4;
  const int rc = r < rows ? r : 0;
  const int p = rc / T, t = rc - p * T;
  const int abase = (p * EM_P * T + t) * EM_C + kq;

  auto issue = [&](int g, int buf) __attribute__((always_inline)) {
    em_issue_x(a, g, lds + buf * EM_GFL, run, tid, nthr);
    const int pyq = g % a.wq, px = (g / a.wq) % a.h;
    em_issue_lin(a.bt + (long long)(px * a.w + pyq * EM_NP) * T * hidp, btl + buf * BTF, btn >> 2, tid, nthr);
  };

  // the LAST EM_LW waves only load: vmcnt counts stores as well as loads, so a wave that both streams its output rows
  // out and waits for the next group's DMA sits through the write latency of its own stores once per group (measured:
  // 7.6 us per group against 2.6 us of MFMA work).  The loaders' counters see nothing but the DMA; the compute waves
  // never wait on vmcnt at all and their stores drain under the next group.  (Also tried: the loaders streaming the
  // finished tile out while the compute waves go straight on - slower, 52-54 us against 49.)
  const int ncw = (nthr >> 6) - EM_LW;
  const bool loader = wave >= ncw;
  const int ltid = tid - 64 * ncw;                     // thread index among the loaders
  auto issue_l = [&](int g, int buf) __attribute__((always_inline)) {
    em_issue_x(a, g, lds + buf * EM_GFL, run, ltid, 64 * EM_LW);
    const int pyq = g % a.wq, px = (g / a.wq) % a.h;
    em_issue_lin(a.bt + (long long)(px * a.w + pyq * EM_NP) * T * hidp, btl + buf * BTF, btn >> 2, ltid, 64 * EM_LW);
  };

  int g = blockIdx.x;
  if (g < a.ngroups) issue(g, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                     // weights + first group landed
  int cur = 0;
  for (; g < a.ngroups; g += gridDim.x) {
    // the other buffers are free: every wave passed the barrier below after its reads of group g - 1
    const int gn = g + gridDim.x;
    if (loader) {
      if (gn < a.ngroups) issue_l(gn, cur ^ 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                 // group g + 1 landed; the compute waves are done with group g
      cur ^= 1;
      continue;
    }
    const float* xs = lds + cur * EM_GFL + abase;
    f32x4 acc[EM_CT];
#pragma unroll
    for (int ct = 0; ct < EM_CT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < EM_P; ++i) {
      float av[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = xs[i * run + j * T4];
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) {
        f32x4 bf[EM_CT];
#pragma unroll
        for (int ct = 0; ct < EM_CT; ++ct)
          bf[ct] = *reinterpret_cast<const f32x4*>(wl + ((i * 2 + jq) * EM_CT + ct) * 256 + lane * 4);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ct = 0; ct < EM_CT; ++ct)
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * jq + jj], bf[ct][jj], acc[ct], 0, 0, 0);
      }
    }
    // epilogue in place: the bias-table tile of the group becomes its Hpre tile (each element is touched by exactly one
    // lane; the 16 rows of a wave are one contiguous piece of the tile), then every wave streams its rows out with
    // 16-byte stores - Hpre as is, Hh through the activation
    float* tile = btl + cur * BTF;
    const int n0 = lane & 15;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 16 * wave + 4 * kq + e;
#pragma unroll
      for (int ct = 0; ct < EM_CT; ++ct) {
        const int n = ct * 16 + n0;
        if (n < hidp && row < rows) tile[row * hidp + n] += acc[ct][e];
      }
    }
    em_wave_sync();
    {
      const int pyq = g % a.wq;
      const int t2 = g / a.wq;
      const int px = t2 % a.h, b = t2 / a.h;
      const long long m0 = ((long long)b * a.h * a.w + px * a.w + pyq * EM_NP) * T;
      const int r0 = 16 * wave;
      const int nr = rows - r0 < 16 ? rows - r0 : 16;                // rows of this wave (may be <= 0 never: waves = tiles)
      const int nq = (nr * hidp) >> 2;                               // float4 pieces
      const float4* src = reinterpret_cast<const float4*>(tile + r0 * hidp);
      float4* o1 = reinterpret_cast<float4*>(a.hpre + (m0 + r0) * hidp);
      float4* o2 = reinterpret_cast<float4*>(a.hh + (m0 + r0) * hidp);
      for (int q = lane; q < nq; q += 64) {
        const float4 v = src[q];
        o1[q] = v;
        float4 u;
        if constexpr (GELU) {
          u = make_float4(gelu_fwd(v.x), gelu_fwd(v.y), gelu_fwd(v.z), gelu_fwd(v.w));
        } else {
          u = make_float4(act_fwd(a.act, v.x), act_fwd(a.act, v.y), act_fwd(a.act, v.z), act_fwd(a.act, v.w));
        }
        o2[q] = u;
      }
    }
    __syncthreads();                                   // group g + 1 landed (loader waves); reads of group g are done
    cur ^= 1;
  }
}

// weight gradient: 4 waves, wave w owns kernel rows i = 2w, 2w+1; contraction over the 4T rows of each group.
// Ring of three group buffers; waves 0-1 load the even iterations' groups, waves 2-3 the odd ones, each two iterations
// ahead - so a loader's `vmcnt(0)` at the top of ITS iteration waits for exactly the group that is needed, while the
// other pair's DMA (next group) stays in flight: two groups (~90 KiB per CU) are always on their way.
__global__ __launch_bounds__(256) void embed_wgrad_kernel(const EmbedArgs a) {
  constexpr int DHF = EM_NP * EM_TMAX * 48 + 64;       // floats of one dHpre buffer (+ slack for the padded columns)
  __shared__ __attribute__((aligned(16))) float lds[3 * EM_GFL + 3 * DHF];
  float* const dhl = lds + 3 * EM_GFL;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int T = a.T, T4 = T * EM_C;
  const int run = EM_NP * EM_P * T4;
  const int rows = EM_NP * T;
  const int dhn = rows * a.hidp;                       // floats of a group's dHpre tile (contiguous rows)
  const int kq = lane >> 4, n0 = lane & 15;
  const int jj = n0 >> 2, cch = n0 & 3;                // B operand column = (j%4, channel)
  const int inv = (65536 + T - 1) / T;                 // row / T for row < 48, T <= 12
  const int ltid = tid & 127, lpar = wave >> 1;        // loader pair: thread index within the pair, parity it serves

  auto issue = [&](int g, int buf) __attribute__((always_inline)) {
    em_issue_x(a, g, lds + buf * EM_GFL, run, ltid, 128);
    const int pyq = g % a.wq;
    const int t2 = g / a.wq;
    const int px = t2 % a.h, b = t2 / a.h;
    const long long m0 = ((long long)b * a.h * a.w + px * a.w + pyq * EM_NP) * T;
    em_issue_lin(a.dh + m0 * a.hidp, dhl + buf * DHF, dhn >> 2, ltid, 128);     // 16-byte aligned: hidp % 4 == 0
  };

  f32x4 acc[2][EM_CT][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int nt = 0; nt < EM_CT; ++nt)
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) acc[s][nt][jh] = f32x4{0.f, 0.f, 0.f, 0.f};

  // iteration k handles group blockIdx.x + k * gridDim.x from ring buffer k % 3
  const int g0 = blockIdx.x, gs = gridDim.x;
  if (g0 + lpar * gs < a.ngroups) issue(g0 + lpar * gs, lpar);       // pair 0: iteration 0, pair 1: iteration 1
  int buf = 0;
  for (int k = 0; g0 + k * gs < a.ngroups; ++k) {
    if ((k & 1) == lpar) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this pair's DMA of iteration k
    __syncthreads();                                   // group k landed; reads of iteration k - 1 are done
    if ((k & 1) == lpar && g0 + (k + 2) * gs < a.ngroups)
      issue(g0 + (k + 2) * gs, buf == 0 ? 2 : buf - 1);              // (k + 2) % 3: the buffer iteration k - 1 used
    const float* xs = lds + buf * EM_GFL + (2 * wave) * run + jj * T4 + cch;
    const float* dl = dhl + buf * DHF + n0;
#pragma unroll 2
    for (int q = 0; q < T; ++q) {                      // 4T rows = T k-quads
      const int row = 4 * q + kq;
      const int p = (row * inv) >> 16, t = row - p * T;
      const int xo = (p * EM_P * T + t) * EM_C;
      float af[EM_CT], bf[2][2];
#pragma unroll
      for (int nt = 0; nt < EM_CT; ++nt) af[nt] = dl[row * a.hidp + nt * 16];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) bf[s][jh] = xs[s * run + xo + 4 * jh * T4];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int nt = 0; nt < EM_CT; ++nt)
#pragma unroll
          for (int jh = 0; jh < 2; ++jh)
            acc[s][nt][jh] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[nt], bf[s][jh], acc[s][nt][jh], 0, 0, 0);
    }
    buf = buf == 2 ? 0 : buf + 1;
  }
  // partial dW0[n][k = c*64 + i*8 + j] of this workgroup
  float* out = a.part + (long long)blockIdx.x * 48 * EM_KD;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int i = 2 * wave + s;
#pragma unroll
    for (int nt = 0; nt < EM_CT; ++nt)
#pragma unroll
      for (int jh = 0; jh < 2; ++jh)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = nt * 16 + 4 * kq + e;
          out[n * EM_KD + cch * 64 + i * 8 + 4 * jh + jj] = acc[s][nt][jh][e];
        }
  }
}

// dw0[n, 0:256] (row stride ldw) = sum over workgroups of the partials.  One workgroup = 16 outputs x 16 slices of the
// partial list: every thread sums its slice with all loads in flight, the slices are combined in a fixed order.
__global__ __launch_bounds__(256) void embed_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, int hid,
                                                                 float* __restrict__ dw0, int ldw) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + o;
  const bool ok = idx < hid * EM_KD;
  const int n = ok ? idx / EM_KD : 0, k = ok ? idx - n * EM_KD : 0;
  const int per = (nparts + 15) >> 4;
  const int s0 = sl * per, s1 = min(nparts, s0 + per);
  float v = 0.f;
#pragma unroll 16
  for (int s = s0; s < s1; ++s) v += part[((long long)s * 48 + n) * EM_KD + k];
  red[sl][o] = v;
  __syncthreads();
  if (sl == 0 && ok) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][o];
    dw0[(long long)n * ldw + k] = t;
  }
}

// w0 [hid, C+3, 8, 8] -> fragment order [i][j/4][ct][lane = (n%16, c)][j%4], zero for n >= hid
__global__ __launch_bounds__(256) void embed_pack_w0_kernel(const float* __restrict__ w0, int hid, int K0,
                                                            float* __restrict__ wf) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= EM_WFL) return;
  const int jj = idx & 3, lane = (idx >> 2) & 63, blk = idx >> 8;
  const int ct = blk % EM_CT, ij = blk / EM_CT;
  const int jq = ij & 1, i = ij >> 1;
  const int n = ct * 16 + (lane & 15), c = lane >> 4;
  wf[idx] = n < hid ? w0[(long long)n * K0 + c * 64 + i * 8 + 4 * jq + jj] : 0.f;
}

}  // namespace dpot

using namespace dpot;

static int embed_grid(int ngroups) {
  int g = ngroups < 256 ? ngroups : 256;
  return g < 1 ? 1 : g;
}

extern "C" int dpot_embed_supported(int C, int P, int T, int hid, int w) {
  return C == EM_C && P == EM_P && T >= 1 && T <= EM_TMAX && hid >= 1 && hid <= 16 * EM_CT && w % EM_NP == 0;
}

extern "C" int dpot_embed_pack_w0(const float* w0, int hid, float* wfrag, dpot_stream_t stream) {
  DPOT_REQUIRE(w0 && wfrag && hid >= 1 && hid <= 16 * EM_CT, "embed_pack_w0: bad argument");
  hipLaunchKernelGGL(embed_pack_w0_kernel, dim3(EM_WFL / 256), dim3(256), 0, as_stream(stream), w0, hid,
                     (EM_C + 3) * EM_P * EM_P, wfrag);
  return check_launch("embed_pack_w0_kernel");
}

extern "C" int dpot_embed_wfrag_elems(void) { return EM_WFL; }

static int embed_fill(EmbedArgs& a, const float* x, int B, int X, int Y, int T, int hidp) {
  DPOT_REQUIRE(x && B > 0 && X > 0 && Y > 0 && X % EM_P == 0 && Y % EM_P == 0, "embed: bad field shape");
  DPOT_REQUIRE(T >= 1 && T <= EM_TMAX && hidp % 4 == 0 && hidp >= 4 && hidp <= 16 * EM_CT, "embed: unsupported T / width");
  DPOT_REQUIRE(aligned16(x), "embed: x must be 16-byte aligned");
  a.x = x;
  a.X = X; a.Y = Y; a.T = T;
  a.h = X / EM_P; a.w = Y / EM_P;
  DPOT_REQUIRE(a.w % EM_NP == 0, "embed: patch-grid width must be a multiple of 4");
  a.wq = a.w / EM_NP;
  a.hidp = hidp;
  const long long ng = (long long)B * a.h * a.wq;
  DPOT_REQUIRE(ng < (1ll << 31) && (long long)a.h * a.w * T * hidp < (1ll << 31), "embed: too many patches");
  a.ngroups = (int)ng;
  return DPOT_OK;
}

extern "C" int dpot_embed_fwd(const float* x, const float* wfrag, const float* btab, float* hpre, float* hh, int B,
                              int X, int Y, int T, int hidp, int act, dpot_stream_t stream) {
  EmbedArgs a{};
  int rc = embed_fill(a, x, B, X, Y, T, hidp);
  if (rc) return rc;
  DPOT_REQUIRE(wfrag && btab && hpre && hh && aligned16(wfrag), "embed_fwd: null / misaligned pointer");
  a.wf = wfrag; a.bt = btab; a.hpre = hpre; a.hh = hh; a.act = act;
  const int waves = (EM_NP * T + 15) / 16 + EM_LW;     // row tiles + the loader waves
  if (act == DPOT_ACT_GELU)
    hipLaunchKernelGGL(embed_fwd_kernel<true>, dim3(embed_grid(a.ngroups)), dim3(64 * waves), 0, as_stream(stream), a);
  else
    hipLaunchKernelGGL(embed_fwd_kernel<false>, dim3(embed_grid(a.ngroups)), dim3(64 * waves), 0, as_stream(stream), a);
  return check_launch("embed_fwd_kernel");
}

extern "C" int dpot_embed_wgrad_ws_elems(int B, int X, int Y) {
  const long long ng = (long long)B * (X / EM_P) * (Y / EM_P / EM_NP);
  return embed_grid((int)(ng < 256 ? ng : 256)) * 48 * EM_KD;
}

extern "C" int dpot_embed_wgrad(const float* x, const float* dhpre, float* workspace, float* dw0, int ldw, int hid,
                                int B, int X, int Y, int T, int hidp, dpot_stream_t stream) {
  EmbedArgs a{};
  int rc = embed_fill(a, x, B, X, Y, T, hidp);
  if (rc) return rc;
  DPOT_REQUIRE(dhpre && workspace && dw0 && aligned16(dhpre) && hid >= 1 && hid <= hidp && ldw >= EM_KD,
               "embed_wgrad: bad argument");
  a.dh = dhpre; a.part = workspace;
  const int grid = embed_grid(a.ngroups);
  hipLaunchKernelGGL(embed_wgrad_kernel, dim3(grid), dim3(256), 0, as_stream(stream), a);
  rc = check_launch("embed_wgrad_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(embed_wgrad_reduce_kernel, dim3((hid * EM_KD + 15) / 16), dim3(256), 0, as_stream(stream),
                     (const float*)workspace, grid, hid, dw0, ldw);
  return check_launch("embed_wgrad_reduce_kernel");
}
