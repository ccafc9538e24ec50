// gemm_epi.h - the fused GEMM epilogue shared by the MFMA GEMM kernels (gemm.hip, gemm_split.h, gemm_bf16p.hip):
// bias, pre-activation save, activation or act'(aux) product, row-mapped residual, accumulate; 16-byte accesses through
// a per-wave LDS slab (32x32 accumulator fragments of the v_mfma_*_32x32x* family).
#pragma once
#include "common.h"

namespace dpot {

typedef float f32x16 __attribute__((ext_vector_type(16)));


struct EpiArgs {
  float* C;
  int ldc;
  long long sC;
  const float* bias;
  long long sBias;
  const float* aux;
  int ldaux;
  long long sAux;
  float* pre;
  int ldpre;
  long long sPre;
  const float* res;
  int ldres;
  int res_div, res_mod;
  long long sRes;
  int act, mode, accumulate;
  int M, N;
};

struct GemmArgs {
  const float* A;
  const float* B;
  int M, N, K;
  int lda, ldb;
  long long sA, sB;
  int batch, splits, ktiles_per_split;
  int tilesM, tilesN;
  int evec;  // epilogue may use 16-byte accesses
  float* ws;  // split-K partials [split][batch][M][N] (+ [split][batch][L] column-sum partials)
  // fused bias gradient: column sums over k of one operand (wgrad: dW = dY^T X and db = colsum(dY) read the same dY)
  float* cs_out;      // [batch][L], L = M (cs_of == 1, A stored [K,M]) or N (cs_of == 2, B stored [K,N])
  long long sCs;
  int cs_of;
  EpiArgs e;
};

__device__ __forceinline__ int res_row(const EpiArgs& e, int m) {
  if (e.res_div > 1) m = m / e.res_div;
  if (e.res_mod > 0) m = m % e.res_mod;
  return m;
}

// scalar epilogue (split-K reduction kernel)
__device__ __forceinline__ void epi_store(const EpiArgs& e, int b, int m, int n, float v) {
  if (e.bias) v += e.bias[b * e.sBias + n];
  if (e.pre) e.pre[b * e.sPre + (long long)m * e.ldpre + n] = v;
  if (e.mode == DPOT_EPI_ACT) {
    v = act_fwd(e.act, v);
  } else if (e.mode == DPOT_EPI_DACT) {
    v *= act_bwd(e.act, e.aux[b * e.sAux + (long long)m * e.ldaux + n]);
  }
  if (e.res) v += e.res[b * e.sRes + (long long)res_row(e, m) * e.ldres + n];
  float* c = e.C + b * e.sC + (long long)m * e.ldc + n;
  if (e.accumulate) v += *c;
  *c = v;
}

// ---- tile epilogue --------------------------------------------------------------------------------
// The 32x32 accumulator fragment of a wave (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) is
// bounced through a per-wave LDS slab so that (a) the epilogue is a compact 4-trip loop instead of 64 inlined
// copies of the activation code and (b) each lane owns 4 consecutive columns: bias / aux / residual loads and the
// C store are 16-byte accesses, 128 B contiguous per row.
constexpr int EPI_LD = 36;  // floats per staged row (16-B aligned rows, conflict-free b128 reads)

struct Vec4 {
  float v[4];
};
__device__ __forceinline__ Vec4 ld4(const float* p, bool vec) {
  Vec4 r;
  if (vec) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
    r.v[0] = p[0]; r.v[1] = p[1]; r.v[2] = p[2]; r.v[3] = p[3];
  }
  return r;
}

__device__ __forceinline__ void epi_fragment(const EpiArgs& e, int evec, int b, int m0f, int n0f, const f32x16& acc,
                                             float* stage, int lane) {
  const int li = lane & 31, kh = lane >> 5;
  const int c4 = (lane & 7) * 4;
  const int n = n0f + c4;
  // vector path: the bias of this lane's four columns is the same for all four row groups - ONE load per fragment,
  // issued before the staging round trip (inside the loop it was a dependent L2 access per trip)
  Vec4 bq;
#pragma unroll
  for (int k = 0; k < 4; ++k) bq.v[k] = 0.f;
  if (evec && e.bias) bq = ld4(e.bias + b * e.sBias + (n < e.N ? n : e.N - 4), true);
#pragma unroll
  for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * kh) * EPI_LD + li] = acc[r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // (the loop stays rolled: unrolled - so that the four trips' aux / residual loads are in flight together - the bf16
  // panel GEMM's epilogues got SLOWER, fc1 forward 139 -> 177 us, and gemm.hip takes 4.5 minutes to compile)
#pragma unroll 1
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3);
    const int m = m0f + row;
    const float4 t = *reinterpret_cast<const float4*>(&stage[row * EPI_LD + c4]);
    float v[4] = {t.x, t.y, t.z, t.w};
    if (evec) {
      // N % 4 == 0: the 4 columns are all valid or all invalid; loads come from clamped addresses
      const bool ok = (m < e.M) && (n < e.N);
      const int mc = m < e.M ? m : e.M - 1;
      const int nc = n < e.N ? n : e.N - 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += bq.v[k];
      if (e.pre && ok)
        *reinterpret_cast<float4*>(e.pre + b * e.sPre + (long long)mc * e.ldpre + nc) =
            make_float4(v[0], v[1], v[2], v[3]);
      if (e.mode == DPOT_EPI_ACT) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = act_fwd(e.act, v[k]);
      } else if (e.mode == DPOT_EPI_DACT) {
        const Vec4 q = ld4(e.aux + b * e.sAux + (long long)mc * e.ldaux + nc, true);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= act_bwd(e.act, q.v[k]);
      }
      if (e.res) {
        const Vec4 q = ld4(e.res + b * e.sRes + (long long)res_row(e, mc) * e.ldres + nc, true);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += q.v[k];
      }
      float* c = e.C + b * e.sC + (long long)mc * e.ldc + nc;
      if (e.accumulate) {
        const Vec4 q = ld4(c, true);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += q.v[k];
      }
      if (ok) *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      // odd shapes (N or a leading dimension not a multiple of 4): scalar, element-wise predicated
      if (m < e.M) {
        for (int k = 0; k < 4; ++k)
          if (n + k < e.N) epi_store(e, b, m, n + k, v[k]);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}


}  // namespace dpot

// gemm_tn.hip: the weight-gradient kernel behind dpot_gemm_f32 (transA, !transB, split-K); -1 = descriptor not eligible
int dpot_gemm_tn_try(const dpot_gemm_desc* d, hipStream_t s);
