// gemm_bf16p_common.h - the argument block, LDS-DMA / wait helpers and the fragment epilogues (staged, in-accumulator-layout
// "direct", transposed store) of the bf16 GEMM kernels in csrc/gemm_bf16p.hip (split out in round 6 so that a second
// translation unit - the measured-and-rejected 256 x 256 kernel - could share them).
#pragma once
#include <type_traits>

#include "common.h"
#include "gemm_epi.h"

namespace dpot {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

struct Bf16pArgs {
  const unsigned short* A;   // packed [Mtiles][K/16][512]
  const unsigned short* W;   // packed [N/32][K/16][512]
  int M, N, K, tilesM, tilesN;
  int splits, slabs_per_split;   // split-K (weight gradients: K = tokens): blockIdx.y = split, partials go to ws
  float* ws;                     // [splits][M][N]
  // optional extra outputs of the epilogue (1-plane packs of the FINAL output, so that the next GEMMs need no separate
  // pack pass over it; M % 32 == 0): row form (A operand of the next data GEMM), transposed form (operand of the weight
  // gradient), per-32-row partial column sums [M/32][N] (bias gradient).  e.C may then be NULL (no fp32 store).
  uint4* out_rows;
  uint4* out_trans;
  float* cs_part;
  // the activation DERIVATIVE as bf16 in the row-form pack layout of the output ([M/32][N/16][64 chunks][8]): written
  // by an EPI_ACT launch (dact_out: act'(pre) - what the backward multiplies by; replaces the fp32 pre-activation save,
  // half the bytes and no second activation evaluation), read by an EPI_DACT launch (dact_in instead of e.aux)
  uint4* dact_out;
  const unsigned short* dact_in;
  int super_r, super_c;          // tile rasterisation: the 32 concurrent tiles of an XCD form super_r x super_c blocks
  EpiArgs e;
};

template <int N>
__device__ __forceinline__ void bwait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_waitcnt vmcnt(n) for the wave-uniform counts the loaders use (multiples of 3, 5 or 6; the immediate must be a constant)
__device__ __forceinline__ void bwait_vm_dyn(int n) {
  switch (n) {
    case 0: bwait_vm<0>(); break;
    case 3: bwait_vm<3>(); break;
    case 5: bwait_vm<5>(); break;
    case 6: bwait_vm<6>(); break;
    case 10: bwait_vm<10>(); break;
    case 15: bwait_vm<15>(); break;
    case 9: bwait_vm<9>(); break;
    case 12: bwait_vm<12>(); break;
    case 18: bwait_vm<18>(); break;
    case 24: bwait_vm<24>(); break;
    default: bwait_vm<0>(); break;          // (not a count this kernel produces: wait for everything)
  }
}
__device__ __forceinline__ void bglds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);   // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, r);
}

// epilogue of one 32x32 fragment with PACKED outputs: like epi_fragment (bias, pre-activation save, act / act'(aux),
// residual, optional fp32 store) but the final values go back into the LDS slab, from which the wave emits the bf16
// row-form chunks (row, 8 consecutive columns), the transposed chunks (column, 8 consecutive rows) and the 32-row partial
// column sums.  Vector path only (N % 4 == 0, aligned), M % 32 == 0.
__device__ __forceinline__ unsigned long long ld8(const void* p) { return *reinterpret_cast<const unsigned long long*>(p); }

__device__ __forceinline__ void epi_fragment_pack(const Bf16pArgs& p, int m0f, int n0f, const f32x16& acc, float* stage,
                                                  int lane) {
  const EpiArgs& e = p.e;
  const int li = lane & 31, kh = lane >> 5;
  const int c4 = (lane & 7) * 4;
  const int n = n0f + c4;
  Vec4 bq;
#pragma unroll
  for (int k = 0; k < 4; ++k) bq.v[k] = 0.f;
  if (e.bias) bq = ld4(e.bias + n, true);
#pragma unroll
  for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * kh) * EPI_LD + li] = acc[r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // the bias of this lane's four columns is the same for all four row groups: ONE load per fragment, issued before the
  // staging round trip (inside the loop it was a dependent L2 access per trip).  The loop itself stays rolled (unrolled:
  // 139 -> 177 us for fc1 forward at DPOT-M)
#pragma unroll 1
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3);
    const int m = m0f + row;
    const float4 t = *reinterpret_cast<const float4*>(&stage[row * EPI_LD + c4]);
    float v[4] = {t.x + bq.v[0], t.y + bq.v[1], t.z + bq.v[2], t.w + bq.v[3]};
    if (e.pre) *reinterpret_cast<float4*>(e.pre + (long long)m * e.ldpre + n) = make_float4(v[0], v[1], v[2], v[3]);
    if (e.mode == DPOT_EPI_ACT) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = act_fwd(e.act, v[k]);
    } else if (e.mode == DPOT_EPI_DACT) {
      const Vec4 q = ld4(e.aux + (long long)m * e.ldaux + n, true);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] *= act_bwd(e.act, q.v[k]);
    }
    if (e.res) {
      const Vec4 q = ld4(e.res + (long long)m * e.ldres + n, true);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += q.v[k];
    }
    if (e.C) *reinterpret_cast<float4*>(e.C + (long long)m * e.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(&stage[row * EPI_LD + c4]) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifndef PB_ABL_NOSTORE
  if (p.out_rows) {     // 128 chunks (row, column octet) = two 1 KiB blocks (16 columns each): instruction s2 writes block s2
#pragma unroll          // whole - lane l = chunk l of the block = (row l & 31, octet l >> 5): 64 consecutive 16-byte chunks
    for (int s2 = 0; s2 < 2; ++s2) {
      const int row = lane & 31, oc = 2 * s2 + (lane >> 5);
      const float4 x0 = *reinterpret_cast<const float4*>(&stage[row * EPI_LD + 8 * oc]);
      const float4 x1 = *reinterpret_cast<const float4*>(&stage[row * EPI_LD + 8 * oc + 4]);
      const int m = m0f + row, nn = n0f + 8 * oc;
      p.out_rows[((long long)(m >> 5) * (p.N >> 4) + (nn >> 4)) * 64 + (m & 31) + 32 * ((nn >> 3) & 1)] =
          make_uint4(pack2(x0.x, x0.y), pack2(x0.z, x0.w), pack2(x1.x, x1.y), pack2(x1.z, x1.w));
    }
  }
#ifndef PB_ABL_NOTRANS
  if (p.out_trans) {    // 128 chunks (column, row octet): id = lane + 64 s -> column = id & 31, octet = id >> 5
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int id = lane + 64 * s2, col = id & 31, oc = id >> 5;
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = stage[(8 * oc + i) * EPI_LD + col];
      const int f = n0f + col, m = m0f + 8 * oc;
      p.out_trans[((long long)(f >> 5) * (p.M >> 4) + (m >> 4)) * 64 + (f & 31) + 32 * ((m >> 3) & 1)] =
          make_uint4(pack2(x[0], x[1]), pack2(x[2], x[3]), pack2(x[4], x[5]), pack2(x[6], x[7]));
    }
  }
#endif
#endif
  if (p.cs_part) {      // column sums of the 32 rows, fixed order: 16 rows per half-wave, then the two halves
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += stage[(16 * kh + i) * EPI_LD + li];
    a += __shfl_xor(a, 32);
    if (kh == 0) p.cs_part[(long long)(m0f >> 5) * p.N + n0f + li] = a;
  }
  __builtin_amdgcn_wave_barrier();
}

// The same packed-output epilogue computed IN THE ACCUMULATOR LAYOUT (round 3; the launches of the bf16 channel MLP:
// no fp32 pre-activation, no residual).  A lane of a 32x32 fragment holds ONE column (lane & 31) and 16 rows
// (row(r) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)), so
//   * bias is one scalar per lane, and the activation + derivative run on 16 independent register values (the staged
//     form above evaluates 4 per trip of a rolled loop behind an LDS round trip: the epilogue was bound by that latency
//     chain - 54 of 142 us at fc1 forward of DPOT-M - not by its stores);
//   * the derivative pack is written / read in FRAGMENT ORDER - [M/32][N/32][2][64 lanes][8 bf16], lane's r = 8 s .. 8 s + 7
//     in half s - straight from / into registers: producer (fc1 forward) and consumer (fc2 data gradient) have the same
//     fragment grid, nobody else reads it;
//   * the transposed pack's chunks are (column, 8 consecutive rows): a lane has rows 8 j + 4 kh .. + 3, its partner in the
//     other half-wave the other four - one v_permlane32_swap per packed pair completes them, no LDS;
//   * the column sums are in-lane sums + one cross-half add;
//   * only the row-form pack (and the optional fp32 store) needs the transpose through LDS: one round trip.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// 16-byte store of a pack chunk, non-temporal: the packs are read by the NEXT kernel (67 MB each at DPOT-M) and only push
// the operand panels out of L2 (fc2 data gradient at DPOT-M 120.4 -> 112.4 us; -DPB_NO_NT_STORES: plain stores)
__device__ __forceinline__ void st_chunk(uint4* p, unsigned a, unsigned b, unsigned c, unsigned d) {
#ifndef PB_NO_NT_STORES
  const u32x4_t v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p));
#else
  *p = make_uint4(a, b, c, d);
#endif
}
__device__ __forceinline__ float bf_lo(unsigned q) { return __uint_as_float(q << 16); }
__device__ __forceinline__ float bf_hi(unsigned q) { return __uint_as_float(q & 0xffff0000u); }

__device__ __forceinline__ void epi_fragment_direct(const Bf16pArgs& p, int m0f, int n0f, const f32x16& acc, float* stage,
                                                    int lane) {
  const EpiArgs& e = p.e;
  const int li = lane & 31, kh = lane >> 5;
  const long long frag = (long long)(m0f >> 5) * (p.N >> 5) + (n0f >> 5);
  const float bq = e.bias ? e.bias[n0f + li] : 0.f;
  const bool to_lds = p.out_rows || e.C;
  const bool want_d = e.mode == DPOT_EPI_ACT && p.dact_out;
  unsigned q[8];                                       // act' pack of this lane (EPI_DACT)
  if (e.mode == DPOT_EPI_DACT) {                       // host-checked: dact_in
    const uint4* dp = reinterpret_cast<const uint4*>(p.dact_in) + frag * 128 + lane;
    const uint4 q0 = dp[0], q1 = dp[64];
    q[0] = q0.x; q[1] = q0.y; q[2] = q0.z; q[3] = q0.w; q[4] = q1.x; q[5] = q1.y; q[6] = q1.z; q[7] = q1.w;
  }
  // four values at a time (register budget: the wave still holds its other three accumulators - 8 at a time spilled 400
  // registers in the two-workgroup kernel); what stays live across the groups is packed: w = the values, dq = act'
  unsigned w[8], dq[8];
  float cs = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v[4], d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = acc[4 * g + k] + bq;
    if (e.mode == DPOT_EPI_ACT) {
      if (want_d) {
#ifdef PB_ABL_NOACT
        if (true) {
#pragma unroll
          for (int k = 0; k < 4; ++k) d[k] = v[k] * 0.5f;
        } else
#endif
        if (e.act == DPOT_ACT_GELU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) gelu_val_der(v[k], v[k], d[k]);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) { d[k] = act_bwd(e.act, v[k]); v[k] = act_fwd(e.act, v[k]); }
        }
        dq[2 * g] = pack2(d[0], d[1]);
        dq[2 * g + 1] = pack2(d[2], d[3]);
      } else if (e.act == DPOT_ACT_GELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = gelu_fwd(v[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = act_fwd(e.act, v[k]);
      }
    } else if (e.mode == DPOT_EPI_DACT) {
      v[0] *= bf_lo(q[2 * g]); v[1] *= bf_hi(q[2 * g]); v[2] *= bf_lo(q[2 * g + 1]); v[3] *= bf_hi(q[2 * g + 1]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) cs += v[k];            // fixed order: r ascending
    if (to_lds) {                                      // rows (r & 3) + 8 (r >> 2) + 4 kh, r = 4 g + k
#pragma unroll
      for (int k = 0; k < 4; ++k) stage[(k + 8 * g + 4 * kh) * EPI_LD + li] = v[k];
    }
    w[2 * g] = pack2(v[0], v[1]);
    w[2 * g + 1] = pack2(v[2], v[3]);
    __builtin_amdgcn_sched_barrier(0);
  }
#ifndef PB_ABL_NOSTORE
#ifndef PB_ABL_NODACT
  if (want_d) {
    uint4* dp = p.dact_out + frag * 128 + lane;
    st_chunk(dp, dq[0], dq[1], dq[2], dq[3]);
    st_chunk(dp + 64, dq[4], dq[5], dq[6], dq[7]);
  }
#endif
#ifndef PB_ABL_NOTRANS
  if (p.out_trans) {
    // packed pairs w[k] = rows (2k, 2k+1 of this lane's 16).  Lane kh = 0 completes row octets 0 and 2, kh = 1 octets 1
    // and 3: swap the upper half-wave of X = w[a] with the lower half-wave of Y = w[a + 2], a in {0, 1, 4, 5} - then
    // (X, Y) = (own, partner's) in the lower half, (partner's, own) in the upper, i.e. rows ascending in both
    unsigned X[4] = {w[0], w[1], w[4], w[5]};
    unsigned Y[4] = {w[2], w[3], w[6], w[7]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const auto sw = __builtin_amdgcn_permlane32_swap(X[k], Y[k], false, false);
      X[k] = sw[0];
      Y[k] = sw[1];
    }
    // octet kh -> block (n0f / 32, m0f / 16), chunk li + 32 kh = lane; octet 2 + kh -> the next block: each instruction
    // writes one whole 1 KiB block
    uint4* tp = p.out_trans + ((long long)(n0f >> 5) * (p.M >> 4) + (m0f >> 4)) * 64 + lane;
    st_chunk(tp, X[0], X[1], Y[0], Y[1]);
    st_chunk(tp + 64, X[2], X[3], Y[2], Y[3]);
  }
#endif
#endif
  if (p.cs_part) {      // column sums of the fragment's 32 rows: 16 in the lane (fixed order), then the two half-waves
    cs += __shfl_xor(cs, 32);
    if (kh == 0) p.cs_part[(long long)(m0f >> 5) * p.N + n0f + li] = cs;
  }
  if (to_lds) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if !defined(PB_ABL_NOSTORE) && !defined(PB_ABL_NOROWS)
    if (p.out_rows) {   // instruction s2 writes block s2 whole: lane l = chunk (row l & 31, column octet 2 s2 + (l >> 5))
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int oc = 2 * s2 + kh;
        const float4 x0 = *reinterpret_cast<const float4*>(&stage[li * EPI_LD + 8 * oc]);
        const float4 x1 = *reinterpret_cast<const float4*>(&stage[li * EPI_LD + 8 * oc + 4]);
        st_chunk(p.out_rows + ((long long)(m0f >> 5) * (p.N >> 4) + (n0f >> 4) + s2) * 64 + lane, pack2(x0.x, x0.y),
                 pack2(x0.z, x0.w), pack2(x1.x, x1.y), pack2(x1.z, x1.w));
      }
    }
#endif
    if (e.C) {
      const int c4 = (lane & 7) * 4;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3);
        *reinterpret_cast<float4*>(e.C + (long long)(m0f + row) * e.ldc + n0f + c4) =
            *reinterpret_cast<const float4*>(&stage[row * EPI_LD + c4]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace dpot
