// micro-benchmark: how many VALU instructions hide behind one v_mfma_f32_32x32x16_bf16 when a SIMD holds ONE wave
// (the regime of the bf16x6 GEMM's 128x128 tile).  MODE 0: independent v_fma fillers, 1: the dependent
// cvt_pk/shl/and/sub split chain (2 chains), 2: independent cvt_pk fillers.  NF = fillers per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

template <int MODE, int NF>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(a + threadIdx.x + i); y[i] = (__bf16)(b + i); }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = a * (i + 1) + threadIdx.x;
  unsigned sink = 0;
  for (int it = 0; it < iters; ++it) {
#define FILL()                                                                                 \
  do {                                                                                         \
    if constexpr (MODE == 0) {                                                                 \
      _Pragma("unroll") for (int q = 0; q < NF; ++q) f[q & 7] = __builtin_fmaf(f[q & 7], a, b); \
    } else if constexpr (MODE == 2) {                                                          \
      _Pragma("unroll") for (int q = 0; q < NF; ++q) sink ^= pk(f[q & 7], f[(q + 1) & 7]);     \
    } else {                                                                                   \
      _Pragma("unroll") for (int q = 0; q < NF / 9; ++q) {                                     \
        float u = f[2 * (q & 3)], v = f[2 * (q & 3) + 1];                                      \
        unsigned p1 = pk(u, v);                                                                \
        u -= __uint_as_float(p1 << 16); v -= __uint_as_float(p1 & 0xffff0000u);                \
        unsigned p2 = pk(u, v);                                                                \
        u -= __uint_as_float(p2 << 16); v -= __uint_as_float(p2 & 0xffff0000u);                \
        unsigned p3 = pk(u, v);                                                                \
        sink ^= p1 ^ p2 ^ p3;                                                                  \
        f[2 * (q & 3)] += 1.0f;                                                                \
      }                                                                                        \
    }                                                                                          \
  } while (0)
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0);
    FILL();
    __builtin_amdgcn_sched_barrier(0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, c1, 0, 0, 0);
    FILL();
    __builtin_amdgcn_sched_barrier(0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, c2, 0, 0, 0);
    FILL();
    __builtin_amdgcn_sched_barrier(0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, c3, 0, 0, 0);
    FILL();
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)sink;
}

template <int MODE, int NF>
void run(float* out, int wg_per_cu) {
  int grid = 256 * wg_per_cu, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NF>), dim3(grid), dim3(256), 0, 0, out, 100, 1.0f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NF>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nm = (double)iters * 4 * wg_per_cu;        // MFMAs per SIMD
  printf("mode %d  fillers/MFMA %2d  wg/cu %d : %.3f ms  %.1f ns/MFMA/SIMD  (%.0f TF bf16)\n", MODE, NF, wg_per_cu, ms,
         ms * 1e6 / nm, (double)grid * 4 * iters * 4 * 2.0 * 32 * 32 * 16 / ms / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  for (int w = 1; w <= 2; ++w) {
    run<0, 0>(out, w); run<0, 2>(out, w); run<0, 4>(out, w); run<0, 5>(out, w); run<0, 6>(out, w); run<0, 8>(out, w);
    run<0, 12>(out, w);
    run<2, 4>(out, w); run<2, 6>(out, w);
    run<1, 9>(out, w); run<1, 18>(out, w);
  }
  return 0;
}
