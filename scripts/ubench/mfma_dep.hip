// micro-benchmark: v_mfma_f32_32x32x2_f32 issue rate vs number of independent accumulator chains per wave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 c[NACC];
  for (int j = 0; j < NACC; ++j) c[j] = (f32x16){0};
  float x = a + threadIdx.x, y = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c[j], 0, 0, 0);
  }
  float s = 0;
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) s += c[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(float* out, int wg_per_cu) {
  int grid = 256 * wg_per_cu, iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f, 0.5f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 16.0 * 2.0 * 32 * 32 * 2;
  double nm = (double)wg_per_cu * iters * 16.0;           // MFMAs per SIMD
  printf("acc/wave=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  -> %.1f cycles/MFMA/SIMD at 2.4 GHz\n", NACC, wg_per_cu, ms,
         flops / ms / 1e9, ms * 1e-3 * 2.4e9 / nm);
}
int main() {
  float* out; (void)hipMalloc(&out, 4 * 256 * 4096);
  for (int w = 1; w <= 4; w *= 2) { run<1>(out, w); run<2>(out, w); run<4>(out, w); }
  return 0;
}
