// micro-benchmark: pure v_mfma_f32_32x32x2_f32 issue rate (no memory traffic) - what clock does the chip sustain?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float x = a + threadIdx.x, y = b;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, c3, 0, 0, 0);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
    int grid = 256 * wg_per_cu, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 100, 1.0f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * 2.0 * 32 * 32 * 2;
    printf("wg/cu=%d  %.3f ms  %.1f TFLOP/s  (implied clock %.2f GHz at 64 FLOP/clk/SIMD)\n", wg_per_cu, ms,
           flops / ms / 1e9, flops / ms / 1e9 / (256.0 * 4 * 64) * 1e3 / 1e3);
  }
  return 0;
}
