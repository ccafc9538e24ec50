// micro-benchmark: does the f32 MFMA rate depend on the operand DATA (power management) or on the instruction shape?
// 1 workgroup of 256 threads per CU, 20 (16x16x4) or 5 (32x32x2) accumulators per wave, operands in registers,
// filled from a buffer that holds zeros / constants / random normals.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k16(const float* __restrict__ src, float* out, int iters) {
  f32x4 c[5][4];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) c[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 af[5], bf[4];
  const float* s = src + (blockIdx.x * 256 + threadIdx.x) * 64;
#pragma unroll
  for (int i = 0; i < 5; ++i) af[i] = *(const f32x4*)(s + 4 * i);
#pragma unroll
  for (int j = 0; j < 4; ++j) bf[j] = *(const f32x4*)(s + 32 + 4 * j);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], c[i][j], 0, 0, 0);
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) r += c[i][j][0] + c[i][j][1] + c[i][j][2] + c[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k32(const float* __restrict__ src, float* out, int iters) {
  f32x16 c[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
  f32x4 af[2], bf[2];
  const float* s = src + (blockIdx.x * 256 + threadIdx.x) * 64;
  af[0] = *(const f32x4*)(s); af[1] = *(const f32x4*)(s + 4);
  bf[0] = *(const f32x4*)(s + 32); bf[1] = *(const f32x4*)(s + 36);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], c[i][j], 0, 0, 0);
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) r += c[i][j][q];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
static float gauss() { float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = rand() / (float)RAND_MAX; return sqrtf(-2 * logf(u)) * cosf(6.2831853f * v); }
int main() {
  const int grid = 256, n = grid * 256 * 64;
  float *h = (float*)malloc(n * 4), *src, *out;
  (void)hipMalloc(&src, n * 4); (void)hipMalloc(&out, grid * 256 * 4);
  const char* names[3] = {"zeros", "constant 1.5/0.5", "random normal"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int i = 0; i < n; ++i) h[i] = mode == 0 ? 0.f : mode == 1 ? ((i & 32) ? 0.5f : 1.5f) : gauss() * 0.05f;
    (void)hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; ++which) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      const int iters = 1000;
      auto launch = [&](int it) { if (which == 0) hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, src, out, it);
                                  else hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, src, out, it); };
      launch(50);
      (void)hipEventRecord(e0); launch(iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double per_iter = which == 0 ? 80 : 16, flop = which == 0 ? 2048 : 4096;
      printf("%-18s %-9s %.3f ms  %.1f TFLOP/s  %.1f cycles/MFMA/SIMD at 2.4 GHz\n", names[mode], which == 0 ? "16x16x4" : "32x32x2", ms,
             (double)grid * 4 * iters * per_iter * flop / ms / 1e9, ms * 1e6 / (iters * per_iter) * 2.4);
    }
  }
  return 0;
}
