// micro-benchmark: v_mfma_f32_16x16x4_f32 issue rate from ONE wave per SIMD (256-thread workgroup, 1 per CU) with
// NACC independent accumulators, operands in registers (no memory traffic); optionally the same with LDS fragment
// reads (ds_read_b128) between the MFMA groups as in csrc/afno_mlp.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f32x4 c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c[i], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// 20 accumulators as 5 x 4 with distinct A / B registers per tile (the afno_mlp2 inner loop shape)
__global__ __launch_bounds__(256) void k16_tile(float* out, int iters, float a, float b) {
  f32x4 c[5][4];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) c[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float af[5], bf[4];
#pragma unroll
  for (int i = 0; i < 5; ++i) af[i] = a + threadIdx.x + i;
#pragma unroll
  for (int j = 0; j < 4; ++j) bf[j] = b + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], c[i][j], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += c[i][j][0] + c[i][j][1] + c[i][j][2] + c[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
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
template <typename F>
static void run(const char* name, F launch, double mfma_per_wave_iter, double flop_per_mfma, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 4000;
  launch(100);
  hipEventRecord(e0);
  launch(iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)iters * mfma_per_wave_iter;                  // MFMAs per wave = per SIMD (1 wave / SIMD)
  double wg_per_cu = grid / 256.0;
  printf("%-34s grid %4d  %.3f ms  %.1f TFLOP/s  %.1f ns/MFMA/SIMD = %.1f cycles at 2.4 GHz\n", name, grid, ms,
         (double)grid * 4 * n * flop_per_mfma / ms / 1e9, ms * 1e6 / (n * wg_per_cu), ms * 1e6 / (n * wg_per_cu) * 2.4);
}
int main() {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  for (int grid = 256; grid <= 512; grid *= 2) {
    run("16x16x4 f32, 4 acc", [&](int it) { hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 4, 2048, grid);
    run("16x16x4 f32, 8 acc", [&](int it) { hipLaunchKernelGGL(k16<8>, dim3(grid), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 8, 2048, grid);
    run("16x16x4 f32, 20 acc (one A/B pair)", [&](int it) { hipLaunchKernelGGL(k16<20>, dim3(grid), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 20, 2048, grid);
    run("16x16x4 f32, 5x4 tile", [&](int it) { hipLaunchKernelGGL(k16_tile, dim3(grid), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 20, 2048, grid);
    run("32x32x2 f32, 4 acc", [&](int it) { hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 4, 4096, grid);
  }
  return 0;
}
