// micro-benchmark: sustained rate of v_mfma_f32_32x32x16_bf16 (operands in registers) - zeros vs random data,
// 1 or 2 waves per SIMD, with and without LDS fragment reads (ds_read_b128 per MFMA operand) in the loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int LDSR>
__global__ __launch_bounds__(512) void kb(const uint4* __restrict__ src, float* out, int iters) {
  __shared__ uint4 sm[64 * 16];
  f32x16 c[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
  const int lane = threadIdx.x & 63;
  for (int q = threadIdx.x; q < 64 * 16; q += blockDim.x) sm[q] = src[q];
  __syncthreads();
  uint4 ra[2], rb[2];
  ra[0] = src[lane]; ra[1] = src[64 + lane]; rb[0] = src[128 + lane]; rb[1] = src[192 + lane];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (LDSR) {      // LDSR fragment reads per 4 MFMAs
#pragma unroll
        for (int q = 0; q < LDSR; ++q) {
          const uint4 v = sm[((it + s + q) & 15) * 64 + lane];
          if (q & 1) rb[(q >> 1) & 1] = v; else ra[(q >> 1) & 1] = v;
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[i]), __builtin_bit_cast(bf16x8, rb[j]), c[i][j], 0, 0, 0);
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) r += c[i][j][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static float gauss() { float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = rand() / (float)RAND_MAX; return sqrtf(-2 * logf(u)) * cosf(6.2831853f * v); }
static unsigned short bf(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
int main() {
  const int n = 64 * 16 * 8;
  unsigned short* h = (unsigned short*)malloc(n * 2);
  uint4* src; float* out;
  (void)hipMalloc(&src, n * 2); (void)hipMalloc(&out, 256 * 512 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < n; ++i) h[i] = mode == 0 ? 0 : bf(gauss() * 0.05f);
    (void)hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice);
    for (int waves = 4; waves <= 8; waves += 4)
      for (int ldsr = 0; ldsr <= 4; ldsr += 2) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int iters = 2000;
        auto launch = [&](int it) {
          if (ldsr == 0) hipLaunchKernelGGL(kb<0>, dim3(256), dim3(64 * waves), 0, 0, src, out, it);
          else if (ldsr == 2) hipLaunchKernelGGL(kb<2>, dim3(256), dim3(64 * waves), 0, 0, src, out, it);
          else hipLaunchKernelGGL(kb<4>, dim3(256), dim3(64 * waves), 0, 0, src, out, it);
        };
        launch(50);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        launch(iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = 256.0 * waves * iters * 16 * 32768.0;
        printf("%s, %d waves/CU, %d ds_read_b128 per 4 MFMAs: %.3f ms  %.0f TFLOP/s\n", mode ? "random" : "zeros", waves, ldsr,
               ms, fl / ms / 1e9);
      }
  }
  return 0;
}
