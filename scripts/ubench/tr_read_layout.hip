// unit check of the LDS image + ds_read_b64_tr_b16 addressing used by the bf16 weight-gradient kernels to read a ROW-form pack
// (token-major chunks: [tok/32][feat/16][64 chunks][8]: chunk l = (token l & 31, features 8 (l >> 5) .. + 7)) as the MFMA
// operand of a contraction over TOKENS: lane (f = l & 31, kg = l >> 5) must end up with tokens 8 kg .. 8 kg + 7 of feature f.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/tr_read_layout.hip -o scripts/ubench/tr_read_layout && ./tr_read_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));

// one 32-feature row tile x 32 tokens: two row-form blocks (feature block fb' = 0, 1) of 1 KiB each in global memory;
// LDS image of the tile: 128 slots of 16 B, slot(t, fb', fhalf) = (t >> 2) * 16 + (fb' * 2 + fhalf) * 4 + (t & 3)
__global__ void k(const unsigned short* __restrict__ g, unsigned short* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[2 * 512];
  const int L = threadIdx.x;
  // what the two LDS-DMA instructions of the tile do (lane-linear destination, permuted source)
  for (int i = 0; i < 2; ++i) {
    const int t = 16 * i + 4 * (L >> 4) + (L & 3), combo = (L >> 2) & 3, fb = combo >> 1, fh = combo & 1;
    const uint4 v = *reinterpret_cast<const uint4*>(g + fb * 512 + (fh * 32 + t) * 8);
    *reinterpret_cast<uint4*>(lds + (64 * i + L) * 8) = v;
  }
  __syncthreads();
  const int kg = L >> 5, fb = (L >> 4) & 1;
  const int lb = ((2 * kg) * 16 + (fb * 2 + ((L & 3) >> 1)) * 4 + ((L >> 2) & 3)) * 16 + (L & 1) * 8;   // bytes
  for (int ks = 0; ks < 2; ++ks)
    for (int r2 = 0; r2 < 2; ++r2) {
      const unsigned char* a = reinterpret_cast<const unsigned char*>(lds) + lb + ks * 1024 + r2 * 256;
      const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
      for (int j = 0; j < 4; ++j) out[((ks * 64 + L) * 2 + r2) * 4 + j] = (unsigned short)v[j];
    }
}

int main() {
  std::vector<unsigned short> h(1024), o(2 * 64 * 8);
  // value = token * 64 + feature (feature 0..31), stored in the row-form layout
  for (int fb = 0; fb < 2; ++fb)
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 8; ++e) h[fb * 512 + l * 8 + e] = (unsigned short)((l & 31) * 64 + fb * 16 + 8 * (l >> 5) + e);
  unsigned short *dg, *dout;
  hipMalloc(&dg, 2048); hipMalloc(&dout, o.size() * 2);
  hipMemcpy(dg, h.data(), 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dg, dout);
  hipMemcpy(o.data(), dout, o.size() * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int ks = 0; ks < 2; ++ks)
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 8; ++j) {
        const int tok = 16 * ks + 8 * (l >> 5) + j, f = l & 31;
        const int got = o[(ks * 64 + l) * 8 + j];
        if (got != tok * 64 + f) { if (bad < 8) printf("ks %d lane %d j %d: got (tok %d, f %d) want (tok %d, f %d)\n", ks, l, j, got / 64, got % 64, tok, f); ++bad; }
      }
  printf(bad ? "MISMATCH: %d elements\n" : "tr-read layout OK (%d)\n", bad);
  return bad != 0;
}
