// data.hip - device side of the input pipeline (reference: utils/griddataset.py:88-101 pad_data and
// :125-174 __getitem__): for every sample of a batch, in ONE launch,
//     raw [H, W, T, C]  --bilinear resize to res x res (F.interpolate(mode='bilinear'), align_corners=False)-->
//     --channel pad with ones up to n_channels--> --temporal window [t0, t0 + t_in + t_ar)-->  xx [res,res,t_in,Cmax],
//                                                                                              yy [res,res,t_ar,Cmax]
// Only the t_in + t_ar frames of the window are resized (the reference resizes the whole trajectory and slices it
// afterwards).  Samples of one batch may come from different datasets (different H, W, T, C): the per-sample geometry
// is a small table in DEVICE memory that the caller uploads together with the raw samples (one H2D copy per batch).
// HBM-bound: reads the window of the raw sample once, writes xx / yy once.
#include "common.h"

namespace dpot {

// ATen's area_pixel_compute_source_index (align_corners = false, not cubic) in float, as upsample_bilinear2d uses it
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float s = scale * (dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

// grid (chunks of the output frames, job); one thread = one (pixel, frame) of xx or yy = n_channels contiguous floats, so
// adjacent lanes write adjacent 4*Cmax-byte pieces (fully coalesced; one float4 per lane at Cmax = 4) and read adjacent
// frames of the same four source pixels
__global__ __launch_bounds__(256) void resize_pad_window_kernel(const dpot_sample_desc* __restrict__ jobs,
                                                                float* __restrict__ xx, float* __restrict__ yy, int res,
                                                                int t_in, int t_ar, int Cmax, int dh, int dw) {
  // (the job table lives in device memory: a by-value table indexed by blockIdx.y is copied to scratch by every thread)
  const int j = blockIdx.y;
  const dpot_sample_desc job = jobs[j];
  const int H = job.H, W = job.W, T = job.T, C = job.C, t0 = job.t0;
  const float* __restrict__ src = job.data;
  if (H <= 0 || W <= 0 || C <= 0 || C > Cmax || t0 < 0 || t0 + t_in + t_ar > T) return;   // malformed entry: skip
  // dh, dw: the strided sub-sampling x[::dh, ::dw] the reference applies AFTER the resize (griddataset.py:170-172):
  // output pixel (oy, ox) is pixel (oy * dh, ox * dw) of the res x res image
  const int ro_h = (res + dh - 1) / dh, ro_w = (res + dw - 1) / dw;
  const long long npix = (long long)ro_h * ro_w;
  const long long nx = npix * t_in, total = npix * (t_in + t_ar);
  const float sh = (float)H / (float)res, sw = (float)W / (float)res;
  const long long TC = (long long)T * C;
  const long long b = j;
  for (long long item = blockIdx.x * 256ll + threadIdx.x; item < total; item += (long long)gridDim.x * 256) {
    int pix, t;
    float* o;
    if (item < nx) {
      pix = (int)(item / t_in);
      t = (int)(item - (long long)pix * t_in);
      o = xx + ((b * npix + pix) * t_in + t) * Cmax;
    } else {
      const long long it = item - nx;
      pix = (int)(it / t_ar);
      const int ty = (int)(it - (long long)pix * t_ar);
      t = t_in + ty;
      o = yy + ((b * npix + pix) * t_ar + ty) * Cmax;
    }
    const int oy = pix / ro_w, ox = pix - oy * ro_w;     // oy indexes the FIRST spatial axis (H), ox the second (W)
    int h0, h1, w0, w1;
    float hl0, hl1, wl0, wl1;
    src_index(oy * dh, sh, H, h0, h1, hl0, hl1);
    src_index(ox * dw, sw, W, w0, w1, wl0, wl1);
    const long long so = (long long)(t0 + t) * C;
    const float* p00 = src + ((long long)h0 * W + w0) * TC + so;
    const float* p01 = src + ((long long)h0 * W + w1) * TC + so;
    const float* p10 = src + ((long long)h1 * W + w0) * TC + so;
    const float* p11 = src + ((long long)h1 * W + w1) * TC + so;
    float v[4];
    for (int c0 = 0; c0 < Cmax; c0 += 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        v[e] = 1.f;                                      // channels the dataset does not have are ones (griddataset.py:98)
        if (c < C) v[e] = hl0 * (wl0 * p00[c] + wl1 * p01[c]) + hl1 * (wl0 * p10[c] + wl1 * p11[c]);
      }
      if (Cmax == 4) {
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c0 + e < Cmax) o[c0 + e] = v[e];
      }
    }
  }
}

}  // namespace dpot

using namespace dpot;

extern "C" int dpot_resize_pad_window(const dpot_sample_desc* samples_dev, int nsamples, float* xx, float* yy, int res,
                                      int t_in, int t_ar, int n_channels, int down_h, int down_w,
                                      dpot_stream_t stream) {
  DPOT_REQUIRE(down_h >= 1 && down_w >= 1 && down_h <= res && down_w <= res, "resize_pad_window: bad down-sampling factors");
  DPOT_REQUIRE(samples_dev && nsamples > 0 && nsamples <= 65535 && xx && res > 0 && t_in > 0 && t_ar >= 0 &&
                   n_channels > 0,
               "resize_pad_window: bad argument");
  DPOT_REQUIRE(t_ar == 0 || yy != nullptr, "resize_pad_window: t_ar > 0 needs yy");
  DPOT_REQUIRE(n_channels != 4 || (aligned16(xx) && aligned16(yy)), "resize_pad_window: outputs must be 16-byte aligned");
  long long blocks = ((long long)((res + down_h - 1) / down_h) * ((res + down_w - 1) / down_w) * (t_in + t_ar) + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(resize_pad_window_kernel, dim3((unsigned)blocks, nsamples), dim3(256), 0, as_stream(stream),
                     samples_dev, xx, yy, res, t_in, t_ar, n_channels, down_h, down_w);
  return check_launch("resize_pad_window_kernel");
}
