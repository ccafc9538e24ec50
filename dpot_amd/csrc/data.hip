// data.hip - device side of the input pipeline (reference: utils/griddataset.py:88-101 pad_data and
// :125-174 __getitem__): for every sample of a batch, in ONE launch,
//     raw [H, W, T, C]  --bilinear resize to res x res (F.interpolate(mode='bilinear'), align_corners=False)-->
//     --channel pad with ones up to n_channels--> --temporal window [t0, t0 + t_in + t_ar)-->  xx [res,res,t_in,Cmax],
//                                                                                              yy [res,res,t_ar,Cmax]
// Only the t_in + t_ar frames of the window are resized (the reference resizes the whole trajectory and slices it
// afterwards).  Samples of one batch may come from different datasets (different H, W, T, C): the per-sample geometry
// travels by value in the launch packet.  HBM-bound: reads the window of the raw sample once, writes xx / yy once.
#include "common.h"

namespace dpot {

constexpr int DATA_MAX_JOBS = 64;
struct WindowJobs {
  const float* src[DATA_MAX_JOBS];
  int H[DATA_MAX_JOBS], W[DATA_MAX_JOBS], T[DATA_MAX_JOBS], C[DATA_MAX_JOBS], t0[DATA_MAX_JOBS];
};

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

// grid (chunks of the res*res pixel grid, job); one thread = one output pixel, all window frames and channels
__global__ __launch_bounds__(256) void resize_pad_window_kernel(const WindowJobs jobs, float* __restrict__ xx,
                                                                float* __restrict__ yy, int res, int t_in, int t_ar,
                                                                int Cmax, int job0) {
  const int j = blockIdx.y;
  const int H = jobs.H[j], W = jobs.W[j], T = jobs.T[j], C = jobs.C[j], t0 = jobs.t0[j];
  const float* __restrict__ src = jobs.src[j];
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= res * res) return;
  const int oy = pix / res, ox = pix - oy * res;         // oy indexes the FIRST spatial axis (H), ox the second (W)
  const float sh = (float)H / (float)res, sw = (float)W / (float)res;
  int h0, h1, w0, w1;
  float hl0, hl1, wl0, wl1;
  src_index(oy, sh, H, h0, h1, hl0, hl1);
  src_index(ox, sw, W, w0, w1, wl0, wl1);
  const long long TC = (long long)T * C;
  const float* p00 = src + ((long long)h0 * W + w0) * TC;
  const float* p01 = src + ((long long)h0 * W + w1) * TC;
  const float* p10 = src + ((long long)h1 * W + w0) * TC;
  const float* p11 = src + ((long long)h1 * W + w1) * TC;
  const long long b = job0 + j;
  float* xo = xx + (b * res * res + pix) * (long long)t_in * Cmax;
  float* yo = yy ? yy + (b * res * res + pix) * (long long)t_ar * Cmax : nullptr;
  const int nt = t_in + t_ar;
  for (int t = 0; t < nt; ++t) {
    float* o = t < t_in ? xo + (long long)t * Cmax : (yo ? yo + (long long)(t - t_in) * Cmax : nullptr);
    if (!o) break;
    const long long so = (long long)(t0 + t) * C;
    for (int c = 0; c < Cmax; ++c) {
      float v = 1.f;                                     // channels the dataset does not have are ones (griddataset.py:98)
      if (c < C) {
        const float v00 = p00[so + c], v01 = p01[so + c], v10 = p10[so + c], v11 = p11[so + c];
        v = hl0 * (wl0 * v00 + wl1 * v01) + hl1 * (wl0 * v10 + wl1 * v11);
      }
      o[c] = v;
    }
  }
}

}  // namespace dpot

using namespace dpot;

extern "C" int dpot_resize_pad_window(const dpot_sample_desc* samples, int nsamples, float* xx, float* yy, int res,
                                      int t_in, int t_ar, int n_channels, dpot_stream_t stream) {
  DPOT_REQUIRE(samples && nsamples > 0 && xx && res > 0 && t_in > 0 && t_ar >= 0 && n_channels > 0,
               "resize_pad_window: bad argument");
  DPOT_REQUIRE(t_ar == 0 || yy != nullptr, "resize_pad_window: t_ar > 0 needs yy");
  for (int j0 = 0; j0 < nsamples; j0 += DATA_MAX_JOBS) {
    const int nj = nsamples - j0 < DATA_MAX_JOBS ? nsamples - j0 : DATA_MAX_JOBS;
    WindowJobs jobs;
    for (int j = 0; j < nj; ++j) {
      const dpot_sample_desc& s = samples[j0 + j];
      DPOT_REQUIRE(s.data && s.H > 0 && s.W > 0 && s.T > 0 && s.C > 0 && s.C <= n_channels,
                   "resize_pad_window: sample %d has a bad shape (C=%d, n_channels=%d)", j0 + j, s.C, n_channels);
      DPOT_REQUIRE(s.t0 >= 0 && s.t0 + t_in + t_ar <= s.T,
                   "resize_pad_window: sample %d: window [%d, %d) exceeds its %d frames", j0 + j, s.t0,
                   s.t0 + t_in + t_ar, s.T);
      jobs.src[j] = s.data; jobs.H[j] = s.H; jobs.W[j] = s.W; jobs.T[j] = s.T; jobs.C[j] = s.C; jobs.t0[j] = s.t0;
    }
    hipLaunchKernelGGL(resize_pad_window_kernel, dim3((unsigned)cdiv(res * res, 256), nj), dim3(256), 0,
                       as_stream(stream), jobs, xx, yy, res, t_in, t_ar, n_channels, j0);
    int rc = check_launch("resize_pad_window_kernel");
    if (rc) return rc;
  }
  return DPOT_OK;
}
