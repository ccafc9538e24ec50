// dft.hip - rfft2 / irfft2 (norm="ortho") of channels-last latent fields as two in-LDS direct DFT passes.
//
// Why not an FFT library plan: the latent grids are 16x16 (128^2 / patch 8) or 32x32 (256^2), with the
// channel axis innermost (stride E between spatial points).  One workgroup stages a [h*w, CC] channel
// slab in LDS, does the row transform (real -> half complex) and the column transform (complex ->
// complex) as dense DFTs straight out of LDS and writes the kept modes in the planar-per-block layout the
// mixer GEMM consumes.  HBM traffic is exactly one read of the field and one write of the spectrum - the
// roofline for this op; 2*(h+w) extra MACs per point are free next to that.
//
// Works for any h, w (no power-of-two requirement: img_size/patch_size is arbitrary in the reference).
#include "common.h"
#include "dft_fast.h"

namespace dpot {

__device__ __forceinline__ void make_twiddles(float* c, float* s, int n) {
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const double a = (double)(2 * t) / (double)n;  // angle / pi
    c[t] = (float)cospi(a);
    s[t] = (float)sinpi(a);
  }
}

__device__ __forceinline__ float col_weight(int colw, int ky, int w) {
  if (!colw) return 1.f;
  if (ky == 0) return 1.f;
  if ((w & 1) == 0 && ky == (w >> 1)) return 1.f;
  return 2.f;
}

// x[B,h,w,E] -> spec[B,mx,my,nb,2,bs]
__global__ __launch_bounds__(256) void rfft2_kernel(const float* __restrict__ x, float* __restrict__ spec, int h,
                                                    int w, int E, int nb, int mx, int my, int CC, int colw,
                                                    float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* cw = sm;
  float* sw = cw + w;
  float* ch = sw + w;
  float* sh = ch + h;
  float* in = sm + ((2 * w + 2 * h + 3) & ~3);
  float* Z = in + h * w * CC;

  const int b = blockIdx.y, c0 = blockIdx.x * CC;
  const int tid = threadIdx.x;
  const int c = tid % CC, g = tid / CC, G = 256 / CC;
  const int bs = E / nb;

  make_twiddles(cw, sw, w);
  make_twiddles(ch, sh, h);
  const int total = h * w * CC;
  const float* xb = x + (long long)b * h * w * E + c0;
  for (int idx = tid; idx < total; idx += 256) in[idx] = xb[(long long)(idx / CC) * E + c];
  __syncthreads();

  // pass 1: along w (real -> half-complex, only ky < my)
  for (int item = g; item < h * my; item += G) {
    const int xr = item / my, ky = item % my;
    const float* row = in + xr * w * CC + c;
    float re = 0.f, im = 0.f;
    int ti = 0;
    for (int y = 0; y < w; ++y) {
      const float v = row[y * CC];
      re = fmaf(v, cw[ti], re);
      im = fmaf(-v, sw[ti], im);
      ti += ky;
      if (ti >= w) ti -= w;
    }
    Z[(item * 2 + 0) * CC + c] = re;
    Z[(item * 2 + 1) * CC + c] = im;
  }
  __syncthreads();

  // pass 2: along h (complex -> complex, only kx < mx), scale, store planar per channel block
  const int chn = c0 + c;
  const int blk = chn / bs, ci = chn % bs;
  for (int item = g; item < mx * my; item += G) {
    const int kx = item / my, ky = item % my;
    float re = 0.f, im = 0.f;
    int ti = 0;
    for (int xr = 0; xr < h; ++xr) {
      const float zr = Z[((xr * my + ky) * 2 + 0) * CC + c];
      const float zi = Z[((xr * my + ky) * 2 + 1) * CC + c];
      const float cc_ = ch[ti], ss_ = sh[ti];
      re = fmaf(zr, cc_, fmaf(zi, ss_, re));
      im = fmaf(zi, cc_, fmaf(-zr, ss_, im));
      ti += kx;
      if (ti >= h) ti -= h;
    }
    const float wgt = scale * col_weight(colw, ky, w);
    const long long o = ((((long long)b * mx + kx) * my + ky) * nb + blk) * 2 * bs + ci;
    spec[o] = re * wgt;
    spec[o + bs] = im * wgt;
  }
}

// spec[B,mx,my,nb,2,bs] (+ res[B,h,w,E]) -> y[B,h,w,E]
__global__ __launch_bounds__(256) void irfft2_kernel(const float* __restrict__ spec, const float* __restrict__ res,
                                                     float* __restrict__ y, int h, int w, int E, int nb, int mx,
                                                     int my, int CC, int colw, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* cw = sm;
  float* sw = cw + w;
  float* ch = sw + w;
  float* sh = ch + h;
  float* S = sm + ((2 * w + 2 * h + 3) & ~3);
  float* U = S + mx * my * 2 * CC;

  const int b = blockIdx.y, c0 = blockIdx.x * CC;
  const int tid = threadIdx.x;
  const int c = tid % CC, g = tid / CC, G = 256 / CC;
  const int bs = E / nb;
  const int chn = c0 + c;
  const int blk = chn / bs, ci = chn % bs;

  make_twiddles(cw, sw, w);
  make_twiddles(ch, sh, h);
  {
    const int total = mx * my * 2 * CC;
    for (int idx = tid; idx < total; idx += 256) {
      const int part = (idx / CC) & 1, mode = idx / (2 * CC);
      S[idx] = spec[((((long long)b * mx * my + mode) * nb + blk) * 2 + part) * bs + ci];
    }
  }
  __syncthreads();

  // pass A: along h, U[x,ky] = sum_kx S[kx,ky] e^{+2 pi i kx x / h}; fold in the column weight
  for (int item = g; item < h * my; item += G) {
    const int xr = item / my, ky = item % my;
    float ur = 0.f, ui = 0.f;
    int ti = 0;
    for (int kx = 0; kx < mx; ++kx) {
      const float sr = S[((kx * my + ky) * 2 + 0) * CC + c];
      const float si = S[((kx * my + ky) * 2 + 1) * CC + c];
      const float cc_ = ch[ti], ss_ = sh[ti];
      ur = fmaf(sr, cc_, fmaf(-si, ss_, ur));
      ui = fmaf(sr, ss_, fmaf(si, cc_, ui));
      ti += xr;
      if (ti >= h) ti -= h;
    }
    const float wgt = col_weight(colw, ky, w);
    U[(item * 2 + 0) * CC + c] = ur * wgt;
    U[(item * 2 + 1) * CC + c] = ui * wgt;
  }
  __syncthreads();

  // pass B: along w, y[x,yy] = scale * sum_ky Re(U[x,ky] e^{+2 pi i ky yy / w}) (+ res)
  const long long base = (long long)b * h * w * E + c0 + c;
  for (int item = g; item < h * w; item += G) {
    const int xr = item / w, yy = item % w;
    float acc = 0.f;
    int ti = 0;
    for (int ky = 0; ky < my; ++ky) {
      const float ur = U[((xr * my + ky) * 2 + 0) * CC + c];
      const float ui = U[((xr * my + ky) * 2 + 1) * CC + c];
      acc = fmaf(ur, cw[ti], fmaf(-ui, sw[ti], acc));
      ti += yy;
      if (ti >= w) ti -= w;
    }
    const long long o = base + (long long)item * E;
    float v = acc * scale;
    if (res) v += res[o];
    y[o] = v;
  }
}

static int pick_cc(int E, long long floats_per_channel, int tw_floats) {
  const long long budget = 150 * 1024 / 4;  // floats; LDS is 160 KiB per CU, keep headroom
  for (int cc = 64; cc >= 1; cc >>= 1) {
    if (E % cc) continue;
    if (floats_per_channel * cc + tw_floats <= budget) return cc;
  }
  return 0;
}

// AFNO weight packing -------------------------------------------------------------------------------
__device__ __forceinline__ void afno_pack_one(const float* __restrict__ w, const float* __restrict__ b,
                                              float* __restrict__ wbig, float* __restrict__ bbig, int nb, int bs,
                                              long long idx);
__global__ void afno_pack_kernel(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ wbig,
                                 float* __restrict__ bbig, int nb, int bs) {
  afno_pack_one(w, b, wbig, bbig, nb, bs, blockIdx.x * 256ll + threadIdx.x);
}
// every (layer, block) weight of a model in one launch: blockIdx.y = job; pointers travel by value
constexpr int PACK_MAX_JOBS = 64;
struct PackJobs {
  const float* w[PACK_MAX_JOBS];
  const float* b[PACK_MAX_JOBS];
  float* wbig[PACK_MAX_JOBS];
  float* bbig[PACK_MAX_JOBS];
};
__global__ void afno_pack_multi_kernel(const PackJobs jobs, int nb, int bs) {
  const int j = blockIdx.y;
  afno_pack_one(jobs.w[j], jobs.b[j], jobs.wbig[j], jobs.bbig[j], nb, bs, blockIdx.x * 256ll + threadIdx.x);
}
__device__ __forceinline__ void afno_pack_one(const float* __restrict__ w, const float* __restrict__ b,
                                              float* __restrict__ wbig, float* __restrict__ bbig, int nb, int bs,
                                              long long idx) {
  const long long nW = (long long)nb * 4 * bs * bs;
  if (idx < nW) {
    const int col = (int)(idx % (2 * bs));
    const int row = (int)((idx / (2 * bs)) % (2 * bs));
    const int k = (int)(idx / ((long long)4 * bs * bs));
    const int i = row % bs, o = col % bs;
    const bool rim = row >= bs, cim = col >= bs;
    const long long wi = ((long long)k * bs + i) * bs + o;
    const long long plane = (long long)nb * bs * bs;
    // [[Wr, Wi], [-Wi, Wr]]
    float v;
    if (!rim && !cim) v = w[wi];
    else if (!rim && cim) v = w[plane + wi];
    else if (rim && !cim) v = -w[plane + wi];
    else v = w[wi];
    wbig[idx] = v;
  }
  if (idx < (long long)nb * 2 * bs) {
    const int ci = (int)(idx % bs), part = (int)((idx / bs) & 1), k = (int)(idx / (2 * bs));
    bbig[idx] = b[((long long)part * nb + k) * bs + ci];
  }
}

__global__ void afno_unpack_grad_kernel(const float* __restrict__ dwbig, const float* __restrict__ dbbig,
                                        float* __restrict__ dw, float* __restrict__ db, int nb, int bs) {
  const long long nw = (long long)nb * bs * bs;
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx < nw) {
    const int o = (int)(idx % bs), i = (int)((idx / bs) % bs), k = (int)(idx / ((long long)bs * bs));
    const float* Wb = dwbig + (long long)k * 4 * bs * bs;
    const int ld = 2 * bs;
    dw[idx] = Wb[(long long)i * ld + o] + Wb[(long long)(bs + i) * ld + bs + o];           // d Wr
    dw[nw + idx] = Wb[(long long)i * ld + bs + o] - Wb[(long long)(bs + i) * ld + o];      // d Wi
  }
  if (idx < (long long)nb * 2 * bs) {
    const int ci = (int)(idx % bs), part = (int)((idx / bs) & 1), k = (int)(idx / (2 * bs));
    db[((long long)part * nb + k) * bs + ci] = dbbig[idx];
  }
}

}  // namespace dpot

using namespace dpot;

static int check_dft_args(const char* who, int B, int h, int w, int E, int nb, int mx, int my) {
  DPOT_REQUIRE(B > 0 && h > 0 && w > 0 && E > 0 && nb > 0, "%s: bad shape", who);
  DPOT_REQUIRE(E % nb == 0, "%s: E=%d not divisible by nb=%d", who, E, nb);
  DPOT_REQUIRE(mx >= 1 && mx <= h && my >= 1 && my <= w / 2 + 1, "%s: kept modes (%d,%d) outside (%d,%d)", who, mx,
               my, h, w / 2 + 1);
  DPOT_REQUIRE(B <= 65535, "%s: batch too large for grid.y", who);
  return DPOT_OK;
}

extern "C" int dpot_rfft2(const float* x, float* spec, int B, int h, int w, int E, int nb, int mx, int my,
                          int col_weights, dpot_stream_t stream) {
  int rc = check_dft_args("rfft2", B, h, w, E, nb, mx, my);
  if (rc) return rc;
  DPOT_REQUIRE(x && spec, "rfft2: null pointer");
  {
    int frc = 0;
    if (try_rfft2_fast(x, spec, B, h, w, E, nb, mx, my, col_weights, (float)(1.0 / sqrt((double)h * (double)w)),
                       as_stream(stream), &frc))
      return frc;
  }
  const int tw = ((2 * w + 2 * h + 3) & ~3);
  const int CC = pick_cc(E, (long long)h * w + (long long)h * my * 2, tw);
  if (CC == 0) {
    set_error("rfft2: latent grid %dx%d does not fit LDS", h, w);
    return DPOT_EUNSUP;
  }
  const size_t lds = sizeof(float) * (tw + (size_t)CC * ((size_t)h * w + (size_t)h * my * 2));
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rfft2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)lds);
  const float scale = (float)(1.0 / sqrt((double)h * (double)w));
  hipLaunchKernelGGL(rfft2_kernel, dim3(E / CC, B), dim3(256), lds, as_stream(stream), x, spec, h, w, E, nb, mx, my,
                     CC, col_weights, scale);
  return check_launch("rfft2_kernel");
}

// rfft2(GroupNorm(x)) with the statistics given: register-FFT grids only (dpot_rfft2_norm_supported)
extern "C" int dpot_rfft2_norm_supported(int h, int w, int E) {
  if (h != w) return 0;
  if (h == 16 || h == 8) return E % 32 == 0;
  if (h == 64) return E % 8 == 0;
  if (h == 32) return E % 16 == 0;
  return 0;
}

extern "C" int dpot_rfft2_norm(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                               int G, float* spec, int B, int h, int w, int E, int nb, int mx, int my, int col_weights,
                               dpot_stream_t stream) {
  int rc = check_dft_args("rfft2_norm", B, h, w, E, nb, mx, my);
  if (rc) return rc;
  DPOT_REQUIRE(x && spec && mean && rstd && gamma && beta && G > 0 && E % G == 0, "rfft2_norm: bad argument");
  DPOT_REQUIRE(dpot_rfft2_norm_supported(h, w, E), "rfft2_norm: latent grid %dx%d has no register-FFT path", h, w);
  int frc = 0;
  if (try_rfft2_fast(x, spec, B, h, w, E, nb, mx, my, col_weights, (float)(1.0 / sqrt((double)h * (double)w)),
                     as_stream(stream), &frc, DftNorm{mean, rstd, gamma, beta, G}))
    return frc;
  set_error("rfft2_norm: no register-FFT kernel for this shape");
  return DPOT_EUNSUP;
}

extern "C" int dpot_irfft2(const float* spec, const float* res, float* y, int B, int h, int w, int E, int nb, int mx,
                           int my, int col_weights, dpot_stream_t stream) {
  int rc = check_dft_args("irfft2", B, h, w, E, nb, mx, my);
  if (rc) return rc;
  DPOT_REQUIRE(spec && y, "irfft2: null pointer");
  {
    int frc = 0;
    if (try_irfft2_fast(spec, res, y, B, h, w, E, nb, mx, my, col_weights,
                        (float)(1.0 / sqrt((double)h * (double)w)), as_stream(stream), &frc))
      return frc;
  }
  const int tw = ((2 * w + 2 * h + 3) & ~3);
  const int CC = pick_cc(E, (long long)mx * my * 2 + (long long)h * my * 2, tw);
  if (CC == 0) {
    set_error("irfft2: latent grid %dx%d does not fit LDS", h, w);
    return DPOT_EUNSUP;
  }
  const size_t lds = sizeof(float) * (tw + (size_t)CC * ((size_t)mx * my * 2 + (size_t)h * my * 2));
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(irfft2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)lds);
  const float scale = (float)(1.0 / sqrt((double)h * (double)w));
  hipLaunchKernelGGL(irfft2_kernel, dim3(E / CC, B), dim3(256), lds, as_stream(stream), spec, res, y, h, w, E, nb,
                     mx, my, CC, col_weights, scale);
  return check_launch("irfft2_kernel");
}

extern "C" int dpot_afno_pack(const float* w, const float* b, float* wbig, float* bbig, int nb, int bs,
                              dpot_stream_t stream) {
  DPOT_REQUIRE(w && b && wbig && bbig && nb > 0 && bs > 0, "afno_pack: bad argument");
  const long long n = (long long)nb * 4 * bs * bs;
  hipLaunchKernelGGL(afno_pack_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, as_stream(stream), w, b, wbig,
                     bbig, nb, bs);
  return check_launch("afno_pack_kernel");
}

extern "C" int dpot_afno_pack_multi(const float* const* w, const float* const* b, float* const* wbig,
                                    float* const* bbig, int njobs, int nb, int bs, dpot_stream_t stream) {
  DPOT_REQUIRE(w && b && wbig && bbig && njobs > 0 && nb > 0 && bs > 0, "afno_pack_multi: bad argument");
  const long long n = (long long)nb * 4 * bs * bs;
  for (int j0 = 0; j0 < njobs; j0 += PACK_MAX_JOBS) {
    const int nj = njobs - j0 < PACK_MAX_JOBS ? njobs - j0 : PACK_MAX_JOBS;
    PackJobs jobs;
    for (int j = 0; j < nj; ++j) {
      DPOT_REQUIRE(w[j0 + j] && b[j0 + j] && wbig[j0 + j] && bbig[j0 + j], "afno_pack_multi: null pointer in job %d", j0 + j);
      jobs.w[j] = w[j0 + j]; jobs.b[j] = b[j0 + j]; jobs.wbig[j] = wbig[j0 + j]; jobs.bbig[j] = bbig[j0 + j];
    }
    hipLaunchKernelGGL(afno_pack_multi_kernel, dim3((unsigned)cdiv64(n, 256), nj), dim3(256), 0, as_stream(stream), jobs,
                       nb, bs);
    int rc = check_launch("afno_pack_multi_kernel");
    if (rc) return rc;
  }
  return DPOT_OK;
}

extern "C" int dpot_afno_unpack_grad(const float* dwbig, const float* dbbig, float* dw, float* db, int nb, int bs,
                                     dpot_stream_t stream) {
  DPOT_REQUIRE(dwbig && dbbig && dw && db && nb > 0 && bs > 0, "afno_unpack_grad: bad argument");
  long long n = (long long)nb * bs * bs;
  if (n < (long long)nb * 2 * bs) n = (long long)nb * 2 * bs;
  hipLaunchKernelGGL(afno_unpack_grad_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, as_stream(stream), dwbig,
                     dbbig, dw, db, nb, bs);
  return check_launch("afno_unpack_grad_kernel");
}

// irfft2(spec) + GroupNorm(res) with the statistics given (the "+ x_orig" of models/dpot.py:106 where x_orig = norm1(x))
extern "C" int dpot_irfft2_norm(const float* spec, const float* res, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, int G, float* y, int B, int h, int w, int E, int nb,
                                int mx, int my, int col_weights, dpot_stream_t stream) {
  int rc = check_dft_args("irfft2_norm", B, h, w, E, nb, mx, my);
  if (rc) return rc;
  DPOT_REQUIRE(spec && res && y && mean && rstd && gamma && beta && G > 0 && E % G == 0, "irfft2_norm: bad argument");
  DPOT_REQUIRE(dpot_rfft2_norm_supported(h, w, E), "irfft2_norm: latent grid %dx%d has no register-FFT path", h, w);
  int frc = 0;
  if (try_irfft2_fast(spec, res, y, B, h, w, E, nb, mx, my, col_weights, (float)(1.0 / sqrt((double)h * (double)w)),
                      as_stream(stream), &frc, DftNorm{mean, rstd, gamma, beta, G}))
    return frc;
  set_error("irfft2_norm: no register-FFT kernel for this shape");
  return DPOT_EUNSUP;
}
