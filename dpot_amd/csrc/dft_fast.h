// dft_fast.h - register-blocked rfft2 / irfft2 for the power-of-two latent grids (and, round 5, the 3 * 2^k ones: 24 / 48 / 96
// = 192^2 / 384^2 / 768^2 fields at patch 8, one radix-3 stage in front of three 2^k-point transforms): 16x16 (128^2 / patch 8) and 32x32
// (256^2 / patch 8) - what the DPOT configs use - plus 8x8, 64x64 and 128x128 (64^2, 512^2 and 1024^2 fields at patch 8:
// the other resolutions utils/griddataset.py:35 lists; 128^2 / 256^2 at patch 16 / 4).  Same contract as the generic
// kernels in dft.hip (which keep every other size, odd ones included):
//   * both passes keep one line of a channel (16 / 32 points) in VGPRs and transform it with a fully unrolled
//     radix-2 FFT whose twiddles are compile-time constants (round 1 evaluated the direct O(N^2) sums there, which left
//     the kernels VALU-bound at 2.7 / 3.5 TB/s; the real-input / one-sided zeros and the unused outputs fold away)
//   * the half-complex intermediate crosses between the passes through LDS, laid out so that both the writes and the
//     reads are conflict free (channel index on the lanes)
//   * HBM traffic stays at the algorithmic minimum: one read of the field, one write of the kept modes.
#pragma once
#include <type_traits>

#include "common.h"

namespace dpot {

template <int N> struct Twid;
template <> struct Twid<16> {
  static __host__ __device__ constexpr float c(int i) {
    constexpr float t[16] = {1.0f, 0.9238795042037964f, 0.7071067690849304f, 0.3826834261417389f, 0.0f, -0.3826834261417389f, -0.7071067690849304f, -0.9238795042037964f, -1.0f, -0.9238795042037964f, -0.7071067690849304f, -0.3826834261417389f, 0.0f, 0.3826834261417389f, 0.7071067690849304f, 0.9238795042037964f};
    return t[i];
  }
  static __host__ __device__ constexpr float s(int i) {
    constexpr float t[16] = {0.0f, 0.3826834261417389f, 0.7071067690849304f, 0.9238795042037964f, 1.0f, 0.9238795042037964f, 0.7071067690849304f, 0.3826834261417389f, 0.0f, -0.3826834261417389f, -0.7071067690849304f, -0.9238795042037964f, -1.0f, -0.9238795042037964f, -0.7071067690849304f, -0.3826834261417389f};
    return t[i];
  }
};

template <> struct Twid<32> {
  static __host__ __device__ constexpr float c(int i) {
    constexpr float t[32] = {1.0f, 0.9807852506637573f, 0.9238795042037964f, 0.8314695954322815f, 0.7071067690849304f, 0.5555702447891235f, 0.3826834261417389f, 0.19509032368659973f, 0.0f, -0.19509032368659973f, -0.3826834261417389f, -0.5555702447891235f, -0.7071067690849304f, -0.8314695954322815f, -0.9238795042037964f, -0.9807852506637573f, -1.0f, -0.9807852506637573f, -0.9238795042037964f, -0.8314695954322815f, -0.7071067690849304f, -0.5555702447891235f, -0.3826834261417389f, -0.19509032368659973f, 0.0f, 0.19509032368659973f, 0.3826834261417389f, 0.5555702447891235f, 0.7071067690849304f, 0.8314695954322815f, 0.9238795042037964f, 0.9807852506637573f};
    return t[i];
  }
  static __host__ __device__ constexpr float s(int i) {
    constexpr float t[32] = {0.0f, 0.19509032368659973f, 0.3826834261417389f, 0.5555702447891235f, 0.7071067690849304f, 0.8314695954322815f, 0.9238795042037964f, 0.9807852506637573f, 1.0f, 0.9807852506637573f, 0.9238795042037964f, 0.8314695954322815f, 0.7071067690849304f, 0.5555702447891235f, 0.3826834261417389f, 0.19509032368659973f, 0.0f, -0.19509032368659973f, -0.3826834261417389f, -0.5555702447891235f, -0.7071067690849304f, -0.8314695954322815f, -0.9238795042037964f, -0.9807852506637573f, -1.0f, -0.9807852506637573f, -0.9238795042037964f, -0.8314695954322815f, -0.7071067690849304f, -0.5555702447891235f, -0.3826834261417389f, -0.19509032368659973f};
    return t[i];
  }
};


template <> struct Twid<8> {   // (every 8-point twiddle is one of the special-cased angles; the table only has to exist)
  static __host__ __device__ constexpr float c(int i) {
    constexpr float t[8] = {1.0f, 0.7071067690849304f, 0.0f, -0.7071067690849304f, -1.0f, -0.7071067690849304f, 0.0f, 0.7071067690849304f};
    return t[i];
  }
  static __host__ __device__ constexpr float s(int i) {
    constexpr float t[8] = {0.0f, 0.7071067690849304f, 1.0f, 0.7071067690849304f, 0.0f, -0.7071067690849304f, -1.0f, -0.7071067690849304f};
    return t[i];
  }
};

template <> struct Twid<64> {
  static __host__ __device__ constexpr float c(int i) {
    constexpr float t[64] = {1.0f, 0.9951847195625305f, 0.9807852506637573f, 0.9569403529167175f, 0.9238795042037964f, 0.8819212913513184f, 0.8314695954322815f, 0.7730104327201843f, 0.7071067690849304f, 0.6343932747840881f, 0.5555702447891235f, 0.4713967442512512f, 0.3826834261417389f, 0.290284663438797f, 0.19509032368659973f, 0.0980171412229538f, 0.0f, -0.0980171412229538f, -0.19509032368659973f, -0.290284663438797f, -0.3826834261417389f, -0.4713967442512512f, -0.5555702447891235f, -0.6343932747840881f, -0.7071067690849304f, -0.7730104327201843f, -0.8314695954322815f, -0.8819212913513184f, -0.9238795042037964f, -0.9569403529167175f, -0.9807852506637573f, -0.9951847195625305f, -1.0f, -0.9951847195625305f, -0.9807852506637573f, -0.9569403529167175f, -0.9238795042037964f, -0.8819212913513184f, -0.8314695954322815f, -0.7730104327201843f, -0.7071067690849304f, -0.6343932747840881f, -0.5555702447891235f, -0.4713967442512512f, -0.3826834261417389f, -0.290284663438797f, -0.19509032368659973f, -0.0980171412229538f, 0.0f, 0.0980171412229538f, 0.19509032368659973f, 0.290284663438797f, 0.3826834261417389f, 0.4713967442512512f, 0.5555702447891235f, 0.6343932747840881f, 0.7071067690849304f, 0.7730104327201843f, 0.8314695954322815f, 0.8819212913513184f, 0.9238795042037964f, 0.9569403529167175f, 0.9807852506637573f, 0.9951847195625305f};
    return t[i];
  }
  static __host__ __device__ constexpr float s(int i) {
    constexpr float t[64] = {0.0f, 0.0980171412229538f, 0.19509032368659973f, 0.290284663438797f, 0.3826834261417389f, 0.4713967442512512f, 0.5555702447891235f, 0.6343932747840881f, 0.7071067690849304f, 0.7730104327201843f, 0.8314695954322815f, 0.8819212913513184f, 0.9238795042037964f, 0.9569403529167175f, 0.9807852506637573f, 0.9951847195625305f, 1.0f, 0.9951847195625305f, 0.9807852506637573f, 0.9569403529167175f, 0.9238795042037964f, 0.8819212913513184f, 0.8314695954322815f, 0.7730104327201843f, 0.7071067690849304f, 0.6343932747840881f, 0.5555702447891235f, 0.4713967442512512f, 0.3826834261417389f, 0.290284663438797f, 0.19509032368659973f, 0.0980171412229538f, 0.0f, -0.0980171412229538f, -0.19509032368659973f, -0.290284663438797f, -0.3826834261417389f, -0.4713967442512512f, -0.5555702447891235f, -0.6343932747840881f, -0.7071067690849304f, -0.7730104327201843f, -0.8314695954322815f, -0.8819212913513184f, -0.9238795042037964f, -0.9569403529167175f, -0.9807852506637573f, -0.9951847195625305f, -1.0f, -0.9951847195625305f, -0.9807852506637573f, -0.9569403529167175f, -0.9238795042037964f, -0.8819212913513184f, -0.8314695954322815f, -0.7730104327201843f, -0.7071067690849304f, -0.6343932747840881f, -0.5555702447891235f, -0.4713967442512512f, -0.3826834261417389f, -0.290284663438797f, -0.19509032368659973f, -0.0980171412229538f};
    return t[i];
  }
};

template <> struct Twid<128> {
  static __host__ __device__ constexpr float c(int i) {
    constexpr float t[128] = {1.0f, 0.9987954497337341f, 0.9951847195625305f, 0.9891765117645264f, 0.9807852506637573f, 0.9700312614440918f, 0.9569403529167175f, 0.9415440559387207f, 0.9238795042037964f, 0.903989315032959f, 0.8819212913513184f, 0.8577286005020142f, 0.8314695954322815f, 0.803207516670227f, 0.7730104327201843f, 0.7409511208534241f, 0.7071067690849304f, 0.6715589761734009f, 0.6343932747840881f, 0.5956993103027344f, 0.5555702447891235f, 0.5141027569770813f, 0.4713967442512512f, 0.4275550842285156f, 0.3826834261417389f, 0.3368898630142212f, 0.290284663438797f, 0.24298018217086792f, 0.19509032368659973f, 0.1467304676771164f, 0.0980171412229538f, 0.049067676067352295f, 0.0f, -0.049067676067352295f, -0.0980171412229538f, -0.1467304676771164f, -0.19509032368659973f, -0.24298018217086792f, -0.290284663438797f, -0.3368898630142212f, -0.3826834261417389f, -0.4275550842285156f, -0.4713967442512512f, -0.5141027569770813f, -0.5555702447891235f, -0.5956993103027344f, -0.6343932747840881f, -0.6715589761734009f, -0.7071067690849304f, -0.7409511208534241f, -0.7730104327201843f, -0.803207516670227f, -0.8314695954322815f, -0.8577286005020142f, -0.8819212913513184f, -0.903989315032959f, -0.9238795042037964f, -0.9415440559387207f, -0.9569403529167175f, -0.9700312614440918f, -0.9807852506637573f, -0.9891765117645264f, -0.9951847195625305f, -0.9987954497337341f, -1.0f, -0.9987954497337341f, -0.9951847195625305f, -0.9891765117645264f, -0.9807852506637573f, -0.9700312614440918f, -0.9569403529167175f, -0.9415440559387207f, -0.9238795042037964f, -0.903989315032959f, -0.8819212913513184f, -0.8577286005020142f, -0.8314695954322815f, -0.803207516670227f, -0.7730104327201843f, -0.7409511208534241f, -0.7071067690849304f, -0.6715589761734009f, -0.6343932747840881f, -0.5956993103027344f, -0.5555702447891235f, -0.5141027569770813f, -0.4713967442512512f, -0.4275550842285156f, -0.3826834261417389f, -0.3368898630142212f, -0.290284663438797f, -0.24298018217086792f, -0.19509032368659973f, -0.1467304676771164f, -0.0980171412229538f, -0.049067676067352295f, 0.0f, 0.049067676067352295f, 0.0980171412229538f, 0.1467304676771164f, 0.19509032368659973f, 0.24298018217086792f, 0.290284663438797f, 0.3368898630142212f, 0.3826834261417389f, 0.4275550842285156f, 0.4713967442512512f, 0.5141027569770813f, 0.5555702447891235f, 0.5956993103027344f, 0.6343932747840881f, 0.6715589761734009f, 0.7071067690849304f, 0.7409511208534241f, 0.7730104327201843f, 0.803207516670227f, 0.8314695954322815f, 0.8577286005020142f, 0.8819212913513184f, 0.903989315032959f, 0.9238795042037964f, 0.9415440559387207f, 0.9569403529167175f, 0.9700312614440918f, 0.9807852506637573f, 0.9891765117645264f, 0.9951847195625305f, 0.9987954497337341f};
    return t[i];
  }
  static __host__ __device__ constexpr float s(int i) {
    constexpr float t[128] = {0.0f, 0.049067676067352295f, 0.0980171412229538f, 0.1467304676771164f, 0.19509032368659973f, 0.24298018217086792f, 0.290284663438797f, 0.3368898630142212f, 0.3826834261417389f, 0.4275550842285156f, 0.4713967442512512f, 0.5141027569770813f, 0.5555702447891235f, 0.5956993103027344f, 0.6343932747840881f, 0.6715589761734009f, 0.7071067690849304f, 0.7409511208534241f, 0.7730104327201843f, 0.803207516670227f, 0.8314695954322815f, 0.8577286005020142f, 0.8819212913513184f, 0.903989315032959f, 0.9238795042037964f, 0.9415440559387207f, 0.9569403529167175f, 0.9700312614440918f, 0.9807852506637573f, 0.9891765117645264f, 0.9951847195625305f, 0.9987954497337341f, 1.0f, 0.9987954497337341f, 0.9951847195625305f, 0.9891765117645264f, 0.9807852506637573f, 0.9700312614440918f, 0.9569403529167175f, 0.9415440559387207f, 0.9238795042037964f, 0.903989315032959f, 0.8819212913513184f, 0.8577286005020142f, 0.8314695954322815f, 0.803207516670227f, 0.7730104327201843f, 0.7409511208534241f, 0.7071067690849304f, 0.6715589761734009f, 0.6343932747840881f, 0.5956993103027344f, 0.5555702447891235f, 0.5141027569770813f, 0.4713967442512512f, 0.4275550842285156f, 0.3826834261417389f, 0.3368898630142212f, 0.290284663438797f, 0.24298018217086792f, 0.19509032368659973f, 0.1467304676771164f, 0.0980171412229538f, 0.049067676067352295f, 0.0f, -0.049067676067352295f, -0.0980171412229538f, -0.1467304676771164f, -0.19509032368659973f, -0.24298018217086792f, -0.290284663438797f, -0.3368898630142212f, -0.3826834261417389f, -0.4275550842285156f, -0.4713967442512512f, -0.5141027569770813f, -0.5555702447891235f, -0.5956993103027344f, -0.6343932747840881f, -0.6715589761734009f, -0.7071067690849304f, -0.7409511208534241f, -0.7730104327201843f, -0.803207516670227f, -0.8314695954322815f, -0.8577286005020142f, -0.8819212913513184f, -0.903989315032959f, -0.9238795042037964f, -0.9415440559387207f, -0.9569403529167175f, -0.9700312614440918f, -0.9807852506637573f, -0.9891765117645264f, -0.9951847195625305f, -0.9987954497337341f, -1.0f, -0.9987954497337341f, -0.9951847195625305f, -0.9891765117645264f, -0.9807852506637573f, -0.9700312614440918f, -0.9569403529167175f, -0.9415440559387207f, -0.9238795042037964f, -0.903989315032959f, -0.8819212913513184f, -0.8577286005020142f, -0.8314695954322815f, -0.803207516670227f, -0.7730104327201843f, -0.7409511208534241f, -0.7071067690849304f, -0.6715589761734009f, -0.6343932747840881f, -0.5956993103027344f, -0.5555702447891235f, -0.5141027569770813f, -0.4713967442512512f, -0.4275550842285156f, -0.3826834261417389f, -0.3368898630142212f, -0.290284663438797f, -0.24298018217086792f, -0.19509032368659973f, -0.1467304676771164f, -0.0980171412229538f, -0.049067676067352295f};
    return t[i];
  }
};

// 3 * 2^k lines (round 5: 24 / 48 / 96-point latent grids = 192^2 / 384^2 / 768^2 fields at patch 8)
template <> struct Twid<24> {
  static __host__ __device__ constexpr float c(int i) {
    constexpr float t[24] = {1.0f, 0.9659258127212524f, 0.8660253882408142f, 0.7071067690849304f, 0.5f, 0.258819043636322f, 0.0f, -0.258819043636322f, -0.5f, -0.7071067690849304f, -0.8660253882408142f, -0.9659258127212524f, -1.0f, -0.9659258127212524f, -0.8660253882408142f, -0.7071067690849304f, -0.5f, -0.258819043636322f, 0.0f, 0.258819043636322f, 0.5f, 0.7071067690849304f, 0.8660253882408142f, 0.9659258127212524f};
    return t[i];
  }
  static __host__ __device__ constexpr float s(int i) {
    constexpr float t[24] = {0.0f, 0.258819043636322f, 0.5f, 0.7071067690849304f, 0.8660253882408142f, 0.9659258127212524f, 1.0f, 0.9659258127212524f, 0.8660253882408142f, 0.7071067690849304f, 0.5f, 0.258819043636322f, 0.0f, -0.258819043636322f, -0.5f, -0.7071067690849304f, -0.8660253882408142f, -0.9659258127212524f, -1.0f, -0.9659258127212524f, -0.8660253882408142f, -0.7071067690849304f, -0.5f, -0.258819043636322f};
    return t[i];
  }
};

template <> struct Twid<48> {
  static __host__ __device__ constexpr float c(int i) {
    constexpr float t[48] = {1.0f, 0.9914448857307434f, 0.9659258127212524f, 0.9238795042037964f, 0.8660253882408142f, 0.7933533191680908f, 0.7071067690849304f, 0.6087614297866821f, 0.5f, 0.3826834261417389f, 0.258819043636322f, 0.13052618503570557f, 0.0f, -0.13052618503570557f, -0.258819043636322f, -0.3826834261417389f, -0.5f, -0.6087614297866821f, -0.7071067690849304f, -0.7933533191680908f, -0.8660253882408142f, -0.9238795042037964f, -0.9659258127212524f, -0.9914448857307434f, -1.0f, -0.9914448857307434f, -0.9659258127212524f, -0.9238795042037964f, -0.8660253882408142f, -0.7933533191680908f, -0.7071067690849304f, -0.6087614297866821f, -0.5f, -0.3826834261417389f, -0.258819043636322f, -0.13052618503570557f, 0.0f, 0.13052618503570557f, 0.258819043636322f, 0.3826834261417389f, 0.5f, 0.6087614297866821f, 0.7071067690849304f, 0.7933533191680908f, 0.8660253882408142f, 0.9238795042037964f, 0.9659258127212524f, 0.9914448857307434f};
    return t[i];
  }
  static __host__ __device__ constexpr float s(int i) {
    constexpr float t[48] = {0.0f, 0.13052618503570557f, 0.258819043636322f, 0.3826834261417389f, 0.5f, 0.6087614297866821f, 0.7071067690849304f, 0.7933533191680908f, 0.8660253882408142f, 0.9238795042037964f, 0.9659258127212524f, 0.9914448857307434f, 1.0f, 0.9914448857307434f, 0.9659258127212524f, 0.9238795042037964f, 0.8660253882408142f, 0.7933533191680908f, 0.7071067690849304f, 0.6087614297866821f, 0.5f, 0.3826834261417389f, 0.258819043636322f, 0.13052618503570557f, 0.0f, -0.13052618503570557f, -0.258819043636322f, -0.3826834261417389f, -0.5f, -0.6087614297866821f, -0.7071067690849304f, -0.7933533191680908f, -0.8660253882408142f, -0.9238795042037964f, -0.9659258127212524f, -0.9914448857307434f, -1.0f, -0.9914448857307434f, -0.9659258127212524f, -0.9238795042037964f, -0.8660253882408142f, -0.7933533191680908f, -0.7071067690849304f, -0.6087614297866821f, -0.5f, -0.3826834261417389f, -0.258819043636322f, -0.13052618503570557f};
    return t[i];
  }
};

template <> struct Twid<96> {
  static __host__ __device__ constexpr float c(int i) {
    constexpr float t[96] = {1.0f, 0.9978589415550232f, 0.9914448857307434f, 0.9807852506637573f, 0.9659258127212524f, 0.9469301104545593f, 0.9238795042037964f, 0.8968727588653564f, 0.8660253882408142f, 0.8314695954322815f, 0.7933533191680908f, 0.751839816570282f, 0.7071067690849304f, 0.659345805644989f, 0.6087614297866821f, 0.5555702447891235f, 0.5f, 0.44228869676589966f, 0.3826834261417389f, 0.3214394748210907f, 0.258819043636322f, 0.19509032368659973f, 0.13052618503570557f, 0.06540312618017197f, 0.0f, -0.06540312618017197f, -0.13052618503570557f, -0.19509032368659973f, -0.258819043636322f, -0.3214394748210907f, -0.3826834261417389f, -0.44228869676589966f, -0.5f, -0.5555702447891235f, -0.6087614297866821f, -0.659345805644989f, -0.7071067690849304f, -0.751839816570282f, -0.7933533191680908f, -0.8314695954322815f, -0.8660253882408142f, -0.8968727588653564f, -0.9238795042037964f, -0.9469301104545593f, -0.9659258127212524f, -0.9807852506637573f, -0.9914448857307434f, -0.9978589415550232f, -1.0f, -0.9978589415550232f, -0.9914448857307434f, -0.9807852506637573f, -0.9659258127212524f, -0.9469301104545593f, -0.9238795042037964f, -0.8968727588653564f, -0.8660253882408142f, -0.8314695954322815f, -0.7933533191680908f, -0.751839816570282f, -0.7071067690849304f, -0.659345805644989f, -0.6087614297866821f, -0.5555702447891235f, -0.5f, -0.44228869676589966f, -0.3826834261417389f, -0.3214394748210907f, -0.258819043636322f, -0.19509032368659973f, -0.13052618503570557f, -0.06540312618017197f, 0.0f, 0.06540312618017197f, 0.13052618503570557f, 0.19509032368659973f, 0.258819043636322f, 0.3214394748210907f, 0.3826834261417389f, 0.44228869676589966f, 0.5f, 0.5555702447891235f, 0.6087614297866821f, 0.659345805644989f, 0.7071067690849304f, 0.751839816570282f, 0.7933533191680908f, 0.8314695954322815f, 0.8660253882408142f, 0.8968727588653564f, 0.9238795042037964f, 0.9469301104545593f, 0.9659258127212524f, 0.9807852506637573f, 0.9914448857307434f, 0.9978589415550232f};
    return t[i];
  }
  static __host__ __device__ constexpr float s(int i) {
    constexpr float t[96] = {0.0f, 0.06540312618017197f, 0.13052618503570557f, 0.19509032368659973f, 0.258819043636322f, 0.3214394748210907f, 0.3826834261417389f, 0.44228869676589966f, 0.5f, 0.5555702447891235f, 0.6087614297866821f, 0.659345805644989f, 0.7071067690849304f, 0.751839816570282f, 0.7933533191680908f, 0.8314695954322815f, 0.8660253882408142f, 0.8968727588653564f, 0.9238795042037964f, 0.9469301104545593f, 0.9659258127212524f, 0.9807852506637573f, 0.9914448857307434f, 0.9978589415550232f, 1.0f, 0.9978589415550232f, 0.9914448857307434f, 0.9807852506637573f, 0.9659258127212524f, 0.9469301104545593f, 0.9238795042037964f, 0.8968727588653564f, 0.8660253882408142f, 0.8314695954322815f, 0.7933533191680908f, 0.751839816570282f, 0.7071067690849304f, 0.659345805644989f, 0.6087614297866821f, 0.5555702447891235f, 0.5f, 0.44228869676589966f, 0.3826834261417389f, 0.3214394748210907f, 0.258819043636322f, 0.19509032368659973f, 0.13052618503570557f, 0.06540312618017197f, 0.0f, -0.06540312618017197f, -0.13052618503570557f, -0.19509032368659973f, -0.258819043636322f, -0.3214394748210907f, -0.3826834261417389f, -0.44228869676589966f, -0.5f, -0.5555702447891235f, -0.6087614297866821f, -0.659345805644989f, -0.7071067690849304f, -0.751839816570282f, -0.7933533191680908f, -0.8314695954322815f, -0.8660253882408142f, -0.8968727588653564f, -0.9238795042037964f, -0.9469301104545593f, -0.9659258127212524f, -0.9807852506637573f, -0.9914448857307434f, -0.9978589415550232f, -1.0f, -0.9978589415550232f, -0.9914448857307434f, -0.9807852506637573f, -0.9659258127212524f, -0.9469301104545593f, -0.9238795042037964f, -0.8968727588653564f, -0.8660253882408142f, -0.8314695954322815f, -0.7933533191680908f, -0.751839816570282f, -0.7071067690849304f, -0.659345805644989f, -0.6087614297866821f, -0.5555702447891235f, -0.5f, -0.44228869676589966f, -0.3826834261417389f, -0.3214394748210907f, -0.258819043636322f, -0.19509032368659973f, -0.13052618503570557f, -0.06540312618017197f};
    return t[i];
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// N-point complex FFT in registers (N = 16 / 32), fully unrolled decimation-in-frequency with compile-time twiddles:
// (N/2) log2 N butterflies instead of the N^2 complex MACs of the direct sum - the direct form made these kernels
// VALU-bound (pass 2 of rfft2 at 16x16: 1024 FMAs per (ky, channel) item, ~7 us of a 13 us kernel).  SGN = -1: forward
// (e^{-i..}), +1: inverse.  Output X[k] is left at array index brev<N>(k).  Twiddles equal to 1, +-i, (+-1 +- i)/sqrt 2
// are special-cased; inputs that are compile-time zeros (real input, truncated spectra) and unused outputs fold away.
// ---------------------------------------------------------------------------------------------------------------------
template <int N>
__host__ __device__ constexpr int brev(int k) {
  int r = 0;
  for (int b = 1; b < N; b <<= 1) {
    r = (r << 1) | (k & 1);
    k >>= 1;
  }
  return r;
}

template <int I>
using fft_ic = std::integral_constant<int, I>;
template <int B, int E, class F>
__device__ __forceinline__ void fft_sfor(F&& f) {
  if constexpr (B < E) {
    f(fft_ic<B>{});
    fft_sfor<B + 1, E>(f);
  }
}

template <int N, int SGN, int LEN>
__device__ __forceinline__ void fft_stage(float (&re)[N], float (&im)[N]) {
  if constexpr (LEN >= 2) {
    constexpr int H = LEN / 2;
    fft_sfor<0, N / 2>([&](auto Q) __attribute__((always_inline)) {
      constexpr int q = decltype(Q)::value;
      constexpr int i = (q / H) * LEN + (q % H), j = i + H;
      constexpr int tw = (q % H) * (N / LEN);          // twiddle angle 2 pi tw / N, tw < N/2
      const float tr = re[i] - re[j], ti = im[i] - im[j];
      re[i] += re[j];
      im[i] += im[j];
      constexpr float S = (float)SGN;
      if constexpr (tw == 0) {
        re[j] = tr;
        im[j] = ti;
      } else if constexpr (4 * tw == N) {              // w = S i
        re[j] = -S * ti;
        im[j] = S * tr;
      } else if constexpr (8 * tw == N) {              // w = (1 + S i) / sqrt 2
        re[j] = (tr - S * ti) * 0.70710678118654752f;
        im[j] = (S * tr + ti) * 0.70710678118654752f;
      } else if constexpr (8 * tw == 3 * N) {          // w = (-1 + S i) / sqrt 2
        re[j] = (-tr - S * ti) * 0.70710678118654752f;
        im[j] = (S * tr - ti) * 0.70710678118654752f;
      } else {
        constexpr float c = Twid<N>::c(tw), sn = S * Twid<N>::s(tw);
        re[j] = fmaf(tr, c, -ti * sn);
        im[j] = fmaf(tr, sn, ti * c);
      }
    });
    fft_stage<N, SGN, LEN / 2>(re, im);
  }
}
// position of output X[k] in the array after fft_regs<N>: bit reversal for N = 2^k; for N = 3 * 2^k (one radix-3
// decimation-in-frequency stage in front of three 2^k-point transforms) X[3 j + r] sits at r * (N / 3) + brev(j)
template <int N>
__host__ __device__ constexpr int fft_pos(int k) {
  if constexpr (N % 3 == 0) return (k % 3) * (N / 3) + brev<N / 3>(k / 3);
  else return brev<N>(k);
}
template <int N, int SGN>
__device__ __forceinline__ void fft_regs(float (&re)[N], float (&im)[N]) {
  if constexpr (N % 3 == 0) {
    // radix-3 stage: y_r[n] = (a + w3^r b + w3^(2r) c) W^(r n), a / b / c = x[n], x[n + M], x[n + 2M], w3 = e^{SGN 2 pi i / 3}
    constexpr int M = N / 3;
    constexpr float S = (float)SGN, h = 0.86602540378443865f;      // sin(2 pi / 3)
    fft_sfor<0, M>([&](auto Q) __attribute__((always_inline)) {
      constexpr int n = decltype(Q)::value;
      const float ar = re[n], ai = im[n], br = re[n + M], bi = im[n + M], cr = re[n + 2 * M], ci = im[n + 2 * M];
      const float sr = br + cr, si = bi + ci, dr = br - cr, di = bi - ci;
      re[n] = ar + sr;
      im[n] = ai + si;
      // a - (b + c) / 2  +-  S i h (b - c)
      const float mr = fmaf(-0.5f, sr, ar), mi = fmaf(-0.5f, si, ai);
      const float y1r = mr - S * h * di, y1i = mi + S * h * dr;
      const float y2r = mr + S * h * di, y2i = mi - S * h * dr;
      if constexpr (n == 0) {
        re[n + M] = y1r;
        im[n + M] = y1i;
        re[n + 2 * M] = y2r;
        im[n + 2 * M] = y2i;
      } else {
        constexpr float c1 = Twid<N>::c(n), s1 = S * Twid<N>::s(n);
        constexpr float c2 = Twid<N>::c((2 * n) % N), s2 = S * Twid<N>::s((2 * n) % N);
        re[n + M] = fmaf(y1r, c1, -y1i * s1);
        im[n + M] = fmaf(y1r, s1, y1i * c1);
        re[n + 2 * M] = fmaf(y2r, c2, -y2i * s2);
        im[n + 2 * M] = fmaf(y2r, s2, y2i * c2);
      }
    });
    fft_stage<N, SGN, M>(re, im);
  } else {
    fft_stage<N, SGN, N>(re, im);
  }
}

__device__ __forceinline__ float colw_f(int colw, int ky, int w) {
  if (!colw || ky == 0 || ((w & 1) == 0 && ky == (w >> 1))) return 1.f;
  return 2.f;
}

// Channel chunk of workgroup x: consecutive workgroups go to different XCDs (8 L2 slices); with CC = 16 channels a
// workgroup touches 64-byte halves of the 128-byte lines of a token, so the chunk sharing a line must sit on the SAME
// XCD or every L2 fetches the line for half its bytes.  XCD k gets the contiguous chunks [k*n/8, (k+1)*n/8).
__device__ __forceinline__ int xcd_chunk(int x, int n) {
  return (n & 7) == 0 ? (x & 7) * (n >> 3) + (x >> 3) : x;
}

// x[B,H*W,E] -> spec[B,mx,my,nb,2,bs]
// optional GroupNorm on the way in (dpot_rfft2_norm): the transform of GN(x) = (x - mean) * rstd * gamma + beta with the
// statistics of (sample, group = channel / (E / G)) - the normalised tensor is never written (same expression as the
// GroupNorm apply kernels of norm.hip: the spectrum is bit-identical to the two-launch form)
struct DftNorm {
  const float* mean;     // [B, G]; NULL: plain transform
  const float* rstd;
  const float* gamma;    // [E]
  const float* beta;
  int G;
};
// (Round 6 measured these kernels with 384 / 512 / 1024 threads per workgroup on the 32 x 32 grid of DPOT-L, on the theory that 4 waves
// per CU - one 139 KiB workgroup - cannot keep enough loads in flight: rfft2 unchanged at 512 threads, 2.6x slower at 1024 (128
// registers: 185 spills); irfft2 15-50 % slower with more threads or wider channel chunks.  profiles/r06_dft32_variants.txt.  They run
// at 3.6 TB/s because the phases of the one resident workgroup - loads, row transforms, LDS, column transforms, stores - do not overlap.)
template <int H, int W, int CC>
__global__ __launch_bounds__(256) void rfft2_fast_kernel(const float* __restrict__ x, float* __restrict__ spec, int E,
                                                         int nb, int mx, int my, int colw, float scale,
                                                         const DftNorm nrm) {
  constexpr int WF = W / 2 + 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Z = sm;                                            // [WF][H][2][CC]
  const int b = blockIdx.y, c0 = xcd_chunk(blockIdx.x, gridDim.x) * CC;
  const int tid = threadIdx.x;
  const int bs = E / nb;
  const float* xb = x + (long long)b * H * W * E + c0;

  // pass 1: rows (real -> half complex)
  for (int it = tid; it < H * CC; it += 256) {
    const int c = it % CC, xr = it / CC;
    float v[W];
#pragma unroll
    for (int y = 0; y < W; ++y) v[y] = xb[(long long)(xr * W + y) * E + c];
    if (nrm.mean) {
      const int sg = b * nrm.G + (c0 + c) / (E / nrm.G);
      const float mu = nrm.mean[sg], ga = nrm.gamma[c0 + c] * nrm.rstd[sg], be = nrm.beta[c0 + c];
#pragma unroll
      for (int y = 0; y < W; ++y) v[y] = fmaf(v[y] - mu, ga, be);
    }
    float vi[W];
#pragma unroll
    for (int y = 0; y < W; ++y) vi[y] = 0.f;               // real input: the zero half folds away
    fft_regs<W, -1>(v, vi);
    fft_sfor<0, WF>([&](auto KY) __attribute__((always_inline)) {
      constexpr int ky = decltype(KY)::value;
      Z[((ky * H + xr) * 2 + 0) * CC + c] = v[fft_pos<W>(ky)];
      Z[((ky * H + xr) * 2 + 1) * CC + c] = vi[fft_pos<W>(ky)];
    });
  }
  __syncthreads();

  // pass 2: columns (complex -> complex), kept modes only
  for (int it = tid; it < my * CC; it += 256) {
    const int c = it % CC, ky = it / CC;
    float zr[H], zi[H];
#pragma unroll
    for (int xr = 0; xr < H; ++xr) {
      zr[xr] = Z[((ky * H + xr) * 2 + 0) * CC + c];
      zi[xr] = Z[((ky * H + xr) * 2 + 1) * CC + c];
    }
    const int chn = c0 + c;
    const int blk = chn / bs, ci = chn % bs;
    const float wgt = scale * colw_f(colw, ky, W);
    float* out = spec + (((long long)b * mx * my + ky) * nb + blk) * 2 * bs + ci;
    const long long kxstride = (long long)my * nb * 2 * bs;
    fft_regs<H, -1>(zr, zi);
    fft_sfor<0, H>([&](auto KX) __attribute__((always_inline)) {
      constexpr int kx = decltype(KX)::value;
      if (kx < mx) {
        out[kx * kxstride] = zr[fft_pos<H>(kx)] * wgt;
        out[kx * kxstride + bs] = zi[fft_pos<H>(kx)] * wgt;
      }
    });
  }
}

// spec[B,mx,my,nb,2,bs] (+ res[B,H*W,E]) -> y[B,H*W,E]
template <int H, int W, int CC>
__global__ __launch_bounds__(256) void irfft2_fast_kernel(const float* __restrict__ spec, const float* __restrict__ res,
                                                          float* __restrict__ y, int E, int nb, int mx, int my,
                                                          int colw, float scale, const DftNorm nrm) {
  constexpr int WF = W / 2 + 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* U = sm;                                            // [H][WF][2][CC]
  const int b = blockIdx.y, c0 = xcd_chunk(blockIdx.x, gridDim.x) * CC;
  const int tid = threadIdx.x;
  const int bs = E / nb;

  // pass A: columns, U[x,ky] = sum_kx S[kx,ky] e^{+2 pi i kx x / H}, times the column weight
  for (int it = tid; it < my * CC; it += 256) {
    const int c = it % CC, ky = it / CC;
    const int chn = c0 + c;
    const int blk = chn / bs, ci = chn % bs;
    const float* in = spec + (((long long)b * mx * my + ky) * nb + blk) * 2 * bs + ci;
    const long long kxstride = (long long)my * nb * 2 * bs;
    float sr[H], si[H];
#pragma unroll
    for (int kx = 0; kx < H; ++kx) {
      const int kc = kx < mx ? kx : mx - 1;                 // clamped address, selected value: no divergent loads
      const float a = in[kc * kxstride], bq = in[kc * kxstride + bs];
      sr[kx] = kx < mx ? a : 0.f;
      si[kx] = kx < mx ? bq : 0.f;
    }
    const float wgt = colw_f(colw, ky, W);
    fft_regs<H, 1>(sr, si);
    fft_sfor<0, H>([&](auto XR) __attribute__((always_inline)) {
      constexpr int xr = decltype(XR)::value;
      U[((xr * WF + ky) * 2 + 0) * CC + c] = sr[fft_pos<H>(xr)] * wgt;
      U[((xr * WF + ky) * 2 + 1) * CC + c] = si[fft_pos<H>(xr)] * wgt;
    });
  }
  __syncthreads();

  // pass B: rows (half complex -> real), + residual
  const long long base = (long long)b * H * W * E + c0;
  for (int it = tid; it < H * CC; it += 256) {
    const int c = it % CC, xr = it / CC;
    float ur[W], ui[W];                                    // one-sided spectrum (weights applied), zero above W/2
#pragma unroll
    for (int ky = 0; ky < W; ++ky) {
      if (ky < WF) {
        const int kc = ky < my ? ky : my - 1;
        const float a = U[((xr * WF + kc) * 2 + 0) * CC + c], bq = U[((xr * WF + kc) * 2 + 1) * CC + c];
        ur[ky] = ky < my ? a : 0.f;
        ui[ky] = ky < my ? bq : 0.f;
      } else {
        ur[ky] = 0.f;
        ui[ky] = 0.f;
      }
    }
    fft_regs<W, 1>(ur, ui);                                // y[yy] = Re(sum_ky V[ky] e^{+2 pi i ky yy / W})
    float q[W];
    if (res) {
#pragma unroll
      for (int yy = 0; yy < W; ++yy) q[yy] = res[base + (long long)(xr * W + yy) * E + c];
      if (nrm.mean) {                                      // the residual is GroupNorm(res) (dpot_irfft2_norm)
        const int sg = b * nrm.G + (c0 + c) / (E / nrm.G);
        const float mu = nrm.mean[sg], ga = nrm.gamma[c0 + c] * nrm.rstd[sg], be = nrm.beta[c0 + c];
#pragma unroll
        for (int yy = 0; yy < W; ++yy) q[yy] = fmaf(q[yy] - mu, ga, be);
      }
    }
    fft_sfor<0, W>([&](auto YY) __attribute__((always_inline)) {
      constexpr int yy = decltype(YY)::value;
      float v = ur[fft_pos<W>(yy)] * scale;
      if (res) v += q[yy];
      y[base + (long long)(xr * W + yy) * E + c] = v;
    });
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 128 x 128 latent grids (1024^2 fields at patch 8 - the largest resolution utils/griddataset.py:35 lists; round 4).  A
// 128-point complex line is 256 registers - too many to keep next to the butterflies - so every line is split by ONE
// decimation-in-frequency stage applied while the inputs are loaded:
//      X[2k]   = FFT64(x[n] + x[n+64])[k]            X[2k+1] = FFT64((x[n] - x[n+64]) w^n)[k],   w = e^{-+2 pi i / 128}
// i.e. two independent 64-point register FFTs (fft_regs<64>, the 512^2 kernels' line) per 128-point line; a thread runs
// them one after the other.  Same contract, passes and LDS plan as the kernels above; only the KEPT ky rows go through
// LDS ([my][128][2][CC]: 4 channels per workgroup up to 36 kept columns, 2 up to all 65).
// ---------------------------------------------------------------------------------------------------------------------
template <int CC>
__global__ __launch_bounds__(256) void rfft2_128_kernel(const float* __restrict__ x, float* __restrict__ spec, int E,
                                                        int nb, int mx, int my, int colw, float scale) {
  constexpr int H = 128, W = 128, HN = 64;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Z = sm;                                            // [my][H][2][CC]
  const int b = blockIdx.y, c0 = xcd_chunk(blockIdx.x, gridDim.x) * CC;
  const int tid = threadIdx.x;
  const int bs = E / nb;
  const float* xb = x + (long long)b * H * W * E + c0;

  // pass 1: rows (real -> half complex).  Even ky from s[n] = x[n] + x[n+64], odd ky from (x[n] - x[n+64]) w^n
  for (int it = tid; it < H * CC; it += 256) {
    const int c = it % CC, xr = it / CC;
    float d[HN], re[HN], im[HN];
#pragma unroll
    for (int n = 0; n < HN; ++n) {
      const float a = xb[(long long)(xr * W + n) * E + c], q = xb[(long long)(xr * W + n + HN) * E + c];
      re[n] = a + q;
      d[n] = a - q;
      im[n] = 0.f;                                           // real input: the zero half folds away
    }
    fft_regs<HN, -1>(re, im);
    fft_sfor<0, HN>([&](auto K) __attribute__((always_inline)) {
      constexpr int k = decltype(K)::value, ky = 2 * k;
      if (ky < my) {
        Z[((ky * H + xr) * 2 + 0) * CC + c] = re[brev<HN>(k)];
        Z[((ky * H + xr) * 2 + 1) * CC + c] = im[brev<HN>(k)];
      }
    });
    // (ky = 64, the Nyquist column, is k = 32 of the even half: written above when it is kept)
    fft_sfor<0, HN>([&](auto Nn) __attribute__((always_inline)) {
      constexpr int n = decltype(Nn)::value;
      re[n] = d[n] * Twid<128>::c(n);
      im[n] = -d[n] * Twid<128>::s(n);
    });
    fft_regs<HN, -1>(re, im);
    fft_sfor<0, HN>([&](auto K) __attribute__((always_inline)) {
      constexpr int k = decltype(K)::value, ky = 2 * k + 1;
      if (ky < my) {
        Z[((ky * H + xr) * 2 + 0) * CC + c] = re[brev<HN>(k)];
        Z[((ky * H + xr) * 2 + 1) * CC + c] = im[brev<HN>(k)];
      }
    });
  }
  __syncthreads();

  // pass 2: columns (complex -> complex), kept modes only; item = (ky, c, parity of kx)
  for (int it = tid; it < my * CC * 2; it += 256) {
    const int c = it % CC, par = (it / CC) & 1, ky = it / (2 * CC);
    float zr[HN], zi[HN];
    fft_sfor<0, HN>([&](auto Nn) __attribute__((always_inline)) {
      constexpr int n = decltype(Nn)::value;
      const float ar = Z[((ky * H + n) * 2 + 0) * CC + c], ai = Z[((ky * H + n) * 2 + 1) * CC + c];
      const float br = Z[((ky * H + n + HN) * 2 + 0) * CC + c], bi = Z[((ky * H + n + HN) * 2 + 1) * CC + c];
      const float dr = ar - br, di = ai - bi;
      constexpr float cw = Twid<128>::c(n), sw = -Twid<128>::s(n);        // w^n = e^{-2 pi i n / 128}
      zr[n] = par ? dr * cw - di * sw : ar + br;
      zi[n] = par ? dr * sw + di * cw : ai + bi;
    });
    const int chn = c0 + c;
    const int blk = chn / bs, ci = chn % bs;
    const float wgt = scale * colw_f(colw, ky, W);
    float* out = spec + (((long long)b * mx * my + ky) * nb + blk) * 2 * bs + ci;
    const long long kxstride = (long long)my * nb * 2 * bs;
    fft_regs<HN, -1>(zr, zi);
    fft_sfor<0, HN>([&](auto K) __attribute__((always_inline)) {
      constexpr int k = decltype(K)::value;
      const int kx = 2 * k + par;
      if (kx < mx) {
        out[kx * kxstride] = zr[brev<HN>(k)] * wgt;
        out[kx * kxstride + bs] = zi[brev<HN>(k)] * wgt;
      }
    });
  }
}

// spec[B,mx,my,nb,2,bs] (+ res[B,H*W,E]) -> y[B,H*W,E], 128 x 128
template <int CC>
__global__ __launch_bounds__(256) void irfft2_128_kernel(const float* __restrict__ spec, const float* __restrict__ res,
                                                         float* __restrict__ y, int E, int nb, int mx, int my, int colw,
                                                         float scale) {
  constexpr int H = 128, W = 128, HN = 64;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* U = sm;                                            // [H][my][2][CC]
  const int b = blockIdx.y, c0 = xcd_chunk(blockIdx.x, gridDim.x) * CC;
  const int tid = threadIdx.x;
  const int bs = E / nb;

  // pass A: columns, U[x, ky] = sum_kx S[kx, ky] e^{+2 pi i kx x / H} times the column weight; item = (ky, c, parity of x)
  for (int it = tid; it < my * CC * 2; it += 256) {
    const int c = it % CC, par = (it / CC) & 1, ky = it / (2 * CC);
    const int chn = c0 + c;
    const int blk = chn / bs, ci = chn % bs;
    const float* in = spec + (((long long)b * mx * my + ky) * nb + blk) * 2 * bs + ci;
    const long long kxstride = (long long)my * nb * 2 * bs;
    float sr[HN], si[HN];
    fft_sfor<0, HN>([&](auto Kk) __attribute__((always_inline)) {
      constexpr int k = decltype(Kk)::value;
      const int k0 = k < mx ? k : mx - 1, k1 = k + HN < mx ? k + HN : mx - 1;   // clamped addresses, selected values
      float ar = in[k0 * kxstride], ai = in[k0 * kxstride + bs];
      float br = in[k1 * kxstride], bi = in[k1 * kxstride + bs];
      ar = k < mx ? ar : 0.f;
      ai = k < mx ? ai : 0.f;
      br = k + HN < mx ? br : 0.f;
      bi = k + HN < mx ? bi : 0.f;
      const float dr = ar - br, di = ai - bi;
      constexpr float cw = Twid<128>::c(k), sw = Twid<128>::s(k);         // w^k = e^{+2 pi i k / 128}
      sr[k] = par ? dr * cw - di * sw : ar + br;
      si[k] = par ? dr * sw + di * cw : ai + bi;
    });
    const float wgt = colw_f(colw, ky, W);
    fft_regs<HN, 1>(sr, si);
    fft_sfor<0, HN>([&](auto M) __attribute__((always_inline)) {
      constexpr int m = decltype(M)::value;
      const int xr = 2 * m + par;
      U[((xr * my + ky) * 2 + 0) * CC + c] = sr[brev<HN>(m)] * wgt;
      U[((xr * my + ky) * 2 + 1) * CC + c] = si[brev<HN>(m)] * wgt;
    });
  }
  __syncthreads();

  // pass B: rows (half complex -> real), + residual; item = (xr, c, parity of the output column)
  const long long base = (long long)b * H * W * E + c0;
  for (int it = tid; it < H * CC * 2; it += 256) {
    const int c = it % CC, par = (it / CC) & 1, xr = it / (2 * CC);
    float ur[HN], ui[HN];
    // one-sided spectrum V[ky], ky < my <= 65 (zero above): a[k] = V[k] + V[k+64], b[k] = (V[k] - V[k+64]) w^k; V[k+64] is
    // non-zero for k = 0 only (the Nyquist column, when it is kept)
    fft_sfor<0, HN>([&](auto Kk) __attribute__((always_inline)) {
      constexpr int k = decltype(Kk)::value;
      const int kc = k < my ? k : my - 1;
      float ar = U[((xr * my + kc) * 2 + 0) * CC + c], ai = U[((xr * my + kc) * 2 + 1) * CC + c];
      ar = k < my ? ar : 0.f;
      ai = k < my ? ai : 0.f;
      float br = 0.f, bi = 0.f;
      if constexpr (k == 0) {
        const int kn = HN < my ? HN : my - 1;
        br = U[((xr * my + kn) * 2 + 0) * CC + c];
        bi = U[((xr * my + kn) * 2 + 1) * CC + c];
        br = HN < my ? br : 0.f;
        bi = HN < my ? bi : 0.f;
      }
      const float dr = ar - br, di = ai - bi;
      constexpr float cw = Twid<128>::c(k), sw = Twid<128>::s(k);
      ur[k] = par ? dr * cw - di * sw : ar + br;
      ui[k] = par ? dr * sw + di * cw : ai + bi;
    });
    fft_regs<HN, 1>(ur, ui);                               // y[2m + par] = Re(IFFT64(.)[m])
    fft_sfor<0, HN>([&](auto M) __attribute__((always_inline)) {
      constexpr int m = decltype(M)::value;
      const int yy = 2 * m + par;
      float v = ur[brev<HN>(m)] * scale;
      if (res) v += res[base + (long long)(xr * W + yy) * E + c];
      y[base + (long long)(xr * W + yy) * E + c] = v;
    });
  }
}

#ifndef DPOT_DFT_NO_KERNELS   // (csrc/gn_dft.hip uses the register FFTs above only)
template <int H, int W, int CC>
static int launch_rfft2_fast(const float* x, float* spec, int B, int E, int nb, int mx, int my, int colw, float scale,
                             hipStream_t s, const DftNorm nrm) {
  constexpr size_t lds = sizeof(float) * ((size_t)(W / 2 + 1) * H * 2 * CC);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rfft2_fast_kernel<H, W, CC>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((rfft2_fast_kernel<H, W, CC>), dim3(E / CC, B), dim3(256), lds, s, x, spec, E, nb, mx, my, colw,
                     scale, nrm);
  return check_launch("rfft2_fast_kernel");
}
template <int H, int W, int CC>
static int launch_irfft2_fast(const float* spec, const float* res, float* y, int B, int E, int nb, int mx, int my,
                              int colw, float scale, hipStream_t s, const DftNorm nrm) {
  constexpr size_t lds = sizeof(float) * ((size_t)(W / 2 + 1) * H * 2 * CC);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(irfft2_fast_kernel<H, W, CC>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((irfft2_fast_kernel<H, W, CC>), dim3(E / CC, B), dim3(256), lds, s, spec, res, y, E, nb, mx, my,
                     colw, scale, nrm);
  return check_launch("irfft2_fast_kernel");
}

template <int CC>
static int launch_rfft2_128(const float* x, float* spec, int B, int E, int nb, int mx, int my, int colw, float scale,
                            hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)my * 128 * 2 * CC);
  const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(rfft2_128_kernel<CC>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  DPOT_REQUIRE(ae == hipSuccess, "rfft2_128: cannot raise the dynamic LDS limit to 160 KiB (%s)", hipGetErrorString(ae));
  hipLaunchKernelGGL((rfft2_128_kernel<CC>), dim3(E / CC, B), dim3(256), lds, s, x, spec, E, nb, mx, my, colw, scale);
  return check_launch("rfft2_128_kernel");
}
template <int CC>
static int launch_irfft2_128(const float* spec, const float* res, float* y, int B, int E, int nb, int mx, int my, int colw,
                             float scale, hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)my * 128 * 2 * CC);
  const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(irfft2_128_kernel<CC>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  DPOT_REQUIRE(ae == hipSuccess, "irfft2_128: cannot raise the dynamic LDS limit to 160 KiB (%s)", hipGetErrorString(ae));
  hipLaunchKernelGGL((irfft2_128_kernel<CC>), dim3(E / CC, B), dim3(256), lds, s, spec, res, y, E, nb, mx, my, colw,
                     scale);
  return check_launch("irfft2_128_kernel");
}

// returns 1 if a fast kernel was launched (rc in *rc), 0 if the shape has no fast path
static inline int try_rfft2_fast(const float* x, float* spec, int B, int h, int w, int E, int nb, int mx, int my,
                                 int colw, float scale, hipStream_t s, int* rc,
                                 const DftNorm nrm = DftNorm{nullptr, nullptr, nullptr, nullptr, 0}) {
  constexpr int forced = 0;      // (rounds 3-5: DPOT_DFT_CC forced a channel-chunk width for the sweeps in profiles/r05_dft_cc_L.txt)
  if (h == 16 && w == 16) {
    if (forced == 16 && E % 16 == 0) { *rc = launch_rfft2_fast<16, 16, 16>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    // 64-channel slabs only when that still gives every CU >= 2 workgroups (latency hiding); else 32-channel slabs
    if (forced != 32 && E % 64 == 0 && ((long long)B * (E / 64) >= 512 || forced == 64)) { *rc = launch_rfft2_fast<16, 16, 64>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (E % 32 == 0) { *rc = launch_rfft2_fast<16, 16, 32>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 8 && w == 8) {
    if (E % 64 == 0 && (long long)B * (E / 64) >= 512) { *rc = launch_rfft2_fast<8, 8, 64>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (E % 32 == 0) { *rc = launch_rfft2_fast<8, 8, 32>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 64 && w == 64) {
    // 64-point lines in registers (128 values per line item): 8-channel chunks, 132 KiB of LDS for the half-complex plane
    if (E % 8 == 0) { *rc = launch_rfft2_fast<64, 64, 8>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 32 && w == 32) {
    // 32-channel chunks (full 128-byte lines per token, 139 KiB of LDS: one workgroup per CU) when that still gives
    // >= 128 workgroups: 18.0 against 22.7 us at DPOT-L B = 4 (the inverse is FASTER with 16: 29.8 against 42.5 us)
    // 12-channel chunks: 512 balanced workgroups at DPOT-L B = 4 (see the inverse below) - 15.7 us (32: 18.1, 16: 22.7)
    // round 5, DPOT-L at batch 16: with >= 2 rounds of 32-channel workgroups the full lines win again (step 90.40 -> 90.22,
    // 90.63 -> 90.11 ms, profiles/r05_dft_cc_L.txt; 16-channel chunks: +1.0 ms)
    if (forced != 16 && forced != 12 && E % 32 == 0 && (long long)B * (E / 32) >= 512) { *rc = launch_rfft2_fast<32, 32, 32>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (forced != 16 && forced != 32 && E % 12 == 0 && (E / nb) % 12 == 0 && (long long)B * (E / 12) >= 256) { *rc = launch_rfft2_fast<32, 32, 12>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (forced != 16 && E % 32 == 0 && (long long)B * (E / 32) >= 128) { *rc = launch_rfft2_fast<32, 32, 32>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (E % 16 == 0) { *rc = launch_rfft2_fast<32, 32, 16>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 24 && w == 24) {
    // round 5: 3 * 2^k lines (192^2 / 384^2 / 768^2 fields at patch 8) - a radix-3 stage in front of three 2^k-point register FFTs
    if (E % 32 == 0) { *rc = launch_rfft2_fast<24, 24, 32>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (E % 8 == 0) { *rc = launch_rfft2_fast<24, 24, 8>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 48 && w == 48) {
    if (E % 16 == 0) { *rc = launch_rfft2_fast<48, 48, 16>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (E % 4 == 0) { *rc = launch_rfft2_fast<48, 48, 4>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 96 && w == 96) {
    if (E % 4 == 0) { *rc = launch_rfft2_fast<96, 96, 4>(x, spec, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 128 && w == 128 && !nrm.mean) {
    // 1024^2 fields at patch 8: two 64-point register FFTs per line; the kept ky rows in LDS ([my][128][2][CC])
    if (E % 4 == 0 && my <= 36) { *rc = launch_rfft2_128<4>(x, spec, B, E, nb, mx, my, colw, scale, s); return 1; }
    if (E % 2 == 0) { *rc = launch_rfft2_128<2>(x, spec, B, E, nb, mx, my, colw, scale, s); return 1; }
  }
  return 0;
}
static inline int try_irfft2_fast(const float* spec, const float* res, float* y, int B, int h, int w, int E, int nb,
                                  int mx, int my, int colw, float scale, hipStream_t s, int* rc,
                                  const DftNorm nrm = DftNorm{nullptr, nullptr, nullptr, nullptr, 0}) {
  constexpr int forced = 0;      // (rounds 3-5: DPOT_DFT_CC forced a channel-chunk width for the sweeps in profiles/r05_dft_cc_L.txt)
  if (h == 16 && w == 16) {
    if (forced == 16 && E % 16 == 0) { *rc = launch_irfft2_fast<16, 16, 16>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (forced != 32 && E % 64 == 0 && ((long long)B * (E / 64) >= 512 || forced == 64)) { *rc = launch_irfft2_fast<16, 16, 64>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (E % 32 == 0) { *rc = launch_irfft2_fast<16, 16, 32>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 8 && w == 8) {
    if (E % 64 == 0 && (long long)B * (E / 64) >= 512) { *rc = launch_irfft2_fast<8, 8, 64>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (E % 32 == 0) { *rc = launch_irfft2_fast<8, 8, 32>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 64 && w == 64) {
    if (E % 8 == 0) { *rc = launch_irfft2_fast<64, 64, 8>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 32 && w == 32) {
    // 12-channel chunks when the blocks allow it: 512 workgroups = 2 per CU at DPOT-L B = 4 (E = 1536) instead of the
    // 384 of 16-channel chunks (1.5 per CU: half the CUs run two in a row) - 20.4 against 29.8 us
    // (32-channel chunks for the inverse, one workgroup per CU: +4 ms on the DPOT-L step at batch 16, profiles/r05_dft_cc_L.txt)
    if (forced != 16 && E % 12 == 0 && (E / nb) % 12 == 0 && (long long)B * (E / 12) >= 256) { *rc = launch_irfft2_fast<32, 32, 12>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (E % 16 == 0) { *rc = launch_irfft2_fast<32, 32, 16>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 24 && w == 24) {
    if (E % 32 == 0) { *rc = launch_irfft2_fast<24, 24, 32>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (E % 8 == 0) { *rc = launch_irfft2_fast<24, 24, 8>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 48 && w == 48) {
    if (E % 16 == 0) { *rc = launch_irfft2_fast<48, 48, 16>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
    if (E % 4 == 0) { *rc = launch_irfft2_fast<48, 48, 4>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 96 && w == 96) {
    if (E % 4 == 0) { *rc = launch_irfft2_fast<96, 96, 4>(spec, res, y, B, E, nb, mx, my, colw, scale, s, nrm); return 1; }
  } else if (h == 128 && w == 128 && !nrm.mean) {
    if (E % 4 == 0 && my <= 36) { *rc = launch_irfft2_128<4>(spec, res, y, B, E, nb, mx, my, colw, scale, s); return 1; }
    if (E % 2 == 0) { *rc = launch_irfft2_128<2>(spec, res, y, B, E, nb, mx, my, colw, scale, s); return 1; }
  }
  return 0;
}
#endif  // DPOT_DFT_NO_KERNELS

}  // namespace dpot
