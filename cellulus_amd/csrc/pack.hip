// Layout conversions at the model boundary: torch-layout conv weights <->
// K-contiguous packed operands of the MFMA kernels, NCHW <-> channels-last.
// All are small HBM-bound streaming kernels (weights total 38.5 MB at the
// benchmark config; boundary tensors have <= 4 channels).
#include "clx_common.h"

namespace {

// FWD:   wp[n][tap][c]  = w[n][c][tap]            (c >= cin -> 0)
// DGRAD: wp[c][tap][n]  = w[n][c][taps-1-tap]     (n >= cout, c >= cin -> 0)
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                    int cout, int cin, int taps, int cin_pad,
                                    int cout_pad, int mode, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (mode == CLX_PACK_FWD) {
      const int c = (int)(i % cin_pad);
      const long long t = i / cin_pad;
      const int tap = (int)(t % taps);
      const int n = (int)(t / taps);
      if (c < cin) v = w[((long long)n * cin + c) * taps + tap];
    } else {
      const int n = (int)(i % cout_pad);
      const long long t = i / cout_pad;
      const int tap = (int)(t % taps);
      const int c = (int)(t / taps);
      if (n < cout && c < cin) v = w[((long long)n * cin + c) * taps + (taps - 1 - tap)];
    }
    wp[i] = v;
  }
}

// dw[n][c][tap] = dwp[tap][n][c], dwp is [taps][rows][cin_pad]
__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw,
                                    int rows, int cin, int taps, int cin_pad, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const long long t = i / taps;
    const int c = (int)(t % cin);
    const int n = (int)(t / cin);
    dw[i] = dwp[((long long)tap * rows + n) * cin_pad + c];
  }
}

__global__ void planar_to_pixel_kernel(const float* __restrict__ planar, float* __restrict__ pixel,
                                       int C, long long n, int ld, long long total) {
  // one thread per (b, pixel): writes ld floats
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / n, q = i - b * n;
    const float* src = planar + b * C * n + q;
    float* dst = pixel + i * ld;
    for (int c = 0; c < ld; ++c) dst[c] = (c < C) ? src[(long long)c * n] : 0.f;
  }
}

__global__ void pixel_to_planar_kernel(const float* __restrict__ pixel, float* __restrict__ planar,
                                       int C, long long n, int ld, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / n, q = i - b * n;
    const float* src = pixel + i * ld;
    float* dst = planar + b * C * n + q;
    for (int c = 0; c < C; ++c) dst[(long long)c * n] = src[c];
  }
}

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int clx_pack_weights(const float* w, float* wp, int cout, int cin, int taps,
                                int cin_pad, int cout_pad, int mode, clx_stream stream) {
  CLX_REQUIRE(w && wp, "clx_pack_weights: null pointer");
  CLX_REQUIRE(cout > 0 && cin > 0 && taps > 0, "clx_pack_weights: bad extents");
  CLX_REQUIRE(cin_pad >= cin && cout_pad >= cout && cin_pad % 4 == 0 && cout_pad % 4 == 0,
              "clx_pack_weights: padded extents must be >= real and multiples of 4");
  if (mode == CLX_PACK_WINO_FWD || mode == CLX_PACK_WINO_DGRAD || mode == CLX_PACK_WINO4_FWD ||
      mode == CLX_PACK_WINO4_DGRAD) {
    const bool four = mode == CLX_PACK_WINO4_FWD || mode == CLX_PACK_WINO4_DGRAD;
    CLX_REQUIRE(taps == 9 || (four && (taps == 4 || taps == 27 || taps == 8)),
                "clx_pack_weights: Winograd packing needs a 3x3 kernel (F(4x4) modes: also 2x2, 3x3x3, 2x2x2)");
    const int ksize = (taps == 9 || taps == 27) ? 3 : 2, kd = (taps == 27 || taps == 8) ? ksize : 1;
    clx_wino_pack(w, wp, cout, cin, cin_pad, cout_pad, mode == CLX_PACK_WINO_DGRAD || mode == CLX_PACK_WINO4_DGRAD,
                  four ? 4 : 2, ksize, kd, (hipStream_t)stream);
    CLX_CHECK_LAUNCH("clx_pack_weights(winograd)");
    return CLX_OK;
  }
  CLX_REQUIRE(mode == CLX_PACK_FWD || mode == CLX_PACK_DGRAD, "clx_pack_weights: bad mode");
  const long long total = (mode == CLX_PACK_FWD) ? (long long)cout * taps * cin_pad
                                                 : (long long)cin_pad * taps * cout_pad;
  pack_weights_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      w, wp, cout, cin, taps, cin_pad, cout_pad, mode, total);
  CLX_CHECK_LAUNCH("clx_pack_weights");
  return CLX_OK;
}

extern "C" int clx_unpack_wgrad(const float* dwpack, float* dw, int cout, int cin, int taps,
                                int rows, int cin_pad, clx_stream stream) {
  CLX_REQUIRE(dwpack && dw, "clx_unpack_wgrad: null pointer");
  CLX_REQUIRE(cout > 0 && cin > 0 && taps > 0 && cin_pad >= cin && rows >= cout,
              "clx_unpack_wgrad: bad extents");
  const long long total = (long long)cout * cin * taps;
  unpack_wgrad_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      dwpack, dw, rows, cin, taps, cin_pad, total);
  CLX_CHECK_LAUNCH("clx_unpack_wgrad");
  return CLX_OK;
}

extern "C" int clx_planar_to_pixel(const float* planar, float* pixel, int B, int C,
                                   long long n, int ld, clx_stream stream) {
  CLX_REQUIRE(planar && pixel && B > 0 && C > 0 && n > 0 && ld >= C,
              "clx_planar_to_pixel: bad arguments");
  const long long total = (long long)B * n;
  planar_to_pixel_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      planar, pixel, C, n, ld, total);
  CLX_CHECK_LAUNCH("clx_planar_to_pixel");
  return CLX_OK;
}

extern "C" int clx_pixel_to_planar(const float* pixel, float* planar, int B, int C,
                                   long long n, int ld, clx_stream stream) {
  CLX_REQUIRE(planar && pixel && B > 0 && C > 0 && n > 0 && ld >= C,
              "clx_pixel_to_planar: bad arguments");
  const long long total = (long long)B * n;
  pixel_to_planar_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      pixel, planar, C, n, ld, total);
  CLX_CHECK_LAUNCH("clx_pixel_to_planar");
  return CLX_OK;
}
