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

// ---- sub-pixel form of the convolution over a nearest-upsampled tensor ----
// Along an axis with factor 2 the three taps of an output pixel of parity a fall on two low-res
// rows: a = 0: taps {0, 1} -> row 0, {2} -> row 1;  a = 1: {0} -> row 0, {1, 2} -> row 1.
// Axes with factor 1 keep their taps (one "phase", low tap = tap).
__device__ __forceinline__ int sp_low_tap(int f, int a, int tap) {
  if (f == 1) return tap;
  return a == 0 ? (tap == 2 ? 1 : 0) : (tap == 0 ? 0 : 1);
}

struct SpGeom {
  int cout, cin, C0, C1, N;      // N = padded cout (rows per phase)
  int k[3], f[3], zk[3];         // kernel, factors, low-res kernel per axis (z, y, x)
};

// w (cout, cin, kd, kh, kw) -> w_skip (cout, C0, taps) and weff (P*N, C1, ztaps), weff = the taps of
// the upsampled half summed per phase; rows of padded output channels are zero
__global__ void subpixel_split_kernel(const float* __restrict__ w, float* __restrict__ w_skip,
                                      float* __restrict__ weff, SpGeom g, long long n_skip, long long n_eff) {
  const int taps = g.k[0] * g.k[1] * g.k[2], ztaps = g.zk[0] * g.zk[1] * g.zk[2];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_skip + n_eff;
       i += (long long)gridDim.x * blockDim.x) {
    if (i < n_skip) {
      const int tap = (int)(i % taps);
      const long long t = i / taps;
      const int c = (int)(t % g.C0), n = (int)(t / g.C0);
      w_skip[i] = w[((long long)n * g.cin + c) * taps + tap];
      continue;
    }
    const long long j = i - n_skip;
    const int zt = (int)(j % ztaps);
    const long long t = j / ztaps;
    const int c = (int)(t % g.C1);
    const int row = (int)(t / g.C1);
    const int n = row % g.N, ph = row / g.N;
    float v = 0.f;
    if (n < g.cout) {
      const int r[3] = {zt / (g.zk[1] * g.zk[2]), (zt / g.zk[2]) % g.zk[1], zt % g.zk[2]};
      const int a[3] = {ph / (g.f[1] * g.f[2]), (ph / g.f[2]) % g.f[1], ph % g.f[2]};
      const float* src = w + ((long long)n * g.cin + g.C0 + c) * taps;
      for (int d = 0; d < g.k[0]; ++d) {
        if (sp_low_tap(g.f[0], a[0], d) != r[0]) continue;
        for (int h = 0; h < g.k[1]; ++h) {
          if (sp_low_tap(g.f[1], a[1], h) != r[1]) continue;
          for (int x = 0; x < g.k[2]; ++x)
            if (sp_low_tap(g.f[2], a[2], x) == r[2]) v += src[(d * g.k[1] + h) * g.k[2] + x];
        }
      }
    }
    weff[j] = v;
  }
}

// the adjoint: g_skip (cout, C0, taps), g_eff (P*N, C1, ztaps) -> gw (cout, cin, taps)
__global__ void subpixel_fold_kernel(const float* __restrict__ g_skip, const float* __restrict__ g_eff,
                                     float* __restrict__ gw, SpGeom g, long long total) {
  const int taps = g.k[0] * g.k[1] * g.k[2], ztaps = g.zk[0] * g.zk[1] * g.zk[2];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const long long t = i / taps;
    const int c = (int)(t % g.cin), n = (int)(t / g.cin);
    if (c < g.C0) {
      gw[i] = g_skip[((long long)n * g.C0 + c) * taps + tap];
      continue;
    }
    const int d = tap / (g.k[1] * g.k[2]), h = (tap / g.k[2]) % g.k[1], x = tap % g.k[2];
    float v = 0.f;
    for (int a0 = 0; a0 < g.f[0]; ++a0)
      for (int a1 = 0; a1 < g.f[1]; ++a1)
        for (int a2 = 0; a2 < g.f[2]; ++a2) {
          const int ph = (a0 * g.f[1] + a1) * g.f[2] + a2;
          const int zt = (sp_low_tap(g.f[0], a0, d) * g.zk[1] + sp_low_tap(g.f[1], a1, h)) * g.zk[2] +
                         sp_low_tap(g.f[2], a2, x);
          v += g_eff[(((long long)ph * g.N + n) * g.C1 + (c - g.C0)) * ztaps + zt];
        }
    gw[i] = v;
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
  if (mode == CLX_PACK_WINO4_ADJOINT) {
    CLX_REQUIRE(taps == 9 || taps == 27, "clx_pack_weights: the adjoint form exists for 3x3 and 3x3x3 kernels");
    clx_wino_pack(w, wp, cout, cin, cin_pad, cout_pad, 2, 4, 3, taps == 27 ? 3 : 1, (hipStream_t)stream);
    CLX_CHECK_LAUNCH("clx_pack_weights(winograd adjoint)");
    return CLX_OK;
  }
  if (mode == CLX_PACK_WINO4_FUSED) {
    CLX_REQUIRE((taps == 9 || taps == 4) && cout_pad % 64 == 0 && cin_pad % 8 == 0,
                "clx_pack_weights: the fused Winograd layout exists for 2-D 3x3 / 2x2 kernels with cout_pad %% 64 == 0 and "
                "cin_pad %% 8 == 0");
    clx_wino_pack(w, wp, cout, cin, cin_pad, cout_pad, 3, 4, taps == 9 ? 3 : 2, 1, (hipStream_t)stream);
    CLX_CHECK_LAUNCH("clx_pack_weights(winograd fused)");
    return CLX_OK;
  }
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

static int sp_geom(SpGeom& g, int cout, int cin, int C0, int N, const int* k, const int* f, const char* who) {
  CLX_REQUIRE(cout > 0 && cin > C0 && C0 > 0 && N >= cout, "%s: bad channel counts", who);
  g.cout = cout; g.cin = cin; g.C0 = C0; g.C1 = cin - C0; g.N = N;
  for (int d = 0; d < 3; ++d) {
    CLX_REQUIRE((f[d] == 1 || (f[d] == 2 && k[d] == 3)) && k[d] >= 1 && k[d] <= 3,
                "%s: factor must be 1, or 2 with a 3-tap axis", who);
    g.k[d] = k[d]; g.f[d] = f[d]; g.zk[d] = f[d] == 2 ? 2 : k[d];
  }
  return CLX_OK;
}

extern "C" int clx_subpixel_split_weights(const float* w, float* w_skip, float* weff, int cout, int cin,
                                          int C0, int N, int kd, int kh, int kw, int fz, int fy, int fx,
                                          clx_stream stream) {
  CLX_REQUIRE(w && w_skip && weff, "clx_subpixel_split_weights: null pointer");
  SpGeom g;
  const int k[3] = {kd, kh, kw}, f[3] = {fz, fy, fx};
  const int rc = sp_geom(g, cout, cin, C0, N, k, f, "clx_subpixel_split_weights");
  if (rc) return rc;
  const long long n_skip = (long long)cout * C0 * kd * kh * kw;
  const long long n_eff = (long long)fz * fy * fx * N * g.C1 * g.zk[0] * g.zk[1] * g.zk[2];
  subpixel_split_kernel<<<grid_for(n_skip + n_eff, 256), 256, 0, (hipStream_t)stream>>>(w, w_skip, weff, g, n_skip, n_eff);
  CLX_CHECK_LAUNCH("clx_subpixel_split_weights");
  return CLX_OK;
}

extern "C" int clx_subpixel_fold_grads(const float* g_skip, const float* g_eff, float* gw, int cout, int cin,
                                       int C0, int N, int kd, int kh, int kw, int fz, int fy, int fx,
                                       clx_stream stream) {
  CLX_REQUIRE(g_skip && g_eff && gw, "clx_subpixel_fold_grads: null pointer");
  SpGeom g;
  const int k[3] = {kd, kh, kw}, f[3] = {fz, fy, fx};
  const int rc = sp_geom(g, cout, cin, C0, N, k, f, "clx_subpixel_fold_grads");
  if (rc) return rc;
  const long long total = (long long)cout * cin * kd * kh * kw;
  subpixel_fold_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(g_skip, g_eff, gw, g, total);
  CLX_CHECK_LAUNCH("clx_subpixel_fold_grads");
  return CLX_OK;
}
