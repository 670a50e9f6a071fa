// Seeding of the mean-shift with use_seeds = true (cellulus/detect.py:128-132):
//   offset_magnitude = np.linalg.norm(embeddings_centered[:-1], axis=0)
//   smooth           = scipy.ndimage.gaussian_filter(offset_magnitude, sigma=2)
//   seeds            = skimage.feature.peak_local_max(-smooth)
// in float64 with the libraries' own operation order (this file is compiled with
// -ffp-contract=off), so the seeds — integer pixel coordinates — are the reference's:
//   norm:     sqrt((x0*x0 + x1*x1) + x2*x2)                       (numpy add.reduce over axis 0)
//   gaussian: per axis, in axis order,  t = x[c]*w[0];  t += (x[c-j] + x[c+j]) * w[j], j = r..1
//             (scipy NI_Correlate1D's symmetric branch), boundary mode "reflect" (d c b a | a b c d)
//   peaks:    x == max over the 3^ND neighbourhood (edge-clamped) and x > min(image), one pixel
//             of border excluded (skimage defaults: min_distance 1, exclude_border True)
// All three are HBM-streaming kernels (8 B per pixel in, 8 B out; the filter's 17 taps hit L2).
#include "clx_common.h"

namespace {

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

__global__ void magnitude_kernel(const double* __restrict__ emb, double* __restrict__ out, int ND, long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    double s = emb[i] * emb[i];
    for (int c = 1; c < ND; ++c) {
      const double v = emb[(long long)c * npix + i];
      s = s + v * v;
    }
    out[i] = sqrt(s);
  }
}

// scipy "reflect": the edge sample is repeated (..., x1, x0 | x0, x1, ... , xn-1 | xn-1, xn-2, ...)
__device__ __forceinline__ int reflect_index(int i, int n) {
  if (n == 1) return 0;
  const int period = 2 * n;
  i %= period;
  if (i < 0) i += period;
  return i < n ? i : period - 1 - i;
}

// one axis of the separable filter; `stride` = elements between neighbours along the axis, n = extent
__global__ void gaussian_axis_kernel(const double* __restrict__ in, double* __restrict__ out,
                                     const double* __restrict__ w, int radius, long long stride, int n,
                                     long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / stride) % n);
    const long long base = i - (long long)c * stride;
    double t = in[i] * w[0];
    // scipy's loop runs from the outermost tap inwards (jj = -size1 .. -1)
    if (c - radius >= 0 && c + radius < n) {
      for (int j = radius; j >= 1; --j) t += (in[i - j * stride] + in[i + j * stride]) * w[j];
    } else {
      for (int j = radius; j >= 1; --j)
        t += (in[base + (long long)reflect_index(c - j, n) * stride] +
              in[base + (long long)reflect_index(c + j, n) * stride]) * w[j];
    }
    out[i] = t;
  }
}

// peaks of `img` (already negated by the caller's convention: maxima are sought): appended as
// (raster index) to `peaks`, count in *npeaks; the host orders them
__global__ void peaks_kernel(const double* __restrict__ img, int Z, int Y, int X, const double* __restrict__ minmax,
                             int* __restrict__ peaks, int capacity, int* __restrict__ npeaks) {
  const long long npix = (long long)Z * Y * X;
  const double lowest = minmax[0];
  const bool three_d = Z > 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % X);
    const long long t = i / X;
    const int y = (int)(t % Y), z = (int)(t / Y);
    // exclude_border: one pixel along every axis of the image (a 2-D image has no z axis)
    if (x == 0 || x == X - 1 || y == 0 || y == Y - 1 || (three_d && (z == 0 || z == Z - 1))) continue;
    const double v = img[i];
    if (!(v > lowest)) continue;
    bool is_max = true;
    for (int dz = (three_d ? -1 : 0); dz <= (three_d ? 1 : 0) && is_max; ++dz)
      for (int dy = -1; dy <= 1 && is_max; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const double o = img[((long long)(z + dz) * Y + (y + dy)) * X + (x + dx)];   // interior: in range
          if (o > v) { is_max = false; break; }
        }
    if (is_max) {
      const int slot = atomicAdd(npeaks, 1);
      if (slot < capacity) peaks[slot] = (int)i;
    }
  }
}

__global__ void negate_kernel(const double* __restrict__ in, double* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = -in[i];
}

}  // namespace

extern "C" int clx_offset_magnitude(const double* emb, double* out, int ND, long long npix, clx_stream stream) {
  CLX_REQUIRE(emb && out && ND >= 1 && npix > 0, "clx_offset_magnitude: bad arguments");
  magnitude_kernel<<<grid_for(npix, 256), 256, 0, (hipStream_t)stream>>>(emb, out, ND, npix);
  CLX_CHECK_LAUNCH("clx_offset_magnitude");
  return CLX_OK;
}

extern "C" int clx_gaussian_filter_f64(const double* in, double* out, double* tmp, int Z, int Y, int X,
                                       const double* weights, int radius, clx_stream stream) {
  CLX_REQUIRE(in && out && tmp && weights, "clx_gaussian_filter_f64: null pointer");
  CLX_REQUIRE(Z > 0 && Y > 0 && X > 0 && radius >= 0, "clx_gaussian_filter_f64: bad extents");
  CLX_REQUIRE(in != out && in != tmp && out != tmp, "clx_gaussian_filter_f64: buffers must be distinct");
  const long long npix = (long long)Z * Y * X;
  CLX_REQUIRE(npix < (1ll << 40), "clx_gaussian_filter_f64: too many pixels");
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for(npix, 256);
  // axis order of scipy.ndimage.gaussian_filter: first axis first.  2-D: y, x.  3-D: z, y, x.
  // Ping-pong so that the last pass lands in `out`.
  if (Z > 1) {
    gaussian_axis_kernel<<<grid, 256, 0, st>>>(in, out, weights, radius, (long long)Y * X, Z, npix);
    gaussian_axis_kernel<<<grid, 256, 0, st>>>(out, tmp, weights, radius, (long long)X, Y, npix);
    gaussian_axis_kernel<<<grid, 256, 0, st>>>(tmp, out, weights, radius, 1, X, npix);
  } else {
    gaussian_axis_kernel<<<grid, 256, 0, st>>>(in, tmp, weights, radius, (long long)X, Y, npix);
    gaussian_axis_kernel<<<grid, 256, 0, st>>>(tmp, out, weights, radius, 1, X, npix);
  }
  CLX_CHECK_LAUNCH("clx_gaussian_filter_f64");
  return CLX_OK;
}

extern "C" int clx_negate_f64(const double* in, double* out, long long n, clx_stream stream) {
  CLX_REQUIRE(in && out && n > 0, "clx_negate_f64: bad arguments");
  negate_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(in, out, n);
  CLX_CHECK_LAUNCH("clx_negate_f64");
  return CLX_OK;
}

extern "C" int clx_peak_local_max(const double* img, int Z, int Y, int X, const double* minmax, int* peaks,
                                  int capacity, int* npeaks, clx_stream stream) {
  CLX_REQUIRE(img && minmax && peaks && npeaks, "clx_peak_local_max: null pointer");
  CLX_REQUIRE(Z > 0 && Y > 0 && X > 0 && capacity > 0, "clx_peak_local_max: bad extents");
  CLX_REQUIRE((long long)Z * Y * X < (1ll << 31), "clx_peak_local_max: too many pixels");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(npeaks, 0, sizeof(int), st) != hipSuccess) {
    clx_set_error("clx_peak_local_max: memset failed");
    return CLX_ERR_LAUNCH;
  }
  peaks_kernel<<<grid_for((long long)Z * Y * X, 256), 256, 0, st>>>(img, Z, Y, X, minmax, peaks, capacity, npeaks);
  CLX_CHECK_LAUNCH("clx_peak_local_max");
  return CLX_OK;
}
