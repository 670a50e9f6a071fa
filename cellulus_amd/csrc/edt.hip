// Exact squared Euclidean distance transform (integer arithmetic) and the
// grow/shrink post-processing built on it.
//
// scipy.ndimage.distance_transform_edt(a) returns, for every non-zero element
// of `a`, the distance to the nearest zero element.  cellulus/segment.py:41-51
// only compares those distances with small integers (grow_distance,
// shrink_distance), and d < g  <=>  d^2 < g^2 for the exact integer d^2, so the
// whole post-processing is done on squared distances — bit-exact.
//
// Separable min-plus: pass X finds the squared distance to the nearest zero in
// the same row by an outward search; passes Y and Z take
// min_{q}( (p-q)^2 + g[q] ) searching outward and stopping as soon as
// (p-q)^2 >= best (no farther candidate can win), which is exact.  The callers only
// compare distances with a small bound, so every search is additionally capped at
// `cap` steps: results < cap^2 are exact, larger ones are reported as >= cap^2.
#include "clx_common.h"

namespace {

constexpr int EDT_INF = 1 << 29;

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

__global__ void edt_pass_x(const unsigned char* __restrict__ in, int* __restrict__ g, int X,
                           long long npix, int cap, int* __restrict__ any_zero) {
  bool seen = false;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % X);
    const long long row = i - x;
    int best = EDT_INF;
    seen = seen || (in[i] == 0);
    for (int d = 0; d < cap; ++d) {
      const int xl = x - d, xr = x + d;
      if (xl < 0 && xr >= X) break;
      if ((xl >= 0 && in[row + xl] == 0) || (xr < X && in[row + xr] == 0)) {
        best = d * d;
        break;
      }
    }
    g[i] = best;
  }
  if (__any(seen) && (threadIdx.x & 63) == 0) atomicOr(any_zero, 1);
}

// min over the axis with stride `stride` and extent `n` (axis index = (i / stride) % n).
// final != 0 (last pass): an image without any zero element has no finite
// distance; scipy then reports the distance to a phantom zero at index -1 of the
// FIRST axis (0 on the others) — reproduced here so the result stays bit-exact.
__global__ void edt_pass_axis(const int* __restrict__ g, int* __restrict__ out, int n,
                              long long stride, long long npix, int final, int Z, int Y, int X,
                              int cap, const int* __restrict__ any_zero) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)((i / stride) % n);
    int best = g[i];
    for (int d = 1; d < cap; ++d) {
      const int d2 = d * d;
      if (d2 >= best) break;
      const int lo = p - d, hi = p + d;
      if (lo < 0 && hi >= n) break;
      if (lo >= 0) best = min(best, d2 + g[i - (long long)d * stride]);
      if (hi < n) best = min(best, d2 + g[i + (long long)d * stride]);
    }
    best = min(best, EDT_INF);
    if (final && best >= EDT_INF && *any_zero == 0) {
      const int x = (int)(i % X);
      const long long t = i / X;
      const int y = (int)(t % Y);
      const int z = (int)(t / Y);
      best = (Z > 1) ? (z + 1) * (z + 1) + y * y + x * x : (y + 1) * (y + 1) + x * x;
    }
    out[i] = best;
  }
}

__global__ void mask_eq_zero(const int* __restrict__ seg, unsigned char* __restrict__ m, long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x)
    m[i] = (seg[i] == 0) ? 1 : 0;
}

__global__ void mask_lt(const int* __restrict__ d, int bound, unsigned char* __restrict__ m, long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x)
    m[i] = (d[i] < bound) ? 1 : 0;
}

__global__ void zero_where_lt(int* __restrict__ seg, const int* __restrict__ d, int bound, long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x)
    if (d[i] < bound) seg[i] = 0;
}

// tmp: npix ints + 1 flag int.  cap <= 0: unlimited search.
int edt_run(const unsigned char* in, int* out, int Z, int Y, int X, int* tmp, int cap, hipStream_t st) {
  const long long npix = (long long)Z * Y * X;
  const int grid = grid_for(npix, 256);
  int* any_zero = tmp + npix;
  const int big = 1 << 20;
  const int cx = cap > 0 ? (cap < X ? cap : X) : X;
  const int cy = cap > 0 ? cap : big, cz = cap > 0 ? cap : big;
  if (hipMemsetAsync(any_zero, 0, sizeof(int), st) != hipSuccess) return CLX_ERR_LAUNCH;
  // X pass -> tmp ; Y pass tmp -> out ; Z pass out -> tmp -> copy (3-D only)
  edt_pass_x<<<grid, 256, 0, st>>>(in, tmp, X, npix, cx, any_zero);
  edt_pass_axis<<<grid, 256, 0, st>>>(tmp, out, Y, (long long)X, npix, Z > 1 ? 0 : 1, Z, Y, X,
                                      cy < Y ? cy : Y, any_zero);
  if (Z > 1) {
    edt_pass_axis<<<grid, 256, 0, st>>>(out, tmp, Z, (long long)X * Y, npix, 1, Z, Y, X,
                                        cz < Z ? cz : Z, any_zero);
    if (hipMemcpyAsync(out, tmp, (size_t)npix * sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess)
      return CLX_ERR_LAUNCH;
  }
  return CLX_OK;
}

}  // namespace

extern "C" size_t clx_edt_workspace(long long npix) { return (size_t)(npix + 4) * sizeof(int); }

extern "C" int clx_edt_sq(const unsigned char* in, int* out, int Z, int Y, int X, int cap,
                          void* workspace, clx_stream stream) {
  CLX_REQUIRE(in && out && workspace, "clx_edt_sq: null pointer");
  CLX_REQUIRE(Z > 0 && Y > 0 && X > 0, "clx_edt_sq: bad extents");
  CLX_REQUIRE((long long)Z * Y * X < (1ll << 31), "clx_edt_sq: too many pixels");
  CLX_REQUIRE(Z < 16384 && Y < 16384 && X < 16384, "clx_edt_sq: extent too large for int32 distances");
  const int rc = edt_run(in, out, Z, Y, X, (int*)workspace, cap, (hipStream_t)stream);
  if (rc) { clx_set_error("clx_edt_sq: copy failed"); return rc; }
  CLX_CHECK_LAUNCH("clx_edt_sq");
  return CLX_OK;
}

extern "C" int clx_grow_shrink(int* seg, int Z, int Y, int X, int grow, int shrink,
                               void* workspace, clx_stream stream) {
  CLX_REQUIRE(seg && workspace, "clx_grow_shrink: null pointer");
  CLX_REQUIRE(Z > 0 && Y > 0 && X > 0, "clx_grow_shrink: bad extents");
  CLX_REQUIRE(Z < 16384 && Y < 16384 && X < 16384, "clx_grow_shrink: extent too large");
  const long long npix = (long long)Z * Y * X;
  CLX_REQUIRE(npix < (1ll << 31), "clx_grow_shrink: too many pixels");
  int* tmp = (int*)workspace;
  int* dist = tmp + npix + 4;
  unsigned char* mask = (unsigned char*)(dist + npix);
  const int grid = grid_for(npix, 256);
  hipStream_t st = (hipStream_t)stream;
  // d1 = edt(seg == 0); expanded = d1 < grow
  mask_eq_zero<<<grid, 256, 0, st>>>(seg, mask, npix);
  // only "d < grow" / "d < shrink" are needed: cap the searches at those radii
  int rc = edt_run(mask, dist, Z, Y, X, tmp, grow > 0 ? grow : 1, st);
  if (rc) { clx_set_error("clx_grow_shrink: copy failed"); return rc; }
  // sqrt(d1) < grow  <=>  d1 < grow^2   (grow <= 0: nothing is expanded)
  mask_lt<<<grid, 256, 0, st>>>(dist, grow > 0 ? grow * grow : 0, mask, npix);
  // d2 = edt(expanded); seg[d2 < shrink] = 0
  rc = edt_run(mask, dist, Z, Y, X, tmp, shrink > 0 ? shrink : 1, st);
  if (rc) { clx_set_error("clx_grow_shrink: copy failed"); return rc; }
  zero_where_lt<<<grid, 256, 0, st>>>(seg, dist, shrink > 0 ? shrink * shrink : 0, npix);
  CLX_CHECK_LAUNCH("clx_grow_shrink");
  return CLX_OK;
}
