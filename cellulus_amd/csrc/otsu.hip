// Histogram primitives behind skimage.filters.threshold_otsu as used on the
// std channel (cellulus/detect.py:88-91): min/max of an f64 image and
// numpy.histogram's uniform-bin index computation, reproduced step by step
// (scale, truncate, then the +-1 corrections against the linspace edges) so the
// counts are bit-exact.  Compiled with the default contraction but the index
// expression has no multiply-add pair to fuse.
#include "clx_common.h"

namespace {

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ void atomic_min_f64(double* addr, double v) {
  unsigned long long* a = (unsigned long long*)addr;
  unsigned long long old = *a;
  while (v < __longlong_as_double((long long)old)) {
    const unsigned long long assumed = old;
    old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
    if (old == assumed) break;
  }
}
__device__ __forceinline__ void atomic_max_f64(double* addr, double v) {
  unsigned long long* a = (unsigned long long*)addr;
  unsigned long long old = *a;
  while (v > __longlong_as_double((long long)old)) {
    const unsigned long long assumed = old;
    old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
    if (old == assumed) break;
  }
}

__global__ void minmax_init(double* mm) {
  mm[0] = __longlong_as_double(0x7ff0000000000000ll);   // +inf
  mm[1] = __longlong_as_double(0xfff0000000000000ll);   // -inf
}

__global__ __launch_bounds__(256) void minmax_kernel(const double* __restrict__ x, long long n, double* mm) {
  __shared__ double smin[4], smax[4];
  double lo = __longlong_as_double(0x7ff0000000000000ll), hi = -lo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const double v = x[i];
    lo = fmin(lo, v);
    hi = fmax(hi, v);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_down(lo, o, 64));
    hi = fmax(hi, __shfl_down(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
    hi = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    atomic_min_f64(mm, lo);
    atomic_max_f64(mm + 1, hi);
  }
}

__global__ __launch_bounds__(256) void histogram_kernel(const double* __restrict__ x, long long n,
                                                        const double* __restrict__ edges, int nbins,
                                                        unsigned long long* __restrict__ counts) {
  extern __shared__ unsigned int local[];
  for (int k = threadIdx.x; k < nbins; k += blockDim.x) local[k] = 0u;
  __syncthreads();
  const double first = edges[0], last = edges[nbins];
  const double denom = last - first;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const double v = x[i];
    if (!(v >= first) || !(v <= last)) continue;
    const double f = ((v - first) / denom) * (double)nbins;
    int idx = (int)f;
    if (idx == nbins) idx -= 1;
    if (v < edges[idx]) idx -= 1;
    if (v >= edges[idx + 1] && idx != nbins - 1) idx += 1;
    atomicAdd(&local[idx], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nbins; k += blockDim.x)
    if (local[k]) atomicAdd(&counts[k], (unsigned long long)local[k]);
}

}  // namespace

extern "C" int clx_minmax_f64(const double* x, long long n, double* minmax, clx_stream stream) {
  CLX_REQUIRE(x && minmax && n > 0, "clx_minmax_f64: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  minmax_init<<<1, 1, 0, st>>>(minmax);
  minmax_kernel<<<grid_for(n, 256), 256, 0, st>>>(x, n, minmax);
  CLX_CHECK_LAUNCH("clx_minmax_f64");
  return CLX_OK;
}

extern "C" int clx_histogram_f64(const double* x, long long n, const double* edges, int nbins,
                                 long long* counts, clx_stream stream) {
  CLX_REQUIRE(x && edges && counts && n > 0, "clx_histogram_f64: bad arguments");
  CLX_REQUIRE(nbins > 0 && nbins <= 8192, "clx_histogram_f64: nbins must be in 1..8192");
  histogram_kernel<<<grid_for(n, 256), 256, (size_t)nbins * sizeof(unsigned int), (hipStream_t)stream>>>(
      x, n, edges, nbins, (unsigned long long*)counts);
  CLX_CHECK_LAUNCH("clx_histogram_f64");
  return CLX_OK;
}
