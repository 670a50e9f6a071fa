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

// gfx950 has float64 min / max atomics in hardware (global_atomic_min_f64 / max_f64): fire-and-forget, where a
// compare-and-swap loop made the 2048 blocks of a launch queue up on two addresses (15 of 57 us at 4096^2)
__device__ __forceinline__ void atomic_min_f64(double* addr, double v) {
  __builtin_amdgcn_global_atomic_fmin_f64(addr, v);
}
__device__ __forceinline__ void atomic_max_f64(double* addr, double v) {
  __builtin_amdgcn_global_atomic_fmax_f64(addr, v);
}

__global__ void minmax_init(double* mm) {
  mm[0] = __longlong_as_double(0x7ff0000000000000ll);   // +inf
  mm[1] = __longlong_as_double(0xfff0000000000000ll);   // -inf
}

typedef double f64x2 __attribute__((ext_vector_type(2)));

// float32 input (the network's std channel handed over in device memory, cellulus_amd/infer.py::fused_stages): the
// float64 values the staged path reads back from zarr are these floats widened, so min / max / bin of the widened
// value are the staged path's bits.  16 bytes = four floats per lane.
__global__ __launch_bounds__(256) void minmax_f32_kernel(const float* __restrict__ x, long long n, double* mm) {
  __shared__ double smin[4], smax[4];
  float lo = __int_as_float(0x7f800000), hi = -lo;
  const long long n4 = n >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const long long per_block = (n4 + gridDim.x - 1) / gridDim.x;
  const long long b0 = (long long)blockIdx.x * per_block, b1 = b0 + per_block < n4 ? b0 + per_block : n4;
  long long i = b0 + threadIdx.x;
  for (; i + 3 * 256 < b1; i += 4 * 256) {
    const f32x4 a = x4[i], b = x4[i + 256], c = x4[i + 512], d = x4[i + 768];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo = fminf(fminf(lo, a[e]), fminf(b[e], fminf(c[e], d[e])));
      hi = fmaxf(fmaxf(hi, a[e]), fmaxf(b[e], fmaxf(c[e], d[e])));
    }
  }
  for (; i < b1; i += 256) {
    const f32x4 a = x4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) { lo = fminf(lo, a[e]); hi = fmaxf(hi, a[e]); }
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    lo = fminf(lo, x[(n4 << 2) + threadIdx.x]);
    hi = fmaxf(hi, x[(n4 << 2) + threadIdx.x]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_down(lo, o, 64));
    hi = fmaxf(hi, __shfl_down(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = (double)lo; smax[threadIdx.x >> 6] = (double)hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomic_min_f64(mm, fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3])));
    atomic_max_f64(mm + 1, fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3])));
  }
}

// 16-byte loads (two doubles per lane): 8-byte accesses run at 0.54-0.70x the 16-byte rate
__global__ __launch_bounds__(256) void minmax_kernel(const double* __restrict__ x, long long n, double* mm) {
  __shared__ double smin[4], smax[4];
  double lo = __longlong_as_double(0x7ff0000000000000ll), hi = -lo;
  const long long n2 = n >> 1;
  const f64x2* x2 = reinterpret_cast<const f64x2*>(x);
  // a block reads ONE contiguous range (four 4-KB pieces of it in flight per round): with the grid-strided walk the four
  // loads of a thread were 8 MB apart and a 4096^2 image ran at 2.3 TB/s
  const long long per_block = (n2 + gridDim.x - 1) / gridDim.x;
  const long long b0 = (long long)blockIdx.x * per_block, b1 = b0 + per_block < n2 ? b0 + per_block : n2;
  long long i = b0 + threadIdx.x;
  for (; i + 3 * 256 < b1; i += 4 * 256) {             // 4 independent 16-byte loads in flight
    const f64x2 a = x2[i], b = x2[i + 256], c = x2[i + 512], d = x2[i + 768];
    lo = fmin(fmin(fmin(lo, a[0]), fmin(a[1], b[0])), fmin(fmin(b[1], c[0]), fmin(fmin(c[1], d[0]), d[1])));
    hi = fmax(fmax(fmax(hi, a[0]), fmax(a[1], b[0])), fmax(fmax(b[1], c[0]), fmax(fmax(c[1], d[0]), d[1])));
  }
  for (; i < b1; i += 256) {
    const f64x2 a = x2[i];
    lo = fmin(lo, fmin(a[0], a[1]));
    hi = fmax(hi, fmax(a[0], a[1]));
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    lo = fmin(lo, x[n - 1]);
    hi = fmax(hi, x[n - 1]);
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

// A thread keeps the bin of its previous value and a count: images are smooth (the std channel of a
// cell image is two plateaus), so consecutive values of a thread mostly share a bin and one LDS
// atomic covers the run — with one atomic per value a two-valued image serialises 64 lanes on two
// addresses.
__device__ __forceinline__ void hist_one(double v, double first, double last, double denom, int nbins,
                                         const double* __restrict__ edges, unsigned int* local, int& cur,
                                         unsigned int& run) {
  if (!(v >= first) || !(v <= last)) return;
  const double f = ((v - first) / denom) * (double)nbins;
  int idx = (int)f;
  if (idx == nbins) idx -= 1;
  if (v < edges[idx]) idx -= 1;
  if (v >= edges[idx + 1] && idx != nbins - 1) idx += 1;
  if (idx == cur) {
    ++run;
  } else {
    if (run) atomicAdd(&local[cur], run);
    cur = idx;
    run = 1u;
  }
}

// per-WAVE private histograms in LDS (4 copies) cut the LDS-atomic contention; the bin
// edges are staged in LDS too (the +-1 corrections read them for every element)
__global__ __launch_bounds__(256) void histogram_kernel(const double* __restrict__ x, long long n,
                                                        const double* __restrict__ edges, int nbins,
                                                        unsigned long long* __restrict__ counts) {
  extern __shared__ unsigned char hist_smem[];
  double* eds = reinterpret_cast<double*>(hist_smem);                       // [nbins + 1]
  unsigned int* local = reinterpret_cast<unsigned int*>(eds + nbins + 1);   // [4][nbins]
  for (int k = threadIdx.x; k <= nbins; k += blockDim.x) eds[k] = edges[k];
  for (int k = threadIdx.x; k < 4 * nbins; k += blockDim.x) local[k] = 0u;
  __syncthreads();
  unsigned int* mine = local + (threadIdx.x >> 6) * nbins;
  const double first = eds[0], last = eds[nbins];
  const double denom = last - first;
  const long long n2 = n >> 1;
  const f64x2* x2 = reinterpret_cast<const f64x2*>(x);
  int cur = 0;
  unsigned int run = 0u;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2;
       i += (long long)gridDim.x * blockDim.x) {
    const f64x2 v = x2[i];
    hist_one(v[0], first, last, denom, nbins, eds, mine, cur, run);
    hist_one(v[1], first, last, denom, nbins, eds, mine, cur, run);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) hist_one(x[n - 1], first, last, denom, nbins, eds, mine, cur, run);
  if (run) atomicAdd(&mine[cur], run);
  __syncthreads();
  for (int k = threadIdx.x; k < nbins; k += blockDim.x) {
    const unsigned int c = local[k] + local[nbins + k] + local[2 * nbins + k] + local[3 * nbins + k];
    if (c) atomicAdd(&counts[k], (unsigned long long)c);
  }
}

// the same histogram of float32 values widened in registers (four per 16-byte load)
__global__ __launch_bounds__(256) void histogram_f32_kernel(const float* __restrict__ x, long long n,
                                                            const double* __restrict__ edges, int nbins,
                                                            unsigned long long* __restrict__ counts) {
  extern __shared__ unsigned char hist_smem[];
  double* eds = reinterpret_cast<double*>(hist_smem);                       // [nbins + 1]
  unsigned int* local = reinterpret_cast<unsigned int*>(eds + nbins + 1);   // [4][nbins]
  for (int k = threadIdx.x; k <= nbins; k += blockDim.x) eds[k] = edges[k];
  for (int k = threadIdx.x; k < 4 * nbins; k += blockDim.x) local[k] = 0u;
  __syncthreads();
  unsigned int* mine = local + (threadIdx.x >> 6) * nbins;
  const double first = eds[0], last = eds[nbins];
  const double denom = last - first;
  const long long n4 = n >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  int cur = 0;
  unsigned int run = 0u;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = x4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) hist_one((double)v[e], first, last, denom, nbins, eds, mine, cur, run);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = n4 << 2; i < n; ++i) hist_one((double)x[i], first, last, denom, nbins, eds, mine, cur, run);
  if (run) atomicAdd(&mine[cur], run);
  __syncthreads();
  for (int k = threadIdx.x; k < nbins; k += blockDim.x) {
    const unsigned int c = local[k] + local[nbins + k] + local[2 * nbins + k] + local[3 * nbins + k];
    if (c) atomicAdd(&counts[k], (unsigned long long)c);
  }
}

}  // namespace

extern "C" int clx_minmax_f32(const float* x, long long n, double* minmax, clx_stream stream) {
  CLX_REQUIRE(x && minmax && n > 0, "clx_minmax_f32: bad arguments");
  CLX_REQUIRE(((uintptr_t)x & 15) == 0, "clx_minmax_f32: x must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  CLX_LAUNCH_KIND(CLX_PROF_MINMAX, minmax_init, dim3(1), dim3(1), 0, st, minmax);
  CLX_LAUNCH_KIND(CLX_PROF_MINMAX, minmax_f32_kernel, dim3(grid_for(n / 16 + 1, 256)), dim3(256), 0, st, x, n, minmax);
  CLX_CHECK_LAUNCH("clx_minmax_f32");
  return CLX_OK;
}

extern "C" int clx_histogram_f32(const float* x, long long n, const double* edges, int nbins,
                                 long long* counts, clx_stream stream) {
  CLX_REQUIRE(x && edges && counts && n > 0, "clx_histogram_f32: bad arguments");
  CLX_REQUIRE(nbins > 0 && nbins <= 2048, "clx_histogram_f32: nbins must be in 1..2048");
  CLX_REQUIRE(((uintptr_t)x & 15) == 0, "clx_histogram_f32: x must be 16-byte aligned");
  const size_t lds = (size_t)(nbins + 1) * sizeof(double) + (size_t)4 * nbins * sizeof(unsigned int);
  CLX_LAUNCH_KIND(CLX_PROF_HISTOGRAM, histogram_f32_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), lds,
                  (hipStream_t)stream, x, n, edges, nbins, (unsigned long long*)counts);
  CLX_CHECK_LAUNCH("clx_histogram_f32");
  return CLX_OK;
}

extern "C" int clx_minmax_f64(const double* x, long long n, double* minmax, clx_stream stream) {
  CLX_REQUIRE(x && minmax && n > 0, "clx_minmax_f64: bad arguments");
  CLX_REQUIRE(((uintptr_t)x & 15) == 0, "clx_minmax_f64: x must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  CLX_LAUNCH_KIND(CLX_PROF_MINMAX, minmax_init, dim3(1), dim3(1), 0, st, minmax);
  CLX_LAUNCH_KIND(CLX_PROF_MINMAX, minmax_kernel, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, st, x, n, minmax);
  CLX_CHECK_LAUNCH("clx_minmax_f64");
  return CLX_OK;
}

extern "C" int clx_histogram_f64(const double* x, long long n, const double* edges, int nbins,
                                 long long* counts, clx_stream stream) {
  CLX_REQUIRE(x && edges && counts && n > 0, "clx_histogram_f64: bad arguments");
  CLX_REQUIRE(nbins > 0 && nbins <= 2048, "clx_histogram_f64: nbins must be in 1..2048");
  CLX_REQUIRE(((uintptr_t)x & 15) == 0, "clx_histogram_f64: x must be 16-byte aligned");
  const size_t lds = (size_t)(nbins + 1) * sizeof(double) + (size_t)4 * nbins * sizeof(unsigned int);
  CLX_LAUNCH_KIND(CLX_PROF_HISTOGRAM, histogram_kernel, dim3(grid_for(n / 2 + 1, 256)), dim3(256), lds, (hipStream_t)stream, 
      x, n, edges, nbins, (unsigned long long*)counts);
  CLX_CHECK_LAUNCH("clx_histogram_f64");
  return CLX_OK;
}
