// Histogram primitives behind skimage.filters.threshold_otsu as used on the
// std channel (cellulus/detect.py:88-91): min/max of an f64 image and
// numpy.histogram's uniform-bin index computation, reproduced step by step
// (scale, truncate, then the +-1 corrections against the linspace edges) so the
// counts are bit-exact.  Compiled with the default contraction but the index
// expression has no multiply-add pair to fuse.
#include "clx_common.h"
#include <stdlib.h>

namespace {

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  static const int cap = getenv("CLX_OTSU_GRID") ? atoi(getenv("CLX_OTSU_GRID")) : 1024;     // (sweeps)
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int MM_BLOCKS = 512;      // == (CLX_MINMAX_DOUBLES - 2) / 2

// Block partials, then one small block over them: no two blocks meet on an address.  (Until round 4 every block ended
// with two float64 atomics on the SAME two words: they are served one after the other at the memory side, ~12 ns each —
// the kernel's time was proportional to its grid, 32 / 61 / 104 / 196 us for 512 / 2048 / 4096 / 8192 blocks over a
// 134-MB image that streams in 25 us.)  mm: [0] min, [1] max, [2 + 2 b], [3 + 2 b] the partials of block b.
template <typename T>
__device__ __forceinline__ void minmax_block_out(T lo, T hi, double* mm) {
  __shared__ double smin[4], smax[4];
  for (int o = 32; o > 0; o >>= 1) {
    // (the shuffles are issued with all lanes active, BEFORE the comparison: inside a conditional expression they would run
    //  under a partial EXEC mask and read zeros from the inactive lanes)
    const T l2 = __shfl_down(lo, o, 64), h2 = __shfl_down(hi, o, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = (double)lo; smax[threadIdx.x >> 6] = (double)hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mm[2 + 2 * blockIdx.x] = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
    mm[3 + 2 * blockIdx.x] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
  }
}

__global__ __launch_bounds__(256) void minmax_final(double* mm, int nblocks) {
  __shared__ double smin[4], smax[4];
  double lo = __longlong_as_double(0x7ff0000000000000ll), hi = -lo;
  for (int b = threadIdx.x; b < nblocks; b += 256) { lo = fmin(lo, mm[2 + 2 * b]); hi = fmax(hi, mm[3 + 2 * b]); }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_down(lo, o, 64));
    hi = fmax(hi, __shfl_down(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mm[0] = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
    mm[1] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
  }
}

// float32 input (the network's std channel handed over in device memory, cellulus_amd/infer.py::fused_stages): the
// float64 values the staged path reads back from zarr are these floats widened, so min / max / bin of the widened
// value are the staged path's bits.  16 bytes = four floats per lane.
// A block reads ONE contiguous range, eight 4-KB pieces of it in flight per round.
__global__ __launch_bounds__(256) void minmax_f32_kernel(const float* __restrict__ x, long long n, double* mm) {
  float lo = __int_as_float(0x7f800000), hi = -lo;
  const long long n4 = n >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const long long per_block = (n4 + gridDim.x - 1) / gridDim.x;
  const long long b0 = (long long)blockIdx.x * per_block, b1 = b0 + per_block < n4 ? b0 + per_block : n4;
  long long i = b0 + threadIdx.x;
  for (; i + 7 * 256 < b1; i += 8 * 256) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x4[i + 256 * u];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo = fminf(lo, v[u][e]); hi = fmaxf(hi, v[u][e]); }
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
  minmax_block_out(lo, hi, mm);
}

// 16-byte loads (two doubles per lane): 8-byte accesses run at 0.54-0.70x the 16-byte rate
__global__ __launch_bounds__(256) void minmax_kernel(const double* __restrict__ x, long long n, double* mm) {
  double lo = __longlong_as_double(0x7ff0000000000000ll), hi = -lo;
  const long long n2 = n >> 1;
  const f64x2* x2 = reinterpret_cast<const f64x2*>(x);
  // a block reads ONE contiguous range (eight 4-KB pieces of it in flight per round): with the grid-strided walk the
  // loads of a thread were 8 MB apart and a 4096^2 image ran at 2.3 TB/s
  const long long per_block = (n2 + gridDim.x - 1) / gridDim.x;
  const long long b0 = (long long)blockIdx.x * per_block, b1 = b0 + per_block < n2 ? b0 + per_block : n2;
  long long i = b0 + threadIdx.x;
  for (; i + 7 * 256 < b1; i += 8 * 256) {
    f64x2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x2[i + 256 * u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      lo = fmin(lo, fmin(v[u][0], v[u][1]));
      hi = fmax(hi, fmax(v[u][0], v[u][1]));
    }
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
  minmax_block_out(lo, hi, mm);
}

// A thread keeps the bin of its previous value and a count: images are smooth (the std channel of a
// cell image is two plateaus), so consecutive values of a thread mostly share a bin and one LDS
// atomic covers the run — with one atomic per value a two-valued image serialises 64 lanes on two
// addresses.
__device__ __forceinline__ void hist_one(double v, double first, double last, double denom, int nbins,
                                         const double* __restrict__ edges, unsigned int* local, int& cur,
                                         unsigned int& run) {
  if (!(v >= first) || !(v <= last)) return;
  // numpy: idx = int((v - first) / denom * nbins), then the two corrections against the edges — i.e. THE bin with
  // edges[idx] <= v < edges[idx + 1] (last bin closed) for any first guess within one bin of it.  The guess here is a
  // multiplication by nbins / denom (`denom` carries that quotient: the division cost a third of the kernel's
  // instructions), a few ulp from numpy's: the same bin after the corrections.
  int idx = (int)((v - first) * denom);
  idx = min(idx, nbins - 1);
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
  const double denom = (double)nbins / (last - first);      // (see hist_one)
  const long long n2 = n >> 1;
  const f64x2* x2 = reinterpret_cast<const f64x2*>(x);
  int cur = 0;
  unsigned int run = 0u;
  // a block takes ONE contiguous range, four 4-KB pieces of it in flight per round (consecutive values of a thread are
  // then 4 KB apart instead of the whole grid's stride: same plateau, same bin, one LDS atomic per run)
  const long long per_block = (n2 + gridDim.x - 1) / gridDim.x;
  const long long b0 = (long long)blockIdx.x * per_block, b1 = b0 + per_block < n2 ? b0 + per_block : n2;
  long long i = b0 + threadIdx.x;
  for (; i + 3 * 256 < b1; i += 4 * 256) {
    f64x2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x2[i + 256 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      hist_one(v[u][0], first, last, denom, nbins, eds, mine, cur, run);
      hist_one(v[u][1], first, last, denom, nbins, eds, mine, cur, run);
    }
  }
  for (; i < b1; i += 256) {
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
  const double denom = (double)nbins / (last - first);      // (see hist_one)
  const long long n4 = n >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  int cur = 0;
  unsigned int run = 0u;
  const long long per_block = (n4 + gridDim.x - 1) / gridDim.x;
  const long long b0 = (long long)blockIdx.x * per_block, b1 = b0 + per_block < n4 ? b0 + per_block : n4;
  long long i = b0 + threadIdx.x;
  for (; i + 3 * 256 < b1; i += 4 * 256) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x4[i + 256 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) hist_one((double)v[u][e], first, last, denom, nbins, eds, mine, cur, run);
  }
  for (; i < b1; i += 256) {
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

inline int mm_grid(long long work) {
  static const int cap = getenv("CLX_MINMAX_GRID") ? atoi(getenv("CLX_MINMAX_GRID")) : MM_BLOCKS;
  long long g = (work + 255) / 256;
  const int c = cap < MM_BLOCKS ? (cap < 1 ? 1 : cap) : MM_BLOCKS;
  return (int)(g > c ? c : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int clx_minmax_f32(const float* x, long long n, double* minmax, clx_stream stream) {
  CLX_REQUIRE(x && minmax && n > 0, "clx_minmax_f32: bad arguments");
  CLX_REQUIRE(((uintptr_t)x & 15) == 0, "clx_minmax_f32: x must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int g = mm_grid(n / 32 + 1);
  CLX_LAUNCH_KIND(CLX_PROF_MINMAX, minmax_f32_kernel, dim3(g), dim3(256), 0, st, x, n, minmax);
  CLX_LAUNCH_KIND(CLX_PROF_MINMAX, minmax_final, dim3(1), dim3(256), 0, st, minmax, g);
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
  const int g = mm_grid(n / 16 + 1);
  CLX_LAUNCH_KIND(CLX_PROF_MINMAX, minmax_kernel, dim3(g), dim3(256), 0, st, x, n, minmax);
  CLX_LAUNCH_KIND(CLX_PROF_MINMAX, minmax_final, dim3(1), dim3(256), 0, st, minmax, g);
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
