// Greedy seed-and-grow clustering of embeddings (clustering = "greedy"):
// cellulus/utils/greedy_cluster.py:46-120 (2-D, float32) and :176-253 (3-D, float64).
//
// The reference loops on the host with two .item() synchronisations and ~10 tensor
// ops per seed.  Here the whole loop runs inside ONE persistent workgroup (1024
// threads): per seed an arg-max reduction (first maximum), one pass that evaluates
// the Gaussian proposal exp(-|e - c|^2 / (2 bw^2)) > 0.5 and counts it, one pass that
// applies it — no host round trip until the loop ends.  The point set (tens of
// thousands of foreground pixels) is L2-resident, so the kernel is latency-, not
// bandwidth-bound; arithmetic follows the reference operation by operation in its
// dtype so that borderline pixels fall on the same side.
#include "clx_common.h"

namespace {

template <typename T>
__device__ __forceinline__ T gexp(T x);
template <>
__device__ __forceinline__ float gexp<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double gexp<double>(double x) { return exp(x); }

template <typename T, int ND>
__global__ __launch_bounds__(1024) void greedy_kernel(const T* __restrict__ emb, const T* __restrict__ seedmap,
                                                      int n, T two_bw2, int min_object_size,
                                                      double seed_thresh, int min_unclustered_sum,
                                                      unsigned char* __restrict__ unclustered,
                                                      unsigned char* __restrict__ proposal,
                                                      int* __restrict__ instance, int* __restrict__ result) {
  __shared__ T sval[16];
  __shared__ int sidx[16];
  __shared__ int scnt[2][16];
  __shared__ int bseed;
  __shared__ T bscore;
  __shared__ int bprop, bunc;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < n; i += 1024) { unclustered[i] = 1; instance[i] = 0; }
  __syncthreads();
  int remaining = n, count = 1, iterations = 0;
  while (remaining > min_unclustered_sum) {
    // ---- arg-max of seedmap * unclustered (first maximum)
    T best = (T)0;
    int bi = 0x7fffffff;
    bool have = false;
    for (int i = tid; i < n; i += 1024) {
      const T s = seedmap[i] * (T)unclustered[i];
      if (!have || s > best) { best = s; bi = i; have = true; }
    }
    if (!have) { best = (T)0; bi = 0x7fffffff; }
    for (int o = 32; o > 0; o >>= 1) {
      const T ov = __shfl_down(best, o, 64);
      const int oi = __shfl_down(bi, o, 64);
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
    }
    if (lane == 0) { sval[wid] = best; sidx[wid] = bi; }
    __syncthreads();
    if (tid == 0) {
      T v = sval[0];
      int ix = sidx[0];
      for (int k = 1; k < 16; ++k) {
        const T ov = sval[k];
        const int oi = sidx[k];
        if (oi != 0x7fffffff && (ix == 0x7fffffff || ov > v || (ov == v && oi < ix))) { v = ov; ix = oi; }
      }
      bseed = ix;
      bscore = v;
    }
    __syncthreads();
    const int seed = bseed;
    if ((double)bscore < seed_thresh) break;
    ++iterations;
    T c[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) c[d] = emb[(long long)d * n + seed];
    // ---- proposal = exp(-sum((e - c)^2 / (2 bw^2))) > 0.5 ; counts with the seed already cleared
    int np = 0, nu = 0;
    for (int i = tid; i < n; i += 1024) {
      T acc = (T)0;
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        const T df = emb[(long long)d * n + i] - c[d];
        acc += (df * df) / two_bw2;
      }
      const bool in = gexp<T>((T)-1 * acc) > (T)0.5;
      proposal[i] = in ? 1 : 0;
      if (in) {
        ++np;
        if (unclustered[i] && i != seed) ++nu;
      }
    }
    for (int o = 32; o > 0; o >>= 1) { np += __shfl_down(np, o, 64); nu += __shfl_down(nu, o, 64); }
    if (lane == 0) { scnt[0][wid] = np; scnt[1][wid] = nu; }
    __syncthreads();
    if (tid == 0) {
      int a = 0, b = 0;
      for (int k = 0; k < 16; ++k) { a += scnt[0][k]; b += scnt[1][k]; }
      bprop = a;
      bunc = b;
    }
    __syncthreads();
    const int nprop = bprop, nunc = bunc;
    const bool accept = nprop > min_object_size && ((float)nunc / (float)nprop) > 0.5f;
    // ---- apply
    int cleared = 0;
    for (int i = tid; i < n; i += 1024) {
      if (i == seed && unclustered[i]) { unclustered[i] = 0; ++cleared; }
      if (proposal[i]) {
        if (accept) instance[i] = count;
        if (unclustered[i]) { unclustered[i] = 0; ++cleared; }
      }
    }
    for (int o = 32; o > 0; o >>= 1) cleared += __shfl_down(cleared, o, 64);
    if (lane == 0) scnt[0][wid] = cleared;
    __syncthreads();
    int tot = 0;
    for (int k = 0; k < 16; ++k) tot += scnt[0][k];
    remaining -= tot;
    if (accept) ++count;
    __syncthreads();
  }
  if (tid == 0) { result[0] = count - 1; result[1] = iterations; }
}

}  // namespace

extern "C" int clx_greedy_cluster(const void* emb, const void* seedmap, int n, int ND, int is_f64,
                                  double bandwidth, int min_object_size, double seed_thresh,
                                  int min_unclustered_sum, void* workspace, int* instance, int* result,
                                  clx_stream stream) {
  CLX_REQUIRE(emb && seedmap && workspace && instance && result, "clx_greedy_cluster: null pointer");
  CLX_REQUIRE(n >= 0 && (ND == 2 || ND == 3), "clx_greedy_cluster: bad extents");
  CLX_REQUIRE(bandwidth > 0.0, "clx_greedy_cluster: bandwidth must be positive");
  unsigned char* unc = (unsigned char*)workspace;
  unsigned char* prop = unc + ((n + 15) / 16) * 16;
  hipStream_t st = (hipStream_t)stream;
  if (is_f64) {
    const double tb = 2.0 * (bandwidth * bandwidth);
    if (ND == 2)
      greedy_kernel<double, 2><<<1, 1024, 0, st>>>((const double*)emb, (const double*)seedmap, n, tb, min_object_size,
                                                   seed_thresh, min_unclustered_sum, unc, prop, instance, result);
    else
      greedy_kernel<double, 3><<<1, 1024, 0, st>>>((const double*)emb, (const double*)seedmap, n, tb, min_object_size,
                                                   seed_thresh, min_unclustered_sum, unc, prop, instance, result);
  } else {
    // the reference divides float32 tensors by the Python double 2*bw**2 -> a float32 scalar
    const float tb = (float)(2.0 * (bandwidth * bandwidth));
    if (ND == 2)
      greedy_kernel<float, 2><<<1, 1024, 0, st>>>((const float*)emb, (const float*)seedmap, n, tb, min_object_size,
                                                  seed_thresh, min_unclustered_sum, unc, prop, instance, result);
    else
      greedy_kernel<float, 3><<<1, 1024, 0, st>>>((const float*)emb, (const float*)seedmap, n, tb, min_object_size,
                                                  seed_thresh, min_unclustered_sum, unc, prop, instance, result);
  }
  CLX_CHECK_LAUNCH("clx_greedy_cluster");
  return CLX_OK;
}
