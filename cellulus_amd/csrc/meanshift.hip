// Mean-shift clustering of per-pixel embeddings in float64 for gfx950.
// Compiled with -ffp-contract=off: the membership test d^2 <= bw^2 must use
// the same un-fused arithmetic as the reference's KD-tree (sum of squared
// differences, one rounding per operation).
//
// Replaces cellulus/utils/mean_shift.py:6-121 -> sklearn.cluster.MeanShift
// (fit: _mean_shift_single_seed per seed; predict: nearest centre).
#include "clx_common.h"

namespace {

constexpr int PREP_BLOCK = 1024;   // pixels per compaction block

typedef double f64x2 __attribute__((ext_vector_type(2)));

// pass 1: add coordinates in place, count foreground per block.  Each thread owns two
// adjacent pixels so every access is a 16-byte load/store (npix even; odd sizes take the
// scalar tail below).
__global__ __launch_bounds__(256) void ms_prepare_count(double* __restrict__ emb,
                                                        const double* __restrict__ sd,
                                                        double thr, int ND, int Y, int X,
                                                        long long npix, int vec,
                                                        int* __restrict__ counts) {
  __shared__ int wsum[4];
  const long long base = (long long)blockIdx.x * PREP_BLOCK;
  int local = 0;
  if (vec) {
    for (int k = 0; k < PREP_BLOCK / 512; ++k) {
      const long long i = base + (long long)(k * 256 + threadIdx.x) * 2;
      if (i < npix) {      // npix is even here, so i + 1 < npix too
        const int x0 = (int)(i % X);
        const long long t = i / X;
        const int y0 = (int)(t % Y);
        const int z0 = (int)(t / Y);
        int x1 = x0 + 1, y1 = y0, z1 = z0;
        if (x1 == X) { x1 = 0; if (++y1 == Y) { y1 = 0; ++z1; } }
        f64x2 v = *reinterpret_cast<f64x2*>(emb + i);
        v[0] += (double)x0; v[1] += (double)x1;
        *reinterpret_cast<f64x2*>(emb + i) = v;
        v = *reinterpret_cast<f64x2*>(emb + npix + i);
        v[0] += (double)y0; v[1] += (double)y1;
        *reinterpret_cast<f64x2*>(emb + npix + i) = v;
        if (ND == 3) {
          v = *reinterpret_cast<f64x2*>(emb + 2 * npix + i);
          v[0] += (double)z0; v[1] += (double)z1;
          *reinterpret_cast<f64x2*>(emb + 2 * npix + i) = v;
        }
        const f64x2 s2 = *reinterpret_cast<const f64x2*>(sd + i);
        local += (s2[0] < thr ? 1 : 0) + (s2[1] < thr ? 1 : 0);
      }
    }
  } else {
    for (int k = 0; k < PREP_BLOCK / 256; ++k) {
      const long long i = base + k * 256 + threadIdx.x;
      if (i < npix) {
        const int x = (int)(i % X);
        const long long t = i / X;
        const int y = (int)(t % Y);
        const int z = (int)(t / Y);
        emb[i] += (double)x;
        emb[npix + i] += (double)y;
        if (ND == 3) emb[2 * npix + i] += (double)z;
        local += (sd[i] < thr) ? 1 : 0;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// pass 2: exclusive scan of the block counts (single block), total -> nfg_out
__global__ __launch_bounds__(1024) void scan_counts(int* __restrict__ counts, int nblocks,
                                                    int* __restrict__ total_out) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (nblocks + 1023) / 1024;
  const int lo = tid * per, hi = min(lo + per, nblocks);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += counts[i];
  part[tid] = s;
  __syncthreads();
  // Hillis-Steele inclusive scan over 1024 partials
  for (int o = 1; o < 1024; o <<= 1) {
    int v = (tid >= o) ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = (tid == 0) ? 0 : part[tid - 1];
  for (int i = lo; i < hi; ++i) {
    const int c = counts[i];
    counts[i] = run;
    run += c;
  }
  if (tid == 1023 && total_out) *total_out = part[1023];
}

// pass 3: stable (raster-order) compaction of the foreground pixels (two adjacent pixels per
// thread when `vec`, so the std read is a 16-byte load)
__global__ __launch_bounds__(256) void ms_prepare_scatter(const double* __restrict__ emb,
                                                          const double* __restrict__ sd,
                                                          double thr, int ND, long long npix, int vec,
                                                          const int* __restrict__ offsets,
                                                          double* __restrict__ Xout,
                                                          int* __restrict__ index) {
  __shared__ int wcount[4];
  __shared__ int running;
  if (threadIdx.x == 0) running = offsets[blockIdx.x];
  __syncthreads();
  const long long base = (long long)blockIdx.x * PREP_BLOCK;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned long long lower = (1ull << lane) - 1ull;
  const int per = vec ? 2 : 1;
  for (int k = 0; k < PREP_BLOCK / (256 * per); ++k) {
    const long long i = base + (long long)(k * 256 + threadIdx.x) * per;
    bool fg0 = false, fg1 = false;
    if (vec) {
      if (i < npix) {
        const f64x2 s2 = *reinterpret_cast<const f64x2*>(sd + i);
        fg0 = s2[0] < thr;
        fg1 = s2[1] < thr;
      }
    } else {
      fg0 = (i < npix) && (sd[i] < thr);
    }
    const unsigned long long b0 = __ballot(fg0), b1 = __ballot(fg1);
    const int before = __popcll(b0 & lower) + __popcll(b1 & lower);
    if (lane == 0) wcount[wid] = __popcll(b0) + __popcll(b1);
    __syncthreads();
    int woff = running;
    for (int w = 0; w < wid; ++w) woff += wcount[w];
    int pos = woff + before;
    if (fg0) {
      for (int c = 0; c < ND; ++c) Xout[(long long)pos * ND + c] = emb[(long long)c * npix + i];
      index[pos] = (int)i;
      ++pos;
    }
    if (fg1) {
      for (int c = 0; c < ND; ++c) Xout[(long long)pos * ND + c] = emb[(long long)c * npix + i + 1];
      index[pos] = (int)(i + 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) running += wcount[0] + wcount[1] + wcount[2] + wcount[3];
    __syncthreads();
  }
}

// One wavefront per seed: sklearn _mean_shift_single_seed.
//   loop: members = fit points with |x - mean|^2 <= bw^2 ; if none: stop
//         new = mean(members); if |new - mean| <= 1e-3 bw or it == max_iter: stop
template <int ND>
__global__ __launch_bounds__(256) void ms_iterate_kernel(const double* __restrict__ fit, int nfit,
                                                         const double* __restrict__ seeds, int nseeds,
                                                         double bw, int max_iter,
                                                         double* __restrict__ centers,
                                                         int* __restrict__ counts,
                                                         int* __restrict__ iters) {
  const int lane = threadIdx.x & 63;
  const int seed = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seed >= nseeds) return;
  const double bw2 = bw * bw;
  const double stop = 1e-3 * bw;
  double mean[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) mean[c] = seeds[(long long)seed * ND + c];
  int completed = 0, members = 0;
  while (true) {
    double sum[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) sum[c] = 0.0;
    int cnt = 0;
    for (int j = lane; j < nfit; j += 64) {
      double x[ND], d2 = 0.0;
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        x[c] = fit[(long long)j * ND + c];
        const double df = x[c] - mean[c];
        d2 += df * df;
      }
      if (d2 <= bw2) {
        ++cnt;
#pragma unroll
        for (int c = 0; c < ND; ++c) sum[c] += x[c];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      cnt += __shfl_xor(cnt, o, 64);
#pragma unroll
      for (int c = 0; c < ND; ++c) sum[c] += __shfl_xor(sum[c], o, 64);
    }
    members = cnt;
    if (cnt == 0) break;
    double shift2 = 0.0;
#pragma unroll
    for (int c = 0; c < ND; ++c) {
      const double nm = sum[c] / (double)cnt;
      const double df = nm - mean[c];
      shift2 += df * df;
      mean[c] = nm;
    }
    if (sqrt(shift2) <= stop || completed == max_iter) break;
    ++completed;
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < ND; ++c) centers[(long long)seed * ND + c] = mean[c];
    counts[seed] = members;
    iters[seed] = completed;
  }
}

// Same iteration with the fit points bucketed into cells of edge h >= bandwidth (sorted by
// cell id, x fastest): the members of a query lie in the 3^ND cells around it, and the 3
// x-adjacent cells of one (z, y) row are one contiguous run of the sorted array.  Cuts the
// pair evaluations from nseeds*nfit to nseeds*(points in 3^ND cells) per iteration.
template <int ND>
__global__ __launch_bounds__(256) void ms_iterate_grid_kernel(
    const double* __restrict__ fit, const int* __restrict__ cell_start, double ox, double oy,
    double oz, double inv_h, int nx, int ny, int nz, const double* __restrict__ seeds, int nseeds,
    double bw, int max_iter, double* __restrict__ centers, int* __restrict__ counts,
    int* __restrict__ iters) {
  const int lane = threadIdx.x & 63;
  const int seed = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seed >= nseeds) return;
  const double bw2 = bw * bw;
  const double stop = 1e-3 * bw;
  double mean[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) mean[c] = seeds[(long long)seed * ND + c];
  int completed = 0, members = 0;
  while (true) {
    double sum[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) sum[c] = 0.0;
    int cnt = 0;
    const int cx = (int)floor((mean[0] - ox) * inv_h);
    const int cy = (int)floor((mean[1] - oy) * inv_h);
    const int cz = (ND == 3) ? (int)floor((mean[2] - oz) * inv_h) : 0;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, nx - 1);
    if (x0 <= x1) {
      for (int zz = (ND == 3 ? cz - 1 : 0); zz <= (ND == 3 ? cz + 1 : 0); ++zz) {
        if (zz < 0 || zz >= nz) continue;
        for (int yy = cy - 1; yy <= cy + 1; ++yy) {
          if (yy < 0 || yy >= ny) continue;
          const long long row = ((long long)zz * ny + yy) * nx;
          const int lo = cell_start[row + x0], hi = cell_start[row + x1 + 1];
          for (int j = lo + lane; j < hi; j += 64) {
            double x[ND], d2 = 0.0;
#pragma unroll
            for (int c = 0; c < ND; ++c) {
              x[c] = fit[(long long)j * ND + c];
              const double df = x[c] - mean[c];
              d2 += df * df;
            }
            if (d2 <= bw2) {
              ++cnt;
#pragma unroll
              for (int c = 0; c < ND; ++c) sum[c] += x[c];
            }
          }
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      cnt += __shfl_xor(cnt, o, 64);
#pragma unroll
      for (int c = 0; c < ND; ++c) sum[c] += __shfl_xor(sum[c], o, 64);
    }
    members = cnt;
    if (cnt == 0) break;
    double shift2 = 0.0;
#pragma unroll
    for (int c = 0; c < ND; ++c) {
      const double nm = sum[c] / (double)cnt;
      const double df = nm - mean[c];
      shift2 += df * df;
      mean[c] = nm;
    }
    if (sqrt(shift2) <= stop || completed == max_iter) break;
    ++completed;
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < ND; ++c) centers[(long long)seed * ND + c] = mean[c];
    counts[seed] = members;
    iters[seed] = completed;
  }
}

// nearest centre (first minimum) for every foreground pixel; centres staged in LDS
template <int ND>
__global__ __launch_bounds__(256) void ms_assign_kernel(const double* __restrict__ X,
                                                        const int* __restrict__ index, int nfg,
                                                        const double* __restrict__ centers,
                                                        int ncenters, int* __restrict__ labels) {
  constexpr int CHUNK = 1024;
  __shared__ double cs[CHUNK * ND];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double x[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) x[c] = (i < nfg) ? X[(long long)i * ND + c] : 0.0;
  double best = 0.0;
  int arg = -1;
  for (int c0 = 0; c0 < ncenters; c0 += CHUNK) {
    const int nc = min(CHUNK, ncenters - c0);
    __syncthreads();
    for (int k = threadIdx.x; k < nc * ND; k += blockDim.x) cs[k] = centers[(long long)c0 * ND + k];
    __syncthreads();
    for (int k = 0; k < nc; ++k) {
      double d2 = 0.0;
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        const double df = x[c] - cs[k * ND + c];
        d2 += df * df;
      }
      if (arg < 0 || d2 < best) { best = d2; arg = c0 + k; }
    }
  }
  if (i < nfg) labels[index[i]] = arg + 1;
}

}  // namespace

extern "C" size_t clx_ms_prepare_workspace(long long npix) {
  const long long nblocks = (npix + PREP_BLOCK - 1) / PREP_BLOCK;
  return (size_t)(nblocks + 1) * sizeof(int);
}

extern "C" int clx_ms_prepare(double* emb, const double* std, double threshold, int ND,
                              int Z, int Y, int X, double* Xout, int* index, int* nfg_out,
                              void* workspace, clx_stream stream) {
  CLX_REQUIRE(emb && std && Xout && index && nfg_out && workspace, "clx_ms_prepare: null pointer");
  CLX_REQUIRE((ND == 2 || ND == 3) && Z > 0 && Y > 0 && X > 0, "clx_ms_prepare: bad extents");
  CLX_REQUIRE(ND == 3 || Z == 1, "clx_ms_prepare: Z must be 1 for 2-D data");
  const long long npix = (long long)Z * Y * X;
  CLX_REQUIRE(npix < (1ll << 31), "clx_ms_prepare: too many pixels");
  const int nblocks = (int)((npix + PREP_BLOCK - 1) / PREP_BLOCK);
  int* counts = (int*)workspace;
  hipStream_t st = (hipStream_t)stream;
  const int vec = (npix % 2 == 0) && (((uintptr_t)emb | (uintptr_t)std) & 15) == 0 ? 1 : 0;
  ms_prepare_count<<<nblocks, 256, 0, st>>>(emb, std, threshold, ND, Y, X, npix, vec, counts);
  scan_counts<<<1, 1024, 0, st>>>(counts, nblocks, nfg_out);
  ms_prepare_scatter<<<nblocks, 256, 0, st>>>(emb, std, threshold, ND, npix, vec, counts, Xout, index);
  CLX_CHECK_LAUNCH("clx_ms_prepare");
  return CLX_OK;
}

extern "C" int clx_ms_iterate(const double* fit, int nfit, const double* seeds, int nseeds,
                              int ND, double bandwidth, int max_iter, double* centers,
                              int* counts, int* iters, clx_stream stream) {
  CLX_REQUIRE(fit && seeds && centers && counts && iters, "clx_ms_iterate: null pointer");
  CLX_REQUIRE((ND == 2 || ND == 3) && nfit >= 0 && nseeds >= 0 && max_iter >= 0,
              "clx_ms_iterate: bad extents");
  CLX_REQUIRE(bandwidth > 0.0, "clx_ms_iterate: bandwidth must be positive");
  if (nseeds == 0) return CLX_OK;
  const int grid = (nseeds + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  if (ND == 2)
    ms_iterate_kernel<2><<<grid, 256, 0, st>>>(fit, nfit, seeds, nseeds, bandwidth, max_iter, centers, counts, iters);
  else
    ms_iterate_kernel<3><<<grid, 256, 0, st>>>(fit, nfit, seeds, nseeds, bandwidth, max_iter, centers, counts, iters);
  CLX_CHECK_LAUNCH("clx_ms_iterate");
  return CLX_OK;
}

extern "C" int clx_ms_iterate_grid(const double* fit_sorted, int nfit, const int* cell_start,
                                   const double* origin, double cell, int nx, int ny, int nz,
                                   const double* seeds, int nseeds, int ND, double bandwidth,
                                   int max_iter, double* centers, int* counts, int* iters,
                                   clx_stream stream) {
  CLX_REQUIRE(fit_sorted && cell_start && origin && seeds && centers && counts && iters,
              "clx_ms_iterate_grid: null pointer");
  CLX_REQUIRE((ND == 2 || ND == 3) && nfit >= 0 && nseeds >= 0 && max_iter >= 0,
              "clx_ms_iterate_grid: bad extents");
  CLX_REQUIRE(bandwidth > 0.0 && cell >= bandwidth, "clx_ms_iterate_grid: cell edge must be >= bandwidth > 0");
  CLX_REQUIRE(nx > 0 && ny > 0 && nz > 0 && (ND == 3 || nz == 1), "clx_ms_iterate_grid: bad grid");
  if (nseeds == 0) return CLX_OK;
  const int grid = (nseeds + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  const double inv = 1.0 / cell;
  if (ND == 2)
    ms_iterate_grid_kernel<2><<<grid, 256, 0, st>>>(fit_sorted, cell_start, origin[0], origin[1], 0.0, inv,
                                                     nx, ny, nz, seeds, nseeds, bandwidth, max_iter,
                                                     centers, counts, iters);
  else
    ms_iterate_grid_kernel<3><<<grid, 256, 0, st>>>(fit_sorted, cell_start, origin[0], origin[1], origin[2],
                                                     inv, nx, ny, nz, seeds, nseeds, bandwidth, max_iter,
                                                     centers, counts, iters);
  CLX_CHECK_LAUNCH("clx_ms_iterate_grid");
  return CLX_OK;
}

extern "C" int clx_ms_assign(const double* X, const int* index, int nfg, const double* centers,
                             int ncenters, int ND, int* labels, clx_stream stream) {
  CLX_REQUIRE(X && index && centers && labels, "clx_ms_assign: null pointer");
  CLX_REQUIRE((ND == 2 || ND == 3) && nfg >= 0 && ncenters > 0, "clx_ms_assign: bad extents");
  if (nfg == 0) return CLX_OK;
  const int grid = (nfg + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  if (ND == 2)
    ms_assign_kernel<2><<<grid, 256, 0, st>>>(X, index, nfg, centers, ncenters, labels);
  else
    ms_assign_kernel<3><<<grid, 256, 0, st>>>(X, index, nfg, centers, ncenters, labels);
  CLX_CHECK_LAUNCH("clx_ms_assign");
  return CLX_OK;
}
