// Mean-shift clustering of per-pixel embeddings in float64 for gfx950.
// Compiled with -ffp-contract=off: the membership test d^2 <= bw^2 must use
// the same un-fused arithmetic as the reference's KD-tree (sum of squared
// differences, one rounding per operation).
//
// Replaces cellulus/utils/mean_shift.py:6-121 -> sklearn.cluster.MeanShift
// (fit: _mean_shift_single_seed per seed; predict: nearest centre).
#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "clx_common.h"

namespace {

typedef double f64x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// Coordinate add + stable (raster-order) compaction of the foreground pixels [mean_shift.py:15-32,83-90] as TWO
// plain streaming launches — no ticket, no look-back, no persistent blocks, no grid barrier, no state in the workspace.
// The unit is a WAVE-TILE: 1024 consecutive pixels, lane l of the wavefront owning the 16-byte groups g * 64 + l.
//   ms_flags_kernel    reads the std plane once: per wave-tile the foreground flags as the wavefront's ballot words (1 bit
//                      per pixel, in the (group, pixel-of-group) order the scatter pass reads them back in) and the count;
//                      a block takes a contiguous chunk of tiles and also leaves the chunk's count;
//   ms_scatter_kernel  per wave-tile: the points in front of it = the chunks in front of its chunk (<= 2048 values, summed
//                      by the block) + the tiles of its chunk in front of it (summed by the wavefront) — these loads go
//                      out with the tile's embedding loads, all in flight at once —, positions from the ballot words
//                      (popcounts), coordinates added (in place for float64), points and raster indices written.
// (Until the end of round 4 a one-block scan of the 16 K tile counts stood between the two: 7 us + a launch boundary.)
// History: rounds 2-3 did this in ONE launch — persistent blocks taking tiles by an atomic ticket, counts
// published and prefixes obtained by decoupled look-back.  At 4096^2 that form lost 15 us to the ticket word (one word
// serves ~88 returning atomics per microsecond) and 40 us to the look-back (a hop is a round trip to the memory side)
// of its 175 us; batching tickets serialised the look-back (13 ms), super-tiles with a 256-wide look-back halved the
// bytes in flight (219 us).  Three launches (flags, scan, scatter): 147-160 us; two: 140-146 us.
// TIn = double: the reference's float64 arrays, coordinates added IN PLACE (WB).  TIn = float: the network's float32
// output handed over in device memory by infer()'s fused predict -> detect path — the float64 values the staged path
// reads back from the `embeddings` dataset are these floats widened (cellulus/predict.py:104-112), so widening in
// registers gives the same bits; the embedding is not modified (the reference's in-place add lands in a copy that
// detect.py:155-160 throws away) and is read only where a 16-byte group holds a foreground pixel.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<double> { typedef f64x2 type; static constexpr int N = 2; };
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };

template <typename TIn, int G>
__global__ __launch_bounds__(256) void ms_flags_kernel(const TIn* __restrict__ sd, double thr, long long npix, int vec,
                                                       int nwt, int tiles_per_block, unsigned long long* __restrict__ flags,
                                                       int* __restrict__ counts, int* __restrict__ chunk_counts) {
  using V = typename Vec16<TIn>::type;
  constexpr int PXL = Vec16<TIn>::N;
  constexpr int WT = 64 * G * PXL;
  const int lane = threadIdx.x & 63;
  // the tile's G * PXL = 16 ballot words leave as ONE 128-byte store (lane k keeps word k), not as sixteen single-lane stores
  __shared__ int wave_total[4];
  auto emit = [&](int wt, unsigned int b) {
    int total = 0;
    unsigned long long mine = 0ull;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int e = 0; e < PXL; ++e) {
        const unsigned long long w = __ballot((b >> (g * PXL + e)) & 1u);
        if (lane == g * PXL + e) mine = w;
        total += __popcll(w);
      }
    if (lane < G * PXL) flags[(long long)wt * (G * PXL) + lane] = mine;
    if (lane == 0) counts[wt] = total;
    return total;
  };
  // block k takes the CONTIGUOUS chunk of tiles [k * tiles_per_block, ...) — wavefront w every fourth of them — and
  // leaves the chunk's total: the scatter pass adds up the chunks in front of its own (<= 2048 values) and the tiles of
  // its chunk in front of its tile, and no scan launch stands between the two passes
  constexpr int stride = 4;
  const int chunk_end = min(nwt, ((int)blockIdx.x + 1) * tiles_per_block);
  int wt = blockIdx.x * tiles_per_block + (threadIdx.x >> 6);
  int mine_total = 0;
  // whole tiles (wave-uniform): unconditional loads — behind per-lane conditions the compiler's wait-count pass puts a
  // wait behind every load —, and the NEXT tile's loads are issued before this tile's ballots: a wavefront
  // that takes two tiles no longer waits out two load latencies one after the other
  const int nfull = min(chunk_end, vec ? (int)(npix / WT) : 0);
  if (wt < nfull) {
    V cur[G];
#pragma unroll
    for (int g = 0; g < G; ++g) cur[g] = *reinterpret_cast<const V*>(sd + (long long)wt * WT + (long long)(g * 64 + lane) * PXL);
    auto bits = [&](const V (&x)[G]) {
      unsigned int b = 0u;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < PXL; ++e) b |= ((double)x[g][e] < thr ? 1u : 0u) << (g * PXL + e);
      return b;
    };
    while (wt + stride < nfull) {                        // (the last tile is peeled: no conditional prefetch)
      V nxt[G];
#pragma unroll
      for (int g = 0; g < G; ++g)
        nxt[g] = *reinterpret_cast<const V*>(sd + (long long)(wt + stride) * WT + (long long)(g * 64 + lane) * PXL);
      mine_total += emit(wt, bits(cur));
#pragma unroll
      for (int g = 0; g < G; ++g) cur[g] = nxt[g];
      wt += stride;
    }
    mine_total += emit(wt, bits(cur));
    wt += stride;
  }
  // the ragged last tile, tiles of an unaligned image
  for (; wt < chunk_end; wt += stride) {
    const long long base = (long long)wt * WT;
    unsigned int b = 0u;
    if (vec) {
      V sv[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const long long i = base + (long long)(g * 64 + lane) * PXL;
        if (i < npix) sv[g] = *reinterpret_cast<const V*>(sd + i);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const long long i = base + (long long)(g * 64 + lane) * PXL;
        if (i < npix) {
#pragma unroll
          for (int e = 0; e < PXL; ++e) b |= ((double)sv[g][e] < thr ? 1u : 0u) << (g * PXL + e);
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const long long i = base + (long long)(g * 64 + lane) * PXL;
#pragma unroll
        for (int e = 0; e < PXL; ++e)
          if (i + e < npix) b |= ((double)sd[i + e] < thr ? 1u : 0u) << (g * PXL + e);
      }
    }
    mine_total += emit(wt, b);
  }
  if (lane == 0) wave_total[threadIdx.x >> 6] = mine_total;
  __syncthreads();
  if (threadIdx.x == 0) chunk_counts[blockIdx.x] = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

template <int ND, typename TIn, int G, bool WB, int BLOCKS>
__global__ __launch_bounds__(256, BLOCKS) void ms_scatter_kernel(TIn* __restrict__ emb, FastDiv dX, FastDiv dY, int Y, int X,
                                                                 long long npix, int vec, int nwt,
                                                                 const unsigned long long* __restrict__ flags,
                                                                 const int* __restrict__ counts,
                                                                 const int* __restrict__ chunk_counts, int tiles_per_chunk,
                                                                 double* __restrict__ Xout, int* __restrict__ index,
                                                                 int* __restrict__ nfg_out, int* __restrict__ tile_start) {
  __shared__ int part[4];
  using V = typename Vec16<TIn>::type;
  constexpr int PXL = Vec16<TIn>::N;
  constexpr int WT = 64 * G * PXL;
  const int lane = threadIdx.x & 63;
  const unsigned long long lower = (1ull << lane) - 1ull;
  for (int wt = blockIdx.x * 4 + (threadIdx.x >> 6); wt < nwt; wt += gridDim.x * 4) {
    const long long base = (long long)wt * WT;
    // points in front of this tile = the chunks of the flags pass in front of its chunk (summed by the block) + the
    // tiles of its chunk in front of it (summed by the wavefront); the loads go out first, the sums are taken below
    const int chunk = wt / tiles_per_chunk, chunk_first = chunk * tiles_per_chunk;
    // (the chunks by the block, the tiles by the wavefront; every wavefront summing the chunks for itself — no barriers
    //  — measured the same for float64 and 10 % slower for float32 input: 51 against 46 us)
    int before = 0;
    for (int j = threadIdx.x; j < chunk; j += 256) before += chunk_counts[j];
    int inside = 0;
    for (int t = chunk_first + lane; t < wt; t += 64) inside += counts[t];
    const unsigned long long* tf = flags + (long long)wt * G * PXL;
    unsigned long long B[G][PXL];            // wave-uniform: this tile's ballot words
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int e = 0; e < PXL; ++e) B[g][e] = tf[g * PXL + e];
    // the tile's embedding values: every load in flight before the first is used
    V ev[G][ND];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const long long i = base + (long long)(g * 64 + lane) * PXL;
      bool any = WB;
#pragma unroll
      for (int e = 0; e < PXL; ++e) any = any || ((B[g][e] >> lane) & 1ull);
      if (i < npix && any) {
        if (vec) {
#pragma unroll
          for (int c = 0; c < ND; ++c) ev[g][c] = *reinterpret_cast<const V*>(emb + (long long)c * npix + i);
        } else {
#pragma unroll
          for (int c = 0; c < ND; ++c)
#pragma unroll
            for (int e = 0; e < PXL; ++e) ev[g][c][e] = (i + e < npix) ? emb[(long long)c * npix + i + e] : (TIn)0;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      before += __shfl_xor(before, o, 64);
      inside += __shfl_xor(inside, o, 64);
    }
    __syncthreads();                          // (nwt and tiles_per_chunk are multiples of four: the block's four
    if (lane == 0) part[threadIdx.x >> 6] = before;   //  wavefronts run the same trips and sit in one chunk)
    __syncthreads();
    int run = part[0] + part[1] + part[2] + part[3] + inside;     // points before group g of this tile
    if (lane == 0) tile_start[wt] = run;                           // (kept for clx_ms_assign_dense)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const long long i = base + (long long)(g * 64 + lane) * PXL;
      unsigned int gb = 0u;
      int bef = 0, cnt = 0;
#pragma unroll
      for (int e = 0; e < PXL; ++e) {
        gb |= (unsigned int)((B[g][e] >> lane) & 1ull) << e;
        bef += __popcll(B[g][e] & lower);
        cnt += __popcll(B[g][e]);
      }
      if (i < npix && (WB || gb != 0u)) {
        int pos = run + bef;
        const unsigned int t = fdiv((unsigned int)i, dX);
        int cx = (int)((unsigned int)i - t * (unsigned int)X);
        const unsigned int z0u = fdiv(t, dY);
        int cy = (int)(t - z0u * (unsigned int)Y), cz = (int)z0u;
        V (&e4)[ND] = ev[g];
#pragma unroll
        for (int e = 0; e < PXL; ++e) {
          const int co[3] = {cx, cy, cz};
          double val[ND];
#pragma unroll
          for (int c = 0; c < ND; ++c) {
            val[c] = (double)e4[c][e] + (double)co[c];
            if (WB) e4[c][e] = (TIn)val[c];
          }
#ifdef CLX_SCATTER_NOSTORE      // EXPERIMENT: the pass without its point stores
          if (((gb >> e) & 1u) && val[0] == 1e300) {
#else
          if ((gb >> e) & 1u) {
#endif
#pragma unroll
            for (int c = 0; c < ND; ++c) Xout[(long long)pos * ND + c] = val[c];
            if (index) index[pos] = (int)(i + e);
            ++pos;
          }
          if (++cx == X) { cx = 0; if (++cy == Y) { cy = 0; ++cz; } }
        }
        if (WB) {
          if (vec) {
#pragma unroll
            for (int c = 0; c < ND; ++c) *reinterpret_cast<V*>(emb + (long long)c * npix + i) = e4[c];
          } else {
#pragma unroll
            for (int c = 0; c < ND; ++c)
#pragma unroll
              for (int e = 0; e < PXL; ++e)
                if (i + e < npix) emb[(long long)c * npix + i + e] = e4[c][e];
          }
        }
      }
      run += cnt;
    }
    if (wt == nwt - 1 && lane == 0) *nfg_out = run;
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

// ---------------------------------------------------------------------------------------------
// Bucketing of the fit points for ms_iterate_grid_kernel: counting sort by uniform-grid cell
// (x fastest), points of one cell in their original (raster) order — the order a stable sort
// by cell id produces, so the sums of the iteration do not depend on the run.
//   count:   cell id per point + histogram (atomics; counts do not depend on arrival order)
//   scan:    exclusive prefix -> cell_start
//   scatter: arrival-order slot inside the cell (atomic cursor)
//   order:   one wavefront per cell ranks the cell's points by original index (rank sort:
//            n^2 / 64 comparisons per cell, cells hold tens to a few thousand points) and writes
//            the point into its final slot.  Embeddings collapse onto object centres, so a
//            degenerate prediction can put 1e5 - 1e6 points into ONE cell: cells above
//            BUCKET_BIG points are ranked by one THREAD per point instead (bucket_order_big_kernel:
//            the same n^2 comparisons spread over the whole chip rather than one wavefront)
// ---------------------------------------------------------------------------------------------
template <int ND>
__global__ void bucket_count_kernel(const double* __restrict__ fit, int n, double ox, double oy, double oz,
                                    double h, int nx, int ny, int nz, int* __restrict__ cid,
                                    int* __restrict__ counts) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    // the division by h is the one the host used to size the grid: floor((x - origin) / h)
    int c = min(max((int)floor((fit[(long long)i * ND] - ox) / h), 0), nx - 1);
    c += nx * min(max((int)floor((fit[(long long)i * ND + 1] - oy) / h), 0), ny - 1);
    if (ND == 3) c += nx * ny * min(max((int)floor((fit[(long long)i * ND + 2] - oz) / h), 0), nz - 1);
    cid[i] = c;
    atomicAdd(&counts[c], 1);
  }
}

// exclusive scan of counts[0..n) into start[0..n], start[n] = total; counts are zeroed for reuse
// as the scatter cursors
__global__ __launch_bounds__(1024) void bucket_scan_kernel(int* __restrict__ counts, int n, int* __restrict__ start) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = min(tid * per, n), hi = min(lo + per, n);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += counts[i];
  part[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = (tid >= o) ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = (tid == 0) ? 0 : part[tid - 1];
  for (int i = lo; i < hi; ++i) {
    const int c = counts[i];
    start[i] = run;
    counts[i] = 0;
    run += c;
  }
  if (tid == 1023) start[n] = part[1023];
}

__global__ void bucket_scatter_kernel(const int* __restrict__ cid, int n, const int* __restrict__ start,
                                      int* __restrict__ cursor, int* __restrict__ slot_idx) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = cid[i];
    slot_idx[start[c] + atomicAdd(&cursor[c], 1)] = i;
  }
}

constexpr int BUCKET_BIG = 2048;

template <int ND>
__global__ __launch_bounds__(256) void bucket_order_kernel(const double* __restrict__ fit,
                                                           const int* __restrict__ start, int ncells,
                                                           const int* __restrict__ slot_idx,
                                                           double* __restrict__ fit_sorted, int* __restrict__ any_big) {
  const int lane = threadIdx.x & 63;
  for (int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < ncells; c += (gridDim.x * blockDim.x) >> 6) {
    const int lo = start[c], hi = start[c + 1];
    if (hi - lo > BUCKET_BIG) {          // left to bucket_order_big_kernel
      if (lane == 0) *any_big = 1;
      continue;
    }
    for (int j = lo + lane; j < hi; j += 64) {
      const int me = slot_idx[j];
      int rank = 0;
      for (int k = lo; k < hi; ++k) rank += (slot_idx[k] < me) ? 1 : 0;
#pragma unroll
      for (int d = 0; d < ND; ++d) fit_sorted[(long long)(lo + rank) * ND + d] = fit[(long long)me * ND + d];
    }
  }
}

// cells with more than BUCKET_BIG points: one thread per point of such a cell counts the cell's smaller indices
// (every thread of a cell reads the same sequence: broadcast loads out of the caches); returns at once if the
// wavefront kernel saw no such cell
template <int ND>
__global__ __launch_bounds__(256) void bucket_order_big_kernel(const double* __restrict__ fit, int n,
                                                               const int* __restrict__ cid,
                                                               const int* __restrict__ start,
                                                               const int* __restrict__ slot_idx,
                                                               double* __restrict__ fit_sorted,
                                                               const int* __restrict__ any_big) {
  if (*any_big == 0) return;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int me = slot_idx[j];
    const int c = cid[me];
    const int lo = start[c], hi = start[c + 1];
    if (hi - lo <= BUCKET_BIG) continue;
    int rank = 0;
    for (int k = lo; k < hi; ++k) rank += (slot_idx[k] < me) ? 1 : 0;
#pragma unroll
    for (int d = 0; d < ND; ++d) fit_sorted[(long long)(lo + rank) * ND + d] = fit[(long long)me * ND + d];
  }
}

// The same assignment with the centres bucketed into a uniform grid of edge h: a pixel looks at
// the 3^ND cells around its own; every centre outside that block is at least h away, so a
// candidate closer than h (strictly) is the exact nearest centre — ties inside the block are broken
// towards the smaller index, which is the "first minimum" of the plain loop.  A pixel farther
// than h from all its candidates doubles the block radius r (centres outside are >= r h away)
// until the candidate is closer than that or the block is the whole grid (rare: pixels cluster
// around their centre by construction).  Cuts nfg x ncentres pair evaluations to
// nfg x (centres in 3^ND cells).
template <int ND>
__global__ __launch_bounds__(256) void ms_assign_grid_kernel(
    const double* __restrict__ X, const int* __restrict__ index, int nfg, const double* __restrict__ centers,
    int ncenters, const int* __restrict__ order, const int* __restrict__ cell_start, double ox, double oy,
    double oz, double h, int nx, int ny, int nz, int* __restrict__ labels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nfg) return;
  double x[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) x[c] = X[(long long)i * ND + c];
  const double inv = 1.0 / h;
  const int cx = min(max((int)floor((x[0] - ox) * inv), 0), nx - 1);
  const int cy = min(max((int)floor((x[1] - oy) * inv), 0), ny - 1);
  const int cz = (ND == 3) ? min(max((int)floor((x[2] - oz) * inv), 0), nz - 1) : 0;
  double best = 0.0;
  int arg = -1;
  // block of cells within r of the pixel's own: every centre outside it is at least r * h away
  for (int r = 1;; r *= 2) {
    const int x0 = max(cx - r, 0), x1 = min(cx + r, nx - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, ny - 1);
    const int z0 = (ND == 3) ? max(cz - r, 0) : 0, z1 = (ND == 3) ? min(cz + r, nz - 1) : 0;
    arg = -1;
    for (int zz = z0; zz <= z1; ++zz)
      for (int yy = y0; yy <= y1; ++yy) {
        const long long row = ((long long)zz * ny + yy) * nx;
        const int lo = cell_start[row + x0], hi = cell_start[row + x1 + 1];
        for (int j = lo; j < hi; ++j) {
          const int k = order[j];
          double d2 = 0.0;
#pragma unroll
          for (int c = 0; c < ND; ++c) {
            const double df = x[c] - centers[(long long)k * ND + c];
            d2 += df * df;
          }
          if (arg < 0 || d2 < best || (d2 == best && k < arg)) { best = d2; arg = k; }
        }
      }
    const bool whole = x0 == 0 && x1 == nx - 1 && y0 == 0 && y1 == ny - 1 && z0 == 0 && z1 == nz - 1;
    const double reach = (double)r * h;
    if (whole || (arg >= 0 && best < reach * reach)) break;
  }
  labels[index[i]] = arg + 1;
}

// The same search with the centres ALREADY in cell order (cs[j] = centre order[j]): a candidate is one 16-byte load
// from a contiguous range instead of order[j] -> centers[order[j]] (two dependent trips to L2), and in 2-D the
// (up to three) rows' ranges are fetched together before any centre — five dependent memory levels per pixel
// instead of ten.  Same arithmetic and the same tie rule (smaller centre id) as ms_assign_grid_kernel.
// One candidate of the search: the centre cs[j] (cell order) with id order[j]; the winner is the minimum under the total
// order (squared distance, centre id), so the visiting order does not matter.
template <int ND>
__device__ __forceinline__ void ms_candidate(const double (&x)[ND], const double* __restrict__ cs,
                                             const int* __restrict__ order, int j, double& best, int& arg) {
  double c[ND];
  if (ND == 2) {
    const f64x2 q = *reinterpret_cast<const f64x2*>(cs + (long long)j * 2);
    c[0] = q[0]; c[1] = q[1];
  } else {
#pragma unroll
    for (int e = 0; e < ND; ++e) c[e] = cs[(long long)j * ND + e];
  }
  const int k = order[j];
  double d2 = 0.0;
#pragma unroll
  for (int e = 0; e < ND; ++e) {
    const double df = x[e] - c[e];
    d2 += df * df;
  }
  if (arg < 0 || d2 < best || (d2 == best && k < arg)) { best = d2; arg = k; }
}

// The block of cells within r of the pixel's own, r = r0, 2 r0, ...: every centre outside it is at least r h away, so
// a winner closer than that (strictly) is the nearest centre; otherwise the block doubles until it is the whole grid.
template <int ND>
__device__ int ms_ring_search(const double (&x)[ND], int cx, int cy, int cz, int r0, const double* __restrict__ cs,
                              const int* __restrict__ order, const int* __restrict__ cell_start, double h, int nx,
                              int ny, int nz) {
  double best = 0.0;
  int arg = -1;
  for (int r = r0;; r *= 2) {
    const int x0 = max(cx - r, 0), x1 = min(cx + r, nx - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, ny - 1);
    const int z0 = (ND == 3) ? max(cz - r, 0) : 0, z1 = (ND == 3) ? min(cz + r, nz - 1) : 0;
    arg = -1;
    for (int zz = z0; zz <= z1; ++zz)
      for (int yy = y0; yy <= y1; ++yy) {
        const long long row = ((long long)zz * ny + yy) * nx;
        const int lo = cell_start[row + x0], hi = cell_start[row + x1 + 1];
        for (int j = lo; j < hi; ++j) ms_candidate<ND>(x, cs, order, j, best, arg);
      }
    const bool whole = x0 == 0 && x1 == nx - 1 && y0 == 0 && y1 == ny - 1 && z0 == 0 && z1 == nz - 1;
    const double reach = (double)r * h;
    if (whole || (arg >= 0 && best < reach * reach)) break;
  }
  return arg;
}

template <int ND>
__global__ __launch_bounds__(256) void ms_assign_cells_kernel(
    const double* __restrict__ X, const int* __restrict__ index, int nfg, const double* __restrict__ cs,
    const int* __restrict__ order, const int* __restrict__ cell_start, double ox, double oy,
    double oz, double h, double inv, int nx, int ny, int nz, int* __restrict__ labels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nfg) return;
  double x[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) x[c] = X[(long long)i * ND + c];
  const int cx = min(max((int)floor((x[0] - ox) * inv), 0), nx - 1);
  const int cy = min(max((int)floor((x[1] - oy) * inv), 0), ny - 1);
  const int cz = (ND == 3) ? min(max((int)floor((x[2] - oz) * inv), 0), nz - 1) : 0;
  labels[index[i]] = ms_ring_search<ND>(x, cx, cy, cz, 1, cs, order, cell_start, h, nx, ny, nz) + 1;
}

// 2-D search, the common case — the 3 x 3 block holds the winner — as one straight line: the six row-range words of the
// block together; then the block's candidates as ONE sequence (the three rows' ranges end to end), two per trip, instead
// of a loop per row (a wavefront holds pixels of ~3 objects, whose centres sit in different rows of their blocks).
// Offsets are 32-bit (the grid has < 2^31 cells, the host checks), inv = 1 / h comes from the host (the same IEEE
// quotient).  A pixel whose block does not decide (rare: embeddings cluster around their centre) goes on with
// ms_ring_search at r = 2.  Returns the centre id (-1: no centre at all).
__device__ __forceinline__ int ms_search_2d(const double (&x)[2], const double* __restrict__ cs,
                                            const int* __restrict__ order, const int* __restrict__ cell_start, double ox,
                                            double oy, double h, double inv, int nx, int ny) {
  const int cx = min(max((int)floor((x[0] - ox) * inv), 0), nx - 1);
  const int cy = min(max((int)floor((x[1] - oy) * inv), 0), ny - 1);
  const int x0 = max(cx - 1, 0), x1 = min(cx + 1, nx - 1);
  const int y0 = max(cy - 1, 0), y1 = min(cy + 1, ny - 1);
  // (byte offsets in 32 bits off the uniform base: one address register per load; selects, not branches)
  const char* csb = reinterpret_cast<const char*>(cell_start);
  int a[3], n[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const bool ok = y0 + t <= y1;
    const unsigned int row = (unsigned int)(ok ? y0 + t : y0) * (unsigned int)nx;
    const int lo = *reinterpret_cast<const int*>(csb + ((row + (unsigned int)x0) << 2));
    const int hi = *reinterpret_cast<const int*>(csb + ((row + (unsigned int)x1 + 1u) << 2));
    a[t] = lo;
    n[t] = ok ? hi - lo : 0;
  }
  const int n01 = n[0] + n[1], total = n01 + n[2];
  const int a1 = a[1] - n[0], a2 = a[2] - n01;      // candidate s of the sequence: row 0, then row 1, then row 2
  const char* cb = reinterpret_cast<const char*>(cs);
  const char* ob = reinterpret_cast<const char*>(order);
  double best = 0.0;
  int arg = -1;
  for (int s = 0; s < total; s += 2) {
    f64x2 q[2];
    int k[2];
    bool valid[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      valid[u] = s + u < total;
      const int sc = valid[u] ? s + u : s;            // (a lane without a second candidate reads its first again)
      int j = a2 + sc;
      j = sc < n01 ? a1 + sc : j;
      j = sc < n[0] ? a[0] + sc : j;
      q[u] = *reinterpret_cast<const f64x2*>(cb + ((unsigned int)j << 4));
      k[u] = *reinterpret_cast<const int*>(ob + ((unsigned int)j << 2));
    }
    __builtin_amdgcn_sched_barrier(0);                // both candidates' loads go out before the first is looked at
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const double d0 = x[0] - q[u][0], d1 = x[1] - q[u][1];
      double d2 = 0.0;
      d2 += d0 * d0;
      d2 += d1 * d1;
      const bool take = valid[u] & ((arg < 0) | (d2 < best) | ((d2 == best) & (k[u] < arg)));
      best = take ? d2 : best;
      arg = take ? k[u] : arg;
    }
  }
  const bool whole = x0 == 0 && x1 == nx - 1 && y0 == 0 && y1 == ny - 1;
  if (!(whole || (arg >= 0 && best < h * h))) arg = ms_ring_search<2>(x, cx, cy, 0, 2, cs, order, cell_start, h, nx, ny, 1);
  return arg;
}

__global__ __launch_bounds__(256) void ms_assign_cells2d_kernel(
    const double* __restrict__ X, const int* __restrict__ index, int nfg, const double* __restrict__ cs,
    const int* __restrict__ order, const int* __restrict__ cell_start, double ox, double oy, double h, double inv,
    int nx, int ny, int* __restrict__ labels) {
  const unsigned int i = blockIdx.x * 256u + threadIdx.x;
  if (i >= (unsigned int)nfg) return;
  const f64x2 p = *reinterpret_cast<const f64x2*>(X + (size_t)i * 2);
  const int dst = index[i];
  const double x[2] = {p[0], p[1]};
  labels[dst] = ms_search_2d(x, cs, order, cell_start, ox, oy, h, inv, nx, ny) + 1;
}

// The assignment that writes the WHOLE label map (round 5).  The form above scatters one 4-byte label per foreground
// pixel into a map somebody zeroed before: at 8192^2 the scatter alone is a third of its time (93 us; 61 us with the
// store taken out, 65 us with the labels stored densely in point order — partial 32-byte sectors at both ends of
// every run of foreground pixels) and the zero fill of the map is another 268 MB that no row of the table counted.
// Here a wavefront (a block of its own: no barrier between wavefronts) takes one wave-tile (1024 pixels) of the compaction
// and the points the compaction made of it, which are contiguous in X: phase 1 — a lane per POINT (all lanes busy, the
// search above, U points of a lane together) leaves the label in LDS; phase 2 — a lane per 16-byte group of PIXELS looks
// its pixels' bits up in the compaction's flag words
// (still in the workspace clx_ms_prepare filled, with the points in front of every tile), takes the labels of the set
// ones from LDS in order and stores the group: every pixel of the map is written once, in full lines, background as 0;
// no raster index is read, no map is zeroed.
// U points of one thread searched TOGETHER: their six row-range words go out at once, then one candidate of each per
// trip — the dependent trips to memory of one search serve U of them (a block of the dense form below has ~3 points per
// thread and would otherwise walk the chain three times between its two barriers).  Same arithmetic, same order of
// candidates, same tie rule as ms_search_2d; slots that are not `live` cost loads of valid addresses and nothing else.
template <int U>
__device__ __forceinline__ void ms_search_2d_multi(const double (&x)[U][2], const bool (&live)[U],
                                                   const double* __restrict__ cs, int ncenters,
                                                   const int* __restrict__ order, const int* __restrict__ cell_start,
                                                   double ox, double oy, double h, double inv, int nx, int ny,
                                                   int (&arg)[U]) {
  const char* csb = reinterpret_cast<const char*>(cell_start);
  const char* cb = reinterpret_cast<const char*>(cs);
  const char* ob = reinterpret_cast<const char*>(order);
  int cx[U], cy[U], a0[U], a1[U], a2[U], n0[U], n01[U], total[U];
  bool whole[U];
  int lo[U][3], hi[U][3];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    cx[u] = min(max((int)floor((x[u][0] - ox) * inv), 0), nx - 1);
    cy[u] = min(max((int)floor((x[u][1] - oy) * inv), 0), ny - 1);
    const int x0 = max(cx[u] - 1, 0), x1 = min(cx[u] + 1, nx - 1);
    const int y0 = max(cy[u] - 1, 0), y1 = min(cy[u] + 1, ny - 1);
    whole[u] = x0 == 0 && x1 == nx - 1 && y0 == 0 && y1 == ny - 1;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const bool ok = y0 + t <= y1;
      const unsigned int row = (unsigned int)(ok ? y0 + t : y0) * (unsigned int)nx;
      lo[u][t] = *reinterpret_cast<const int*>(csb + ((row + (unsigned int)x0) << 2));
      hi[u][t] = *reinterpret_cast<const int*>(csb + ((row + (unsigned int)x1 + 1u) << 2));
    }
  }
  __builtin_amdgcn_sched_barrier(0);                  // every point's row ranges are on their way before one is used
  int smax = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int y0 = max(cy[u] - 1, 0), y1 = min(cy[u] + 1, ny - 1);
    int n[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) n[t] = (y0 + t <= y1) ? hi[u][t] - lo[u][t] : 0;
    n0[u] = n[0];
    n01[u] = n[0] + n[1];
    total[u] = live[u] ? n01[u] + n[2] : 0;
    a0[u] = lo[u][0];
    a1[u] = lo[u][1] - n0[u];
    a2[u] = lo[u][2] - n01[u];
    smax = max(smax, total[u]);
  }
  double best[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { best[u] = 0.0; arg[u] = -1; }
  for (int s = 0; s < smax; ++s) {
    f64x2 q[U];
    int k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sc = s < total[u] ? s : 0;
      int j = a2[u] + sc;
      j = sc < n01[u] ? a1[u] + sc : j;
      j = sc < n0[u] ? a0[u] + sc : j;
      j = min(j, ncenters - 1);                       // (a slot without candidates: any valid address)
      q[u] = *reinterpret_cast<const f64x2*>(cb + ((unsigned int)j << 4));
      k[u] = *reinterpret_cast<const int*>(ob + ((unsigned int)j << 2));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double d0 = x[u][0] - q[u][0], d1 = x[u][1] - q[u][1];
      double d2 = 0.0;
      d2 += d0 * d0;
      d2 += d1 * d1;
      const bool take = (s < total[u]) & ((arg[u] < 0) | (d2 < best[u]) | ((d2 == best[u]) & (k[u] < arg[u])));
      best[u] = take ? d2 : best[u];
      arg[u] = take ? k[u] : arg[u];
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (live[u] && !(whole[u] || (arg[u] >= 0 && best[u] < h * h)))
      arg[u] = ms_ring_search<2>(x[u], cx[u], cy[u], 0, 2, cs, order, cell_start, h, nx, ny, 1);
}

#ifndef CLX_DENSE_WAVES
#define CLX_DENSE_WAVES 1
#endif
constexpr int DENSE_WAVES = CLX_DENSE_WAVES;          // wave-tiles per block: 1 / 2 / 4 measured 111 / 115 / 116 us at 8192^2
template <int ND, int PXL, int G, int U>
__global__ __launch_bounds__(64 * DENSE_WAVES) void ms_assign_dense_kernel(
    const double* __restrict__ X, const double* __restrict__ cs, int ncenters, const int* __restrict__ order,
    const int* __restrict__ cell_start, double ox, double oy, double oz, double h, double inv, int nx, int ny, int nz,
    const unsigned long long* __restrict__ flags, const int* __restrict__ counts, const int* __restrict__ tile_start,
    long long npix, int vec, int* __restrict__ labels) {
  constexpr int WT = 64 * G * PXL;
  __shared__ int lab_all[DENSE_WAVES * WT];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int* lab = lab_all + wave * WT;                      // (a wavefront's own tile: no barrier between wavefronts)
  const int wt = blockIdx.x * DENSE_WAVES + wave;
  // wave-uniform words first: the tile's points, its flag words (scalar loads, in flight under phase 1)
  const int p0 = tile_start[wt];
  const int total = counts[wt];
  int run = 0;
  const unsigned long long* tf = flags + (long long)wt * G * PXL;
  unsigned long long B[G][PXL];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int e = 0; e < PXL; ++e) B[g][e] = tf[g * PXL + e];
  if (ND == 2) {
    for (int q0 = 0; q0 < total; q0 += 64 * U) {
      double x[U][2];
      bool live[U];
      int arg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = q0 + u * 64 + lane;
        live[u] = q < total;
        const f64x2 p = *reinterpret_cast<const f64x2*>(X + (size_t)(p0 + (live[u] ? q : 0)) * 2);
        x[u][0] = p[0]; x[u][1] = p[1];
      }
      ms_search_2d_multi<U>(x, live, cs, ncenters, order, cell_start, ox, oy, h, inv, nx, ny, arg);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (live[u]) lab[q0 + u * 64 + lane] = arg[u] + 1;
    }
  } else {
    for (int q = lane; q < total; q += 64) {
      double x[ND];
#pragma unroll
      for (int c = 0; c < ND; ++c) x[c] = X[(size_t)(p0 + q) * ND + c];
      const int cx = min(max((int)floor((x[0] - ox) * inv), 0), nx - 1);
      const int cy = min(max((int)floor((x[1] - oy) * inv), 0), ny - 1);
      const int cz = min(max((int)floor((x[ND - 1] - oz) * inv), 0), nz - 1);
      lab[q] = ms_ring_search<ND>(x, cx, cy, cz, 1, cs, order, cell_start, h, nx, ny, nz) + 1;
    }
  }
  __builtin_amdgcn_wave_barrier();                     // the wavefront's LDS operations execute in order
  const long long base = (long long)wt * WT;
  if (base >= npix) return;
  const unsigned long long lower = (1ull << lane) - 1ull;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    int bef = 0, cnt = 0;
#pragma unroll
    for (int e = 0; e < PXL; ++e) {
      bef += __popcll(B[g][e] & lower);
      cnt += __popcll(B[g][e]);
    }
    int pos = run + bef;
    int out[PXL];
#pragma unroll
    for (int e = 0; e < PXL; ++e) {
      const bool fg = (B[g][e] >> lane) & 1ull;
      out[e] = fg ? lab[fg ? pos : 0] : 0;
      pos += fg ? 1 : 0;
    }
    const long long i = base + (long long)(g * 64 + lane) * PXL;
    if (vec) {
      if (i < npix) {
        if constexpr (PXL == 4) *reinterpret_cast<i32x4*>(labels + i) = i32x4{out[0], out[1], out[2], out[3]};
        else *reinterpret_cast<i32x2*>(labels + i) = i32x2{out[0], out[1]};
      }
    } else {
#pragma unroll
      for (int e = 0; e < PXL; ++e)
        if (i + e < npix) labels[i + e] = out[e];
    }
    run += cnt;
  }
}

}  // namespace

// workspace of the compaction: [flags: one bit per pixel, whole wave-tiles][counts: nwt ints][chunk counts of the flags pass: <= nwt / 4 ints]
static long long wave_tiles(long long npix) { return (npix + 4095) / 4096 * 4; }      // whole groups of four

extern "C" size_t clx_ms_prepare_workspace(long long npix) {
  const long long nwt = wave_tiles(npix);
  return (size_t)nwt * 128 + (size_t)(3 * nwt + 2) * sizeof(int) + 64;
}

template <int ND, typename TIn, int G, bool WB, int BLOCKS>
static void launch_prepare(TIn* emb, const TIn* std, double threshold, int Z, int Y, int X, double* Xout, int* index,
                           int* nfg_out, void* workspace, hipStream_t st) {
  constexpr int PXL = 16 / (int)sizeof(TIn);
  static_assert(64 * G * PXL == 1024, "wave-tiles of 1024 pixels");
  const long long npix = (long long)Z * Y * X;
  const int nwt = (int)wave_tiles(npix);
  unsigned long long* flags = (unsigned long long*)workspace;
  int* counts = (int*)(flags + (size_t)nwt * 16);          // 16-byte aligned: nwt is a multiple of 4
  int* chunk_counts = counts + nwt;                         // one per block of the flags pass (<= nwt / 4)
  int* tile_start = counts + 2 * nwt;                       // the points in front of every tile
  // every channel plane starts at a multiple of npix elements: 16-byte accesses need npix % PXL == 0 and aligned bases
  const int vec = (npix % PXL == 0) && (((uintptr_t)emb | (uintptr_t)std) & 15) == 0 ? 1 : 0;
  const FastDiv dX = make_fastdiv((uint32_t)X), dY = make_fastdiv((uint32_t)Y);
  static const int g1cap = getenv("CLX_MS_FLAGS_GRID") ? atoi(getenv("CLX_MS_FLAGS_GRID")) : 2048;        // (sweeps)
  static const int g3cap = getenv("CLX_MS_SCATTER_GRID") ? atoi(getenv("CLX_MS_SCATTER_GRID")) : 1 << 20;
  // the flags pass in at most g1cap blocks of contiguous chunks, a multiple of four tiles each
  int tiles_per_block = (nwt + g1cap - 1) / (g1cap > 0 ? g1cap : 1);
  tiles_per_block = (tiles_per_block + 3) / 4 * 4;
  const int g1 = (nwt + tiles_per_block - 1) / tiles_per_block;
  const int nb = (nwt + 3) / 4;
  const int g3 = nb < g3cap ? nb : g3cap;
  CLX_LAUNCH_KIND(CLX_PROF_MS_PREPARE, (ms_flags_kernel<TIn, G>), dim3(g1), dim3(256), 0, st, std, threshold, npix, vec,
                  nwt, tiles_per_block, flags, counts, chunk_counts);
  CLX_LAUNCH_KIND(CLX_PROF_MS_PREPARE, (ms_scatter_kernel<ND, TIn, G, WB, BLOCKS>), dim3(g3), dim3(256), 0, st, emb, dX, dY,
                  Y, X, npix, vec, nwt, flags, counts, chunk_counts, tiles_per_block, Xout, index, nfg_out, tile_start);
}

extern "C" int clx_ms_prepare(double* emb, const double* std, double threshold, int ND,
                              int Z, int Y, int X, double* Xout, int* index, int* nfg_out,
                              void* workspace, clx_stream stream) {
  CLX_REQUIRE(emb && std && Xout && nfg_out && workspace, "clx_ms_prepare: null pointer");
  CLX_REQUIRE((ND == 2 || ND == 3) && Z > 0 && Y > 0 && X > 0, "clx_ms_prepare: bad extents");
  CLX_REQUIRE(ND == 3 || Z == 1, "clx_ms_prepare: Z must be 1 for 2-D data");
  CLX_REQUIRE(((uintptr_t)workspace & 15) == 0, "clx_ms_prepare: workspace must be 16-byte aligned");
  const long long npix = (long long)Z * Y * X;
  CLX_REQUIRE(npix < (1ll << 31), "clx_ms_prepare: too many pixels");
  hipStream_t st = (hipStream_t)stream;
  if (ND == 2) launch_prepare<2, double, 8, true, 3>(emb, std, threshold, Z, Y, X, Xout, index, nfg_out, workspace, st);
  else launch_prepare<3, double, 8, true, 2>(emb, std, threshold, Z, Y, X, Xout, index, nfg_out, workspace, st);
  CLX_CHECK_LAUNCH("clx_ms_prepare");
  return CLX_OK;
}

extern "C" int clx_ms_prepare_f32(const float* emb, const float* std, double threshold, int ND,
                                  int Z, int Y, int X, double* Xout, int* index, int* nfg_out,
                                  void* workspace, clx_stream stream) {
  CLX_REQUIRE(emb && std && Xout && nfg_out && workspace, "clx_ms_prepare_f32: null pointer");
  CLX_REQUIRE((ND == 2 || ND == 3) && Z > 0 && Y > 0 && X > 0, "clx_ms_prepare_f32: bad extents");
  CLX_REQUIRE(ND == 3 || Z == 1, "clx_ms_prepare_f32: Z must be 1 for 2-D data");
  CLX_REQUIRE(((uintptr_t)workspace & 15) == 0, "clx_ms_prepare_f32: workspace must be 16-byte aligned");
  const long long npix = (long long)Z * Y * X;
  CLX_REQUIRE(npix < (1ll << 31), "clx_ms_prepare_f32: too many pixels");
  hipStream_t st = (hipStream_t)stream;
  float* e = const_cast<float*>(emb);          // (WB = false: never written)
  if (ND == 2) launch_prepare<2, float, 4, false, 4>(e, std, threshold, Z, Y, X, Xout, index, nfg_out, workspace, st);
  else launch_prepare<3, float, 4, false, 3>(e, std, threshold, Z, Y, X, Xout, index, nfg_out, workspace, st);
  CLX_CHECK_LAUNCH("clx_ms_prepare_f32");
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
    CLX_LAUNCH_KIND(CLX_PROF_MS_ASSIGN, (ms_assign_kernel<2>), dim3(grid), dim3(256), 0, st, X, index, nfg, centers, ncenters, labels);
  else
    CLX_LAUNCH_KIND(CLX_PROF_MS_ASSIGN, (ms_assign_kernel<3>), dim3(grid), dim3(256), 0, st, X, index, nfg, centers, ncenters, labels);
  CLX_CHECK_LAUNCH("clx_ms_assign");
  return CLX_OK;
}

extern "C" int clx_ms_assign_grid(const double* X, const int* index, int nfg, const double* centers,
                                  int ncenters, int ND, const int* order, const int* cell_start,
                                  const double* origin, double cell, int nx, int ny, int nz, int* labels,
                                  clx_stream stream) {
  CLX_REQUIRE(X && index && centers && labels && order && cell_start && origin, "clx_ms_assign_grid: null pointer");
  CLX_REQUIRE((ND == 2 || ND == 3) && nfg >= 0 && ncenters > 0, "clx_ms_assign_grid: bad extents");
  CLX_REQUIRE(cell > 0.0 && nx > 0 && ny > 0 && nz > 0 && (ND == 3 || nz == 1), "clx_ms_assign_grid: bad grid");
  if (nfg == 0) return CLX_OK;
  const int grid = (nfg + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  if (ND == 2)
    CLX_LAUNCH_KIND(CLX_PROF_MS_ASSIGN, (ms_assign_grid_kernel<2>), dim3(grid), dim3(256), 0, st, X, index, nfg, centers, ncenters, order, cell_start, origin[0],
                                                    origin[1], 0.0, cell, nx, ny, nz, labels);
  else
    CLX_LAUNCH_KIND(CLX_PROF_MS_ASSIGN, (ms_assign_grid_kernel<3>), dim3(grid), dim3(256), 0, st, X, index, nfg, centers, ncenters, order, cell_start, origin[0],
                                                    origin[1], origin[2], cell, nx, ny, nz, labels);
  CLX_CHECK_LAUNCH("clx_ms_assign_grid");
  return CLX_OK;
}

extern "C" int clx_ms_assign_cells(const double* X, const int* index, int nfg, const double* centers_sorted,
                                   int ncenters, int ND, const int* order, const int* cell_start,
                                   const double* origin, double cell, int nx, int ny, int nz, int* labels,
                                   clx_stream stream) {
  CLX_REQUIRE(X && index && centers_sorted && labels && order && cell_start && origin, "clx_ms_assign_cells: null pointer");
  CLX_REQUIRE((ND == 2 || ND == 3) && nfg >= 0 && ncenters > 0, "clx_ms_assign_cells: bad extents");
  CLX_REQUIRE(cell > 0.0 && nx > 0 && ny > 0 && nz > 0 && (ND == 3 || nz == 1), "clx_ms_assign_cells: bad grid");
  CLX_REQUIRE((long long)nx * ny * nz < (1ll << 31) - 1, "clx_ms_assign_cells: more than 2^31 cells");
  CLX_REQUIRE(ND == 3 || ((((uintptr_t)X | (uintptr_t)centers_sorted) & 15) == 0), "clx_ms_assign_cells: 16-byte alignment");
  if (nfg == 0) return CLX_OK;
  const int grid = (nfg + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  const double inv = 1.0 / cell;
  if (ND == 2)
    CLX_LAUNCH_KIND(CLX_PROF_MS_ASSIGN, ms_assign_cells2d_kernel, dim3(grid), dim3(256), 0, st, X, index, nfg, centers_sorted, order, cell_start,
                    origin[0], origin[1], cell, inv, nx, ny, labels);
  else
    CLX_LAUNCH_KIND(CLX_PROF_MS_ASSIGN, (ms_assign_cells_kernel<3>), dim3(grid), dim3(256), 0, st, X, index, nfg, centers_sorted, order, cell_start, origin[0],
                                                     origin[1], origin[2], cell, inv, nx, ny, nz, labels);
  CLX_CHECK_LAUNCH("clx_ms_assign_cells");
  return CLX_OK;
}

template <int ND, int PXL, int G>
static void launch_assign_dense(const double* X, const double* cs, int ncenters, const int* order, const int* cell_start,
                                const double* origin, double cell, int nx, int ny, int nz, const void* workspace,
                                long long npix, int* labels, hipStream_t st) {
  const int nwt = (int)wave_tiles(npix);
  const unsigned long long* flags = (const unsigned long long*)workspace;
  const int* counts = (const int*)(flags + (size_t)nwt * 16);
  const int* tile_start = counts + 2 * nwt;
  const int vec = (npix % PXL == 0) && (((uintptr_t)labels) & 15) == 0 ? 1 : 0;
  static const int u = getenv("CLX_MS_DENSE_U") ? atoi(getenv("CLX_MS_DENSE_U")) : 2;      // (sweep: 104 us at 8192^2; 1: 128, 3: 117, 4: 127)
#define CLX_DENSE(UU) CLX_LAUNCH_KIND(CLX_PROF_MS_ASSIGN, (ms_assign_dense_kernel<ND, PXL, G, UU>), dim3(nwt / DENSE_WAVES), dim3(64 * DENSE_WAVES), 0, st, X, cs, ncenters, order, \
                  cell_start, origin[0], origin[1], ND == 3 ? origin[2] : 0.0, cell, 1.0 / cell, nx, ny, nz, flags, counts, \
                  tile_start, npix, vec, labels)
  if (ND == 2 && u == 1) CLX_DENSE(1); else if (ND == 2 && u == 2) CLX_DENSE(2); else if (ND == 2 && u == 3) CLX_DENSE(3); else if (ND == 2 && u == 4) CLX_DENSE(4); else CLX_DENSE(2);
#undef CLX_DENSE
}

extern "C" int clx_ms_assign_dense(const double* X, const double* centers_sorted, int ncenters, int ND, const int* order,
                                   const int* cell_start, const double* origin, double cell, int nx, int ny, int nz,
                                   const void* prepare_workspace, int prepared_from_f32, int Z, int Y, int Xdim,
                                   int* labels, clx_stream stream) {
  CLX_REQUIRE(X && centers_sorted && labels && order && cell_start && origin && prepare_workspace,
              "clx_ms_assign_dense: null pointer");
  CLX_REQUIRE((ND == 2 || ND == 3) && ncenters > 0 && Z > 0 && Y > 0 && Xdim > 0 && (ND == 3 || Z == 1),
              "clx_ms_assign_dense: bad extents");
  CLX_REQUIRE(cell > 0.0 && nx > 0 && ny > 0 && nz > 0 && (ND == 3 || nz == 1), "clx_ms_assign_dense: bad grid");
  CLX_REQUIRE((long long)nx * ny * nz < (1ll << 31) - 1, "clx_ms_assign_dense: more than 2^31 cells");
  CLX_REQUIRE(ND == 3 || ((((uintptr_t)X | (uintptr_t)centers_sorted) & 15) == 0), "clx_ms_assign_dense: 16-byte alignment");
  CLX_REQUIRE(((uintptr_t)prepare_workspace & 15) == 0, "clx_ms_assign_dense: workspace must be 16-byte aligned");
  const long long npix = (long long)Z * Y * Xdim;
  CLX_REQUIRE(npix < (1ll << 31), "clx_ms_assign_dense: too many pixels");
  hipStream_t st = (hipStream_t)stream;
  if (prepared_from_f32) {
    if (ND == 2) launch_assign_dense<2, 4, 4>(X, centers_sorted, ncenters, order, cell_start, origin, cell, nx, ny, nz, prepare_workspace, npix, labels, st);
    else launch_assign_dense<3, 4, 4>(X, centers_sorted, ncenters, order, cell_start, origin, cell, nx, ny, nz, prepare_workspace, npix, labels, st);
  } else {
    if (ND == 2) launch_assign_dense<2, 2, 8>(X, centers_sorted, ncenters, order, cell_start, origin, cell, nx, ny, nz, prepare_workspace, npix, labels, st);
    else launch_assign_dense<3, 2, 8>(X, centers_sorted, ncenters, order, cell_start, origin, cell, nx, ny, nz, prepare_workspace, npix, labels, st);
  }
  CLX_CHECK_LAUNCH("clx_ms_assign_dense");
  return CLX_OK;
}

// ---------------------------------------------------------------------------------------------
// HOST: sklearn MeanShift.fit's post-processing of the converged seeds (_mean_shift.py: center_intensity_dict ->
// sorted by (count, centre) descending -> greedy removal of every centre within `bandwidth` of a kept one), the step
// between clx_ms_iterate* and clx_ms_assign* [cellulus/utils/mean_shift.py:62-74 -> MeanShift.fit].  Sequential by
// definition; in numpy it was the largest single cost of the detect stage per 512^2 sample (a Python trip per kept
// centre over all ~4 500 seeds).  Here: two sorts and the greedy pass over a hash grid of edge `bandwidth`.
// ---------------------------------------------------------------------------------------------
extern "C" int clx_ms_dedup_centers(const double* centers, const int* counts, int n, int ND, double bandwidth,
                                    double* out, int* n_out) {
  CLX_REQUIRE(n_out != nullptr && n >= 0 && (ND == 2 || ND == 3) && (n == 0 || (centers && counts && out)),
              "clx_ms_dedup_centers: bad arguments");
  CLX_REQUIRE(bandwidth > 0.0, "clx_ms_dedup_centers: bandwidth must be positive");
  // a NaN coordinate breaks the strict weak ordering std::sort requires (undefined behaviour, not just a garbage order)
  for (int i = 0; i < n; ++i)
    if (counts[i] > 0)
      for (int d = 0; d < ND; ++d)
        CLX_REQUIRE(std::isfinite(centers[(size_t)i * ND + d]), "clx_ms_dedup_centers: centre %d has a non-finite coordinate", i);
  try {        // (allocation failures must not cross the C boundary: ctypes would std::terminate the process)
  auto at = [&](int i, int d) { return centers[(size_t)i * ND + d]; };
  // dict semantics: identical centre tuples are one key (first occurrence), the LAST count wins
  struct Item { double c[3]; int src; int count; };
  std::vector<Item> items;
  items.reserve(n);
  for (int i = 0; i < n; ++i)
    if (counts[i] > 0) {
      Item it;
      for (int d = 0; d < 3; ++d) it.c[d] = d < ND ? at(i, d) : 0.0;
      it.src = i;
      it.count = counts[i];
      items.push_back(it);
    }
  std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) {
    for (int d = 0; d < 3; ++d) {
      if (a.c[d] < b.c[d]) return true;
      if (a.c[d] > b.c[d]) return false;
    }
    return a.src < b.src;
  });
  struct Key { int src; int count; double c[3]; };
  std::vector<Key> keys;
  keys.reserve(items.size());
  for (size_t k = 0; k < items.size();) {
    size_t e = k + 1;
    while (e < items.size() && items[k].c[0] == items[e].c[0] && items[k].c[1] == items[e].c[1] && items[k].c[2] == items[e].c[2]) ++e;
    Key key;
    key.src = items[k].src;                               // (ties in the tuple are ordered by index: first, last)
    key.count = items[e - 1].count;
    for (int d = 0; d < 3; ++d) key.c[d] = items[k].c[d];
    keys.push_back(key);
    k = e;
  }
  // sorted(items, key = (count, centre tuple), reverse = True)
  std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
    if (a.count != b.count) return a.count > b.count;
    for (int d = 0; d < 3; ++d) {
      if (a.c[d] > b.c[d]) return true;
      if (a.c[d] < b.c[d]) return false;
    }
    return false;
  });
  const int m = (int)keys.size();
  const double bw2 = bandwidth * bandwidth;
  auto within = [&](int a, int b) {
    double d2 = 0.0;
    for (int d = 0; d < ND; ++d) {
      const double df = keys[a].c[d] - keys[b].c[d];
      d2 += df * df;
    }
    return d2 <= bw2;
  };
  std::vector<char> unique((size_t)m, 1);
  // grid of edge `bandwidth`: everything within the radius of a centre sits in the 3^ND cells around its own
  double lo[3] = {0, 0, 0};
  bool grid_ok = m > 64;
  for (int d = 0; d < ND && grid_ok; ++d) {
    double mn = at(keys[0].src, d), mx = mn;
    for (int k = 0; k < m; ++k) {
      const double v = at(keys[k].src, d);
      if (!std::isfinite(v)) grid_ok = false;
      mn = std::min(mn, v);
      mx = std::max(mx, v);
    }
    lo[d] = mn;
    if (!((mx - mn) / bandwidth < 2097152.0)) grid_ok = false;      // 21 bits per axis in the cell key
  }
  if (grid_ok) {
    auto cell_of = [&](int k, long long (&c)[3]) {
      c[0] = c[1] = c[2] = 0;
      for (int d = 0; d < ND; ++d) c[d] = (long long)std::floor((at(keys[k].src, d) - lo[d]) / bandwidth);
    };
    auto key_of = [](long long x, long long y, long long z) { return (unsigned long long)((z << 42) | (y << 21) | x); };
    std::vector<std::pair<unsigned long long, int>> cells((size_t)m);        // (cell key, centre), sorted by key
    for (int k = 0; k < m; ++k) {
      long long c[3];
      cell_of(k, c);
      cells[k] = std::make_pair(key_of(c[0], c[1], c[2]), k);
    }
    std::sort(cells.begin(), cells.end());
    for (int i = 0; i < m; ++i) {
      if (!unique[i]) continue;
      long long c[3];
      cell_of(i, c);
      for (long long dz = (ND == 3 ? -1 : 0); dz <= (ND == 3 ? 1 : 0); ++dz)
        for (long long dy = -1; dy <= 1; ++dy) {
          const long long y = c[1] + dy, z = c[2] + dz;
          if (y < 0 || z < 0) continue;
          // the three cells of a row of the block are consecutive keys
          const long long x0 = c[0] > 0 ? c[0] - 1 : 0;
          auto it = std::lower_bound(cells.begin(), cells.end(), std::make_pair(key_of(x0, y, z), -1));
          const unsigned long long last = key_of(c[0] + 1, y, z);
          for (; it != cells.end() && it->first <= last; ++it)
            if (within(it->second, i)) unique[it->second] = 0;
        }
      unique[i] = 1;
    }
  } else {
    for (int i = 0; i < m; ++i) {
      if (!unique[i]) continue;
      for (int j = 0; j < m; ++j)
        if (within(j, i)) unique[j] = 0;
      unique[i] = 1;
    }
  }
  int kept = 0;
  for (int k = 0; k < m; ++k)
    if (unique[k]) {
      for (int d = 0; d < ND; ++d) out[(size_t)kept * ND + d] = at(keys[k].src, d);
      ++kept;
    }
  *n_out = kept;
  return CLX_OK;
  } catch (const std::exception& e) {
    clx_set_error("clx_ms_dedup_centers: %s", e.what());
    return CLX_ERR_WORKSPACE;
  }
}

extern "C" size_t clx_ms_bucket_workspace(int n, long long ncells) {
  return (size_t)(2 * (long long)n + ncells + 4) * sizeof(int);
}

extern "C" int clx_ms_bucket(const double* fit, int n, int ND, const double* origin, double cell, int nx,
                             int ny, int nz, double* fit_sorted, int* cell_start, void* workspace,
                             clx_stream stream) {
  CLX_REQUIRE(fit && origin && fit_sorted && cell_start && workspace, "clx_ms_bucket: null pointer");
  CLX_REQUIRE((ND == 2 || ND == 3) && n >= 0 && cell > 0.0, "clx_ms_bucket: bad extents");
  CLX_REQUIRE(nx > 0 && ny > 0 && nz > 0 && (ND == 3 || nz == 1), "clx_ms_bucket: bad grid");
  const long long ncells = (long long)nx * ny * nz;
  CLX_REQUIRE(ncells < (1ll << 30), "clx_ms_bucket: too many cells");
  hipStream_t st = (hipStream_t)stream;
  int* cid = (int*)workspace;
  int* slot_idx = cid + n;
  int* counts = slot_idx + n;
  int* any_big = counts + ncells;           // one flag behind the counters, cleared with them
  if (hipMemsetAsync(counts, 0, (size_t)(ncells + 1) * sizeof(int), st) != hipSuccess) {
    clx_set_error("clx_ms_bucket: memset failed");
    return CLX_ERR_LAUNCH;
  }
  const int grid = n > 0 ? (n + 255) / 256 : 1;
  const double oz = ND == 3 ? origin[2] : 0.0;
  if (n > 0) {
    if (ND == 2)
      bucket_count_kernel<2><<<grid, 256, 0, st>>>(fit, n, origin[0], origin[1], oz, cell, nx, ny, nz, cid, counts);
    else
      bucket_count_kernel<3><<<grid, 256, 0, st>>>(fit, n, origin[0], origin[1], oz, cell, nx, ny, nz, cid, counts);
  }
  bucket_scan_kernel<<<1, 1024, 0, st>>>(counts, (int)ncells, cell_start);
  if (n > 0) {
    bucket_scatter_kernel<<<grid, 256, 0, st>>>(cid, n, cell_start, counts, slot_idx);
    long long waves = ncells < 65536 ? ncells : 65536;
    const int ogrid = (int)((waves * 64 + 255) / 256);
    const int bgrid = grid < 8192 ? grid : 8192;
    if (ND == 2) {
      bucket_order_kernel<2><<<ogrid, 256, 0, st>>>(fit, cell_start, (int)ncells, slot_idx, fit_sorted, any_big);
      bucket_order_big_kernel<2><<<bgrid, 256, 0, st>>>(fit, n, cid, cell_start, slot_idx, fit_sorted, any_big);
    } else {
      bucket_order_kernel<3><<<ogrid, 256, 0, st>>>(fit, cell_start, (int)ncells, slot_idx, fit_sorted, any_big);
      bucket_order_big_kernel<3><<<bgrid, 256, 0, st>>>(fit, n, cid, cell_start, slot_idx, fit_sorted, any_big);
    }
  }
  CLX_CHECK_LAUNCH("clx_ms_bucket");
  return CLX_OK;
}
