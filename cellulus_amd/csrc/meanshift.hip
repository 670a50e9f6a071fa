// Mean-shift clustering of per-pixel embeddings in float64 for gfx950.
// Compiled with -ffp-contract=off: the membership test d^2 <= bw^2 must use
// the same un-fused arithmetic as the reference's KD-tree (sum of squared
// differences, one rounding per operation).
//
// Replaces cellulus/utils/mean_shift.py:6-121 -> sklearn.cluster.MeanShift
// (fit: _mean_shift_single_seed per seed; predict: nearest centre).
#include <stdlib.h>
#include "clx_common.h"

namespace {

constexpr int PREP_TILE_MAX = 4096;   // pixels per block of the compaction: 256 threads x KP pairs, KP = 4 or 8

typedef double f64x2 __attribute__((ext_vector_type(2)));

// Coordinate add + stable (raster-order) compaction of the foreground pixels in ONE pass over
// the image [mean_shift.py:15-32,83-90]: every block takes the next tile (atomic ticket), adds the
// pixel coordinates to the embedding in place, counts its foreground pixels, publishes the count
// and obtains the number of foreground pixels before its tile by decoupled look-back over the
// predecessors' published counts / prefixes (one 64-bit word each: status in the high half,
// value in the low half, so a reader never sees one without the other — and because that word is
// ALL the blocks tell each other, the atomics are relaxed: an agent-scope release / acquire on this
// 8-XCD part writes back / invalidates a whole L2 per descriptor), then writes its points.
// Per pixel: (ND + 1) * 8 B read, ND * 8 B written; per foreground pixel ND * 8 + 4 B more.
#ifndef MSP_WAVES
#define MSP_WAVES 3
#endif
template <int ND, int KP>
__global__ __launch_bounds__(256, MSP_WAVES) void ms_prepare_kernel(double* __restrict__ emb,
                                                         const double* __restrict__ sd, double thr,
                                                         FastDiv dX, FastDiv dY, int Y, int X,
                                                         long long npix, int vec, int ntiles,
                                                         unsigned int* __restrict__ ticket,   // [0] next tile, [1] blocks done
                                                         unsigned long long* __restrict__ desc,
                                                         double* __restrict__ Xout,
                                                         int* __restrict__ index, int* __restrict__ nfg_out) {
  // Persistent blocks: a block takes the next tile by ticket until the tiles run out (no partial last round of
  // blocks; the NEXT ticket is requested while the current tile is processed, so its latency is never waited for).
  // Per tile: (A) the std channel alone decides foreground — load it first, count, and PUBLISH the tile's
  // aggregate before anything else, (B) the embedding channels: add the coordinates in place, (C) look back — by now
  // the predecessors' aggregates are old — and write the tile's points.  (With the aggregate published after all of a
  // tile's loads, as in round 2, the look-back cost 35-55 us of a 4096^2 image's 170.  Looking back BEFORE (B) and
  // writing the points as the values arrive needs half the registers but serialises the loads: 238 against 179 us.)
  constexpr int PREP_TILE = 512 * KP;
  __shared__ int s_tile[2], s_excl;
  __shared__ int wcount[KP][4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const unsigned long long lower = (1ull << lane) - 1ull;
  // (taking several tiles per ticket — the word serves ~88 returning atomics per microsecond — delays the later tiles'
  //  aggregates, and every look-back behind them waits: 13 ms instead of 0.17.  Fewer tickets need larger units whose
  //  aggregate is published at once: ms_compact_kernel below.)
  if (tid == 0) s_tile[0] = (int)atomicAdd(ticket, 1u);
  __syncthreads();
  for (int round = 0;; ++round) {
    const int tile = s_tile[round & 1];
    if (tile >= ntiles) break;
    if (tid == 0) s_tile[(round + 1) & 1] = (int)atomicAdd(ticket, 1u);
    const long long base = (long long)tile * PREP_TILE;

    // ---- (A) foreground flags, counts, aggregate
    bool fg[KP][2];
    int before[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const long long i = base + (long long)(k * 256 + tid) * 2;
      fg[k][0] = fg[k][1] = false;
      if (i < npix) {
        if (vec) {              // npix even and 16-byte aligned: i + 1 < npix, 16-byte accesses
          const f64x2 s2 = *reinterpret_cast<const f64x2*>(sd + i);
          fg[k][0] = s2[0] < thr; fg[k][1] = s2[1] < thr;
        } else {
          fg[k][0] = sd[i] < thr;
          fg[k][1] = i + 1 < npix && sd[i + 1] < thr;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const unsigned long long b0 = __ballot(fg[k][0]), b1 = __ballot(fg[k][1]);
      // raster order inside the wave: lane l owns pixels 2l, 2l+1
      before[k] = __popcll(b0 & lower) + __popcll(b1 & lower);
      if (lane == 0) wcount[k][wid] = __popcll(b0) + __popcll(b1);
    }
    __syncthreads();
    int total = 0, mine[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (w == wid) mine[k] = total;
        total += wcount[k][w];
      }
    if (tid == 0)
      __hip_atomic_store(&desc[tile], ((tile == 0 ? 2ull : 1ull) << 32) | (unsigned int)total, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);

    // ---- (B) coordinates added to the embedding in place (all of the tile's loads in flight at once; issuing them
    // before (A)'s barrier as well costs a third block per CU: 201 against 179 us)
    double v[KP][2][ND];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const long long i = base + (long long)(k * 256 + tid) * 2;
      if (i < npix) {
        const bool two = i + 1 < npix;
        const unsigned int t = fdiv((unsigned int)i, dX);
        const int x0 = (int)((unsigned int)i - t * (unsigned int)X);
        const unsigned int z0u = fdiv(t, dY);
        const int y0 = (int)(t - z0u * (unsigned int)Y), z0 = (int)z0u;
        int x1 = x0 + 1, y1 = y0, z1 = z0;
        if (x1 == X) { x1 = 0; if (++y1 == Y) { y1 = 0; ++z1; } }
        const int c0[3] = {x0, y0, z0}, c1[3] = {x1, y1, z1};
        if (vec) {
#pragma unroll
          for (int c = 0; c < ND; ++c) {
            f64x2 e = *reinterpret_cast<const f64x2*>(emb + (long long)c * npix + i);
            e[0] += (double)c0[c]; e[1] += (double)c1[c];
            *reinterpret_cast<f64x2*>(emb + (long long)c * npix + i) = e;
            v[k][0][c] = e[0]; v[k][1][c] = e[1];
          }
        } else {
#pragma unroll
          for (int c = 0; c < ND; ++c) {
            double e0 = emb[(long long)c * npix + i] + (double)c0[c];
            emb[(long long)c * npix + i] = e0;
            v[k][0][c] = e0;
            if (two) {
              double e1 = emb[(long long)c * npix + i + 1] + (double)c1[c];
              emb[(long long)c * npix + i + 1] = e1;
              v[k][1][c] = e1;
            }
          }
        }
      }
    }

    // ---- (C) decoupled look-back by the first wavefront: lane l inspects predecessor tile - 1 - l of the
    // current window of 64; the nearest predecessor that already knows its inclusive prefix ends
    // the walk, the aggregates in front of it are summed (a single thread doing this one
    // descriptor at a time serialises the whole grid)
    if (wid == 0) {
      int excl = 0;
      for (int hi = tile - 1; hi >= 0; hi -= 64) {
        const int j = hi - lane;
        unsigned long long d = 2ull << 32;              // tiles before the first: prefix 0
        if (j >= 0) {
          do {
            d = __hip_atomic_load(&desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((unsigned int)(d >> 32) == 0);       // predecessor has not published yet
        }
        const unsigned long long has_prefix = __ballot((unsigned int)(d >> 32) == 2u);
        const int stop = has_prefix ? __builtin_ctzll(has_prefix) : 64;      // nearest tile with a prefix
        int a = (lane <= stop) ? (int)(unsigned int)d : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        excl += a;
        if (has_prefix) break;
      }
      if (lane == 0) {
        if (tile > 0)
          __hip_atomic_store(&desc[tile], (2ull << 32) | (unsigned int)(excl + total), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        s_excl = excl;
        if (tile == ntiles - 1) *nfg_out = excl + total;
      }
    }
    __syncthreads();
    const int excl = s_excl;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      int pos = excl + mine[k] + before[k];
      const long long i = base + (long long)(k * 256 + tid) * 2;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (fg[k][e]) {
#pragma unroll
          for (int c = 0; c < ND; ++c) Xout[(long long)pos * ND + c] = v[k][e][c];
          index[pos] = (int)(i + e);
          ++pos;
        }
      }
    }
    __syncthreads();      // s_excl, wcount and the ticket slots are reused by the next round
  }
  // The workspace is handed back ZEROED: the block that finishes last (every other block has left its last look-back)
  // clears the descriptors and the two counters — a fill launch in front of every call was 4 % of a 4096^2 image.
  if (tid == 0) s_excl = (atomicAdd(ticket + 1, 1u) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (s_excl) {
    for (int j = tid; j < ntiles; j += 256)
      __hip_atomic_store(&desc[j], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 2) __hip_atomic_store(ticket + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------
// Round 4: the same single-pass compaction in SUPER-TILES, for the float32 hand-over of infer()'s fused predict -> detect
// path (clx_ms_prepare_f32): the float64 values the staged path reads back from the `embeddings` dataset are the network's
// floats widened (cellulus/predict.py:104-112), so widening in registers gives the same bits; the embedding is NOT
// modified (the reference's in-place coordinate add lands in a copy that detect.py:155-160 throws away) and is read
// only where a 16-byte group holds a foreground pixel: per pixel 4 B read, per foreground pixel ND * 4 B read and
// ND * 8 + 4 B written.  With so few bytes per pixel the per-TILE costs of the float64 kernel's structure dominated (a
// first float32 version with 4096-pixel tiles: 119 us at 4096^2): one returning atomic per tile on ONE ticket word —
// the word serves ~88 of them per microsecond, 46 us — and a prefix that travels 64 tiles per look-back hop.  Here a
// block takes a super-tile of R sub-tiles (16 K pixels) per ticket: all of its std values first (flags in one 64-bit
// register, counts scanned in LDS, ONE aggregate published), a look-back by all 256 threads (256 predecessors per hop),
// then sub-tile by sub-tile the embedding values, whose loads run one sub-tile ahead of the point writes: 91 us.
// (TIn = double, WB = true is the float64 form with the in-place add; measured slower than ms_prepare_kernel — 219
// against 175 us at 4096^2: half the bytes in flight per block — and not dispatched.)
// ---------------------------------------------------------------------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<double> { typedef f64x2 type; static constexpr int N = 2; };
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };

template <int ND, typename TIn, int G, int R, bool WB, int BLOCKS>
__global__ __launch_bounds__(256, BLOCKS) void ms_compact_kernel(TIn* __restrict__ emb, const TIn* __restrict__ sd,
                                                                     double thr, FastDiv dX, FastDiv dY, int Y, int X,
                                                                     long long npix, int vec, int nsuper,
                                                                     unsigned int* __restrict__ ticket,
                                                                     unsigned long long* __restrict__ desc,
                                                                     double* __restrict__ Xout, int* __restrict__ index,
                                                                     int* __restrict__ nfg_out) {
  using V = typename Vec16<TIn>::type;
  constexpr int PXL = Vec16<TIn>::N;            // pixels per 16-byte group
  constexpr int SUB = 256 * G * PXL;            // pixels per sub-tile
  constexpr int SUPER = SUB * R;
  constexpr int NK = R * G * 4;                 // (sub-tile, group, wave) counts, in raster order
  constexpr unsigned int GMASK = (1u << PXL) - 1u;
  constexpr unsigned int SMASK = (G * PXL == 32) ? 0xffffffffu : ((1u << (G * PXL)) - 1u);
  static_assert(R * G * PXL <= 64 && NK <= 256 && R % 2 == 0, "flags of a super-tile fit one 64-bit register; one scan pass");
  __shared__ int s_tile[2];
  __shared__ int woff[NK + 1];
  __shared__ int lb_sum[4], lb_stop[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const unsigned long long lower = (1ull << lane) - 1ull;
  if (tid == 0) s_tile[0] = (int)atomicAdd(ticket, 1u);
  __syncthreads();
  for (int round = 0;; ++round) {
    const int tile = s_tile[round & 1];
    if (tile >= nsuper) break;
    if (tid == 0) s_tile[(round + 1) & 1] = (int)atomicAdd(ticket, 1u);
    const long long base = (long long)tile * SUPER;

    // ---- (A) every std value of the super-tile: foreground flags (one bit per pixel of this thread, sub-tile r in
    // bits r * G * PXL ..), counts per (sub-tile, group, wave); two sub-tiles' loads in flight at a time
    unsigned long long allbits = 0ull;
#pragma unroll 2
    for (int r = 0; r < R; ++r) {
      unsigned int b = 0u;
      if (vec) {
        V sv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const long long i = base + (long long)r * SUB + (long long)(g * 256 + tid) * PXL;
          if (i < npix) sv[g] = *reinterpret_cast<const V*>(sd + i);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const long long i = base + (long long)r * SUB + (long long)(g * 256 + tid) * PXL;
          if (i < npix) {
#pragma unroll
            for (int e = 0; e < PXL; ++e) b |= ((double)sv[g][e] < thr ? 1u : 0u) << (g * PXL + e);
          }
        }
      } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const long long i = base + (long long)r * SUB + (long long)(g * 256 + tid) * PXL;
#pragma unroll
          for (int e = 0; e < PXL; ++e)
            if (i + e < npix) b |= ((double)sd[i + e] < thr ? 1u : 0u) << (g * PXL + e);
        }
      }
      allbits |= (unsigned long long)b << (r * G * PXL);
    }
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
      const unsigned int b = (unsigned int)(allbits >> (r * G * PXL)) & SMASK;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        int tot = 0;
#pragma unroll
        for (int e = 0; e < PXL; ++e) tot += __popcll(__ballot((b >> (g * PXL + e)) & 1u));
        if (lane == 0) woff[(r * G + g) * 4 + wid] = tot;
      }
    }
    __syncthreads();
    // exclusive scan of the NK counts by the first wavefront; the super-tile's aggregate goes out at once
    int total;
    if (wid == 0) {
      constexpr int PER = (NK + 63) / 64;
      int c[PER], sum = 0;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int idx = lane * PER + u;
        c[u] = idx < NK ? woff[idx] : 0;
        sum += c[u];
      }
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      int ex = incl - sum;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int idx = lane * PER + u;
        if (idx < NK) woff[idx] = ex;
        ex += c[u];
      }
      if (lane == 63) {
        woff[NK] = incl;
        __hip_atomic_store(&desc[tile], ((tile == 0 ? 2ull : 1ull) << 32) | (unsigned int)incl, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // ---- the first sub-tile's embedding values, in flight during the look-back
    V ev[2][G][ND];
    auto load_emb = [&](int r, V (&dst)[G][ND]) {
      const unsigned int sb = (unsigned int)(allbits >> (r * G * PXL)) & SMASK;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const long long i = base + (long long)r * SUB + (long long)(g * 256 + tid) * PXL;
        const bool need = i < npix && (WB || ((sb >> (g * PXL)) & GMASK) != 0u);
        if (need) {
          if (vec) {
#pragma unroll
            for (int c = 0; c < ND; ++c) dst[g][c] = *reinterpret_cast<const V*>(emb + (long long)c * npix + i);
          } else {
#pragma unroll
            for (int c = 0; c < ND; ++c)
#pragma unroll
              for (int e = 0; e < PXL; ++e) dst[g][c][e] = (i + e < npix) ? emb[(long long)c * npix + i + e] : (TIn)0;
          }
        }
      }
    };
    load_emb(0, ev[0]);
    __syncthreads();
    total = woff[NK];

    // ---- (C) look-back by the whole block: thread t inspects predecessor tile - 1 - t of the current window of 256
    int excl = 0;
    for (int hi = tile - 1; hi >= 0; hi -= 256) {
      const int j = hi - tid;
      unsigned long long d = 2ull << 32;              // tiles before the first: prefix 0
      if (j >= 0) {
        do {
          d = __hip_atomic_load(&desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((unsigned int)(d >> 32) == 0);       // predecessor has not published yet
      }
      const unsigned long long has_prefix = __ballot((unsigned int)(d >> 32) == 2u);
      const int stop = has_prefix ? __builtin_ctzll(has_prefix) : 64;      // nearest tile of this wave's 64 with a prefix
      int a = (lane <= stop) ? (int)(unsigned int)d : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if (lane == 0) { lb_sum[wid] = a; lb_stop[wid] = has_prefix ? 1 : 0; }
      __syncthreads();
      bool found = false;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (!found) {
          excl += lb_sum[w];
          found = lb_stop[w] != 0;
        }
      }
      __syncthreads();
      if (found) break;
    }
    if (tid == 0) {
      if (tile > 0)
        __hip_atomic_store(&desc[tile], (2ull << 32) | (unsigned int)(excl + total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      if (tile == nsuper - 1) *nfg_out = excl + total;
    }

    // ---- (D) sub-tile by sub-tile: coordinates added (in place for WB), points and raster indices written; the next
    // sub-tile's loads are issued before the current one is written (two register sets, used alternately)
    auto emit = [&](int r, V (&cur)[G][ND]) {
      const unsigned int sb = (unsigned int)(allbits >> (r * G * PXL)) & SMASK;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const unsigned int gb = (sb >> (g * PXL)) & GMASK;
        int bef = 0;
#pragma unroll
        for (int e = 0; e < PXL; ++e) bef += __popcll(__ballot((gb >> e) & 1u) & lower);
        const long long i = base + (long long)r * SUB + (long long)(g * 256 + tid) * PXL;
        if (i < npix && (WB || gb != 0u)) {
          int pos = excl + woff[(r * G + g) * 4 + wid] + bef;
          const unsigned int t = fdiv((unsigned int)i, dX);
          int cx = (int)((unsigned int)i - t * (unsigned int)X);
          const unsigned int z0u = fdiv(t, dY);
          int cy = (int)(t - z0u * (unsigned int)Y), cz = (int)z0u;
          V (&e4)[ND] = cur[g];
#pragma unroll
          for (int e = 0; e < PXL; ++e) {
            const int co[3] = {cx, cy, cz};
            double val[ND];
#pragma unroll
            for (int c = 0; c < ND; ++c) {
              val[c] = (double)e4[c][e] + (double)co[c];
              if (WB) e4[c][e] = (TIn)val[c];
            }
            if ((gb >> e) & 1u) {
#pragma unroll
              for (int c = 0; c < ND; ++c) Xout[(long long)pos * ND + c] = val[c];
              index[pos] = (int)(i + e);
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
      }
    };
#pragma unroll 1
    for (int r = 0; r < R; r += 2) {
      load_emb(r + 1, ev[1]);
      emit(r, ev[0]);
      if (r + 2 < R) load_emb(r + 2, ev[0]);
      emit(r + 1, ev[1]);
    }
    __syncthreads();      // woff, lb_* and the ticket slots are reused by the next round
  }
  // the workspace is handed back ZEROED by the block that finishes last (every other block has left its last look-back)
  __shared__ int s_last;
  if (tid == 0) s_last = (atomicAdd(ticket + 1, 1u) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (s_last) {
    for (int j = tid; j < nsuper; j += 256)
      __hip_atomic_store(&desc[j], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 2) __hip_atomic_store(ticket + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
template <int ND>
__global__ __launch_bounds__(256) void ms_assign_cells_kernel(
    const double* __restrict__ X, const int* __restrict__ index, int nfg, const double* __restrict__ cs,
    const int* __restrict__ order, const int* __restrict__ cell_start, double ox, double oy,
    double oz, double h, int nx, int ny, int nz, int* __restrict__ labels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nfg) return;
  double x[ND];
  if (ND == 2) {
    const f64x2 p = *reinterpret_cast<const f64x2*>(X + (long long)i * 2);
    x[0] = p[0]; x[1] = p[1];
  } else {
#pragma unroll
    for (int c = 0; c < ND; ++c) x[c] = X[(long long)i * ND + c];
  }
  const int dst = index[i];
  const double inv = 1.0 / h;
  const int cx = min(max((int)floor((x[0] - ox) * inv), 0), nx - 1);
  const int cy = min(max((int)floor((x[1] - oy) * inv), 0), ny - 1);
  const int cz = (ND == 3) ? min(max((int)floor((x[2] - oz) * inv), 0), nz - 1) : 0;
  double best = 0.0;
  int arg = -1;
  auto candidate = [&](int j) {
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
  };
  for (int r = 1;; r *= 2) {
    const int x0 = max(cx - r, 0), x1 = min(cx + r, nx - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, ny - 1);
    const int z0 = (ND == 3) ? max(cz - r, 0) : 0, z1 = (ND == 3) ? min(cz + r, nz - 1) : 0;
    arg = -1;
    if (ND == 2 && r == 1) {
      int lo[3], hi[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int yy = y0 + t;
        const bool ok = yy <= y1;
        const long long row = (long long)(ok ? yy : y0) * nx;
        const int a = cell_start[row + x0], b = cell_start[row + x1 + 1];
        lo[t] = a;
        hi[t] = ok ? b : a;
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
        for (int j = lo[t]; j < hi[t]; ++j) candidate(j);
    } else {
      for (int zz = z0; zz <= z1; ++zz)
        for (int yy = y0; yy <= y1; ++yy) {
          const long long row = ((long long)zz * ny + yy) * nx;
          const int lo = cell_start[row + x0], hi = cell_start[row + x1 + 1];
          for (int j = lo; j < hi; ++j) candidate(j);
        }
    }
    const bool whole = x0 == 0 && x1 == nx - 1 && y0 == 0 && y1 == ny - 1 && z0 == 0 && z1 == nz - 1;
    const double reach = (double)r * h;
    if (whole || (arg >= 0 && best < reach * reach)) break;
  }
  labels[dst] = arg + 1;
}

// (Two pixels per thread — both points, both pixels' row ranges, then the candidates — measured the same: 22.4
// against 21.7 us at 4096^2, 97.6 against 97.8 at 8192^2.  At full occupancy the kernel moves 3.9 TB/s of real traffic:
// points, indices, and label runs of ~24 pixels that fill their 128-byte lines partly.)

}  // namespace

static int prep_pairs() {      // pairs of pixels per thread: 8 (4096-pixel tiles, measured 0.74 vs 0.82 ms at
  static const int env = getenv("CLX_MS_PREP_K") ? atoi(getenv("CLX_MS_PREP_K")) : 8;             // 8192^2), 4 or 16
  return env == 4 ? 4 : env == 16 ? 16 : 8;
}


template <int ND, typename TIn, int G, int R, bool WB, int BLOCKS>
static int launch_compact(TIn* emb, const TIn* std, double threshold, int Z, int Y, int X, double* Xout, int* index,
                          int* nfg_out, void* workspace, hipStream_t st) {
  constexpr int PXL = 16 / (int)sizeof(TIn);
  constexpr long long SUPER = 256ll * G * PXL * R;
  const long long npix = (long long)Z * Y * X;
  const int nsuper = (int)((npix + SUPER - 1) / SUPER);
  unsigned int* ticket = (unsigned int*)workspace;
  unsigned long long* desc = (unsigned long long*)workspace + 1;
  // every channel plane starts at a multiple of npix elements: 16-byte accesses need npix % PXL == 0 and aligned bases
  const int vec = (npix % PXL == 0) && (((uintptr_t)emb | (uintptr_t)std) & 15) == 0 ? 1 : 0;
  const FastDiv dX = make_fastdiv((uint32_t)X), dY = make_fastdiv((uint32_t)Y);
  static const int cus = [] {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount;
  }();
  static const int per_cu = [] {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ms_compact_kernel<ND, TIn, G, R, WB, BLOCKS>, 256, 0) != hipSuccess || nb < 1)
      nb = 2;
    return nb;
  }();
  // persistent blocks: as many as are resident at once, not more than there are super-tiles
  const int nblocks = nsuper < cus * per_cu ? nsuper : cus * per_cu;
  CLX_LAUNCH_KIND(CLX_PROF_MS_PREPARE, (ms_compact_kernel<ND, TIn, G, R, WB, BLOCKS>), dim3(nblocks), dim3(256), 0, st, emb, std,
                  threshold, dX, dY, Y, X, npix, vec, nsuper, ticket, desc, Xout, index, nfg_out);
  return CLX_OK;
}


extern "C" size_t clx_ms_prepare_workspace(long long npix) {
  const long long ntiles = (npix + 2048 - 1) / 2048;          // the smaller tile: enough for either
  return (size_t)(ntiles + 2) * sizeof(unsigned long long);
}

extern "C" int clx_ms_prepare(double* emb, const double* std, double threshold, int ND,
                              int Z, int Y, int X, double* Xout, int* index, int* nfg_out,
                              void* workspace, clx_stream stream) {
  CLX_REQUIRE(emb && std && Xout && index && nfg_out && workspace, "clx_ms_prepare: null pointer");
  CLX_REQUIRE((ND == 2 || ND == 3) && Z > 0 && Y > 0 && X > 0, "clx_ms_prepare: bad extents");
  CLX_REQUIRE(ND == 3 || Z == 1, "clx_ms_prepare: Z must be 1 for 2-D data");
  CLX_REQUIRE(((uintptr_t)workspace & 7) == 0, "clx_ms_prepare: workspace must be 8-byte aligned");
  const long long npix = (long long)Z * Y * X;
  CLX_REQUIRE(npix < (1ll << 31), "clx_ms_prepare: too many pixels");
  hipStream_t st = (hipStream_t)stream;
  const int kp = (ND == 3 && prep_pairs() == 16) ? 8 : prep_pairs();
  const int ntiles = (int)((npix + 512 * kp - 1) / (512 * kp));
  unsigned int* ticket = (unsigned int*)workspace;
  unsigned long long* desc = (unsigned long long*)workspace + 1;
  const int vec = (npix % 2 == 0) && (((uintptr_t)emb | (uintptr_t)std) & 15) == 0 ? 1 : 0;
  const FastDiv dX = make_fastdiv((uint32_t)X), dY = make_fastdiv((uint32_t)Y);
  // persistent blocks: as many as are resident at once, not more than there are tiles
  static const int cus = [] {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount;
  }();
#define CLX_PREP(ND_, KP_)                                                                                   \
  do {                                                                                                       \
    static const int per_cu = [] {                                                                           \
      int nb = 0;                                                                                            \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ms_prepare_kernel<ND_, KP_>, 256, 0) != hipSuccess || nb < 1) \
        nb = 2;                                                                                              \
      return nb;                                                                                             \
    }();                                                                                                     \
    const int nblocks = ntiles < cus * per_cu ? ntiles : cus * per_cu;                                       \
    CLX_LAUNCH_KIND(CLX_PROF_MS_PREPARE, (ms_prepare_kernel<ND_, KP_>), dim3(nblocks), dim3(256), 0, st, emb, std, threshold, dX, dY, Y, X, npix, vec, ntiles, ticket, \
                                                         desc, Xout, index, nfg_out);                        \
  } while (0)
  if (ND == 2) { if (kp == 16) CLX_PREP(2, 16); else if (kp == 8) CLX_PREP(2, 8); else CLX_PREP(2, 4); }
  else         { if (kp >= 8) CLX_PREP(3, 8); else CLX_PREP(3, 4); }
#undef CLX_PREP
  CLX_CHECK_LAUNCH("clx_ms_prepare");
  return CLX_OK;
}

extern "C" int clx_ms_prepare_f32(const float* emb, const float* std, double threshold, int ND,
                                  int Z, int Y, int X, double* Xout, int* index, int* nfg_out,
                                  void* workspace, clx_stream stream) {
  CLX_REQUIRE(emb && std && Xout && index && nfg_out && workspace, "clx_ms_prepare_f32: null pointer");
  CLX_REQUIRE((ND == 2 || ND == 3) && Z > 0 && Y > 0 && X > 0, "clx_ms_prepare_f32: bad extents");
  CLX_REQUIRE(ND == 3 || Z == 1, "clx_ms_prepare_f32: Z must be 1 for 2-D data");
  CLX_REQUIRE(((uintptr_t)workspace & 7) == 0, "clx_ms_prepare_f32: workspace must be 8-byte aligned");
  const long long npix = (long long)Z * Y * X;
  CLX_REQUIRE(npix < (1ll << 31), "clx_ms_prepare_f32: too many pixels");
  // super-tiles of 4 sub-tiles x 4096 pixels (4 groups of four float32 pixels per thread); clx_ms_prepare_workspace(npix)
  // (one descriptor per 2048 pixels) covers them
  float* e = const_cast<float*>(emb);
  if (ND == 2) launch_compact<2, float, 4, 4, false, 3>(e, std, threshold, Z, Y, X, Xout, index, nfg_out, workspace, (hipStream_t)stream);
  else launch_compact<3, float, 4, 4, false, 2>(e, std, threshold, Z, Y, X, Xout, index, nfg_out, workspace, (hipStream_t)stream);
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
  CLX_REQUIRE(ND == 3 || ((((uintptr_t)X | (uintptr_t)centers_sorted) & 15) == 0), "clx_ms_assign_cells: 16-byte alignment");
  if (nfg == 0) return CLX_OK;
  const int grid = (nfg + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  if (ND == 2)
    CLX_LAUNCH_KIND(CLX_PROF_MS_ASSIGN, (ms_assign_cells_kernel<2>), dim3(grid), dim3(256), 0, st, X, index, nfg, centers_sorted, order, cell_start, origin[0],
                                                     origin[1], 0.0, cell, nx, ny, nz, labels);
  else
    CLX_LAUNCH_KIND(CLX_PROF_MS_ASSIGN, (ms_assign_cells_kernel<3>), dim3(grid), dim3(256), 0, st, X, index, nfg, centers_sorted, order, cell_start, origin[0],
                                                     origin[1], origin[2], cell, nx, ny, nz, labels);
  CLX_CHECK_LAUNCH("clx_ms_assign_cells");
  return CLX_OK;
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
