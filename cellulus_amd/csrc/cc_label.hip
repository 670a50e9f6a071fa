// Connected-component labelling of an integer label image with full
// connectivity (8 in 2-D, 26 in 3-D), equal-value regions, background 0,
// followed by the size filter and raster-order renumbering — bit-exact with
//   size_filter(seg, min_size) = label(seg with small components zeroed)
// of cellulus/utils/misc.py:11-25 (skimage.measure.label semantics: ids are
// assigned in raster order of each component's first pixel).
//
// Lock-free union-find: every pixel unions with its equal-valued "backward"
// neighbours; links always point from the larger root to the smaller, so the
// final root of a component is its smallest raster index.  Stale reads of the
// parent array are harmless (parents only decrease along a chain; progress is
// made by the values returned from the atomics).
#include <stdlib.h>
#include "clx_common.h"

namespace {


__device__ __forceinline__ int uf_find(const int* L, int a) {
  int p = L[a];
  while (p != a) { a = p; p = L[a]; }
  return a;
}

// find with path halving: every visited node is re-pointed at its grandparent.  A plain store is
// enough: only non-roots are written (a root is changed by the atomicMin of a union alone), and any
// ancestor is a valid parent, so a lost or stale write merely leaves a longer path.
__device__ __forceinline__ int uf_find_halve(int* L, int a) {
  int p = L[a];
  while (p != a) {
    const int gp = L[p];
    if (gp != p) L[a] = gp;
    a = p;
    p = gp;
  }
  return a;
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  bool done;
  do {
    a = uf_find_halve(L, a);
    b = uf_find_halve(L, b);
    if (a < b) {
      const int old = atomicMin(&L[b], a);
      done = (old == b);
      b = old;
    } else if (b < a) {
      const int old = atomicMin(&L[a], b);
      done = (old == a);
      a = old;
    } else {
      done = true;
    }
  } while (!done);
}

// Wave w of the grid owns 64 consecutive pixels of one row ("segment").  Returns the pixel index
// of lane 0, the row's first pixel index and x of this lane; valid = inside the row.
struct SegPos { long long i; int x; bool valid; };
__device__ __forceinline__ SegPos seg_pos(long long w, int nseg, int X, int lane) {
  const long long row = w / nseg;
  const int sg = (int)(w - row * nseg);
  SegPos p;
  p.x = sg * 64 + lane;
  p.valid = p.x < X;
  p.i = row * X + p.x;
  return p;
}

// lane of the first pixel of this lane's horizontal run of equal non-zero values inside the
// segment (background lanes: themselves)
__device__ __forceinline__ int run_start_lane(int v, int lane, unsigned long long* starts_out) {
  const int vl = __shfl_up(v, 1, 64);
  const bool same_left = lane > 0 && v != 0 && vl == v;
  const unsigned long long starts = __ballot(!same_left);
  if (starts_out) *starts_out = starts;
  const unsigned long long upto = starts & ((2ull << lane) - 1ull);
  return 63 - __builtin_clzll(upto);
}

// pass 1: L[i] = first pixel of i's run inside its 64-pixel segment (background: -1).  Runs,
// not pixels, are what the union-find links: a 32-pixel-wide object costs one union per row
// instead of ~100 (every pixel with each of its 4 / 13 backward neighbours).
__global__ __launch_bounds__(256) void cc_init_runs(const int* __restrict__ seg, int* __restrict__ L,
                                                    int* __restrict__ size, int X, int nseg,
                                                    long long nwaves) {
  const int lane = threadIdx.x & 63;
  for (long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwaves;
       w += ((long long)gridDim.x * blockDim.x) >> 6) {
    const SegPos p = seg_pos(w, nseg, X, lane);
    const int v = p.valid ? seg[p.i] : 0;
    const int s = run_start_lane(v, lane, nullptr);
    if (p.valid) {
      L[p.i] = v ? (int)(p.i - lane + s) : -1;
      if (s == lane) size[p.i] = 0;        // sizes live at run starts (the only possible roots)
    }
  }
}

// pass 2: link the runs.  For pixel (x, y) and a preceding row r (2-D: y-1; 3-D also the three
// rows of slice z-1) with a = same(r, x-1), b = same(r, x), c = same(r, x+1):
//   b and not (left pixel in my run and a)  -> union with (r, x)    [else the left pixel did it]
//   a and not b and no left pixel in my run -> union with (r, x-1)
//   c and not b and no right pixel in my run-> union with (r, x+1)  [else the right pixel does]
// plus the run continuation across a segment boundary.  Every adjacency is covered by a chain
// of these unions; nothing else touches an atomic.
__global__ __launch_bounds__(256) void cc_merge_runs(const int* __restrict__ seg, int* L, int Z, int Y,
                                                     int X, int nseg, long long nwaves) {
  const int lane = threadIdx.x & 63;
  for (long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwaves;
       w += ((long long)gridDim.x * blockDim.x) >> 6) {
    const SegPos p = seg_pos(w, nseg, X, lane);
    const int v = p.valid ? seg[p.i] : 0;
    if (v == 0) continue;
    const int x = p.x;
    const long long row = (p.i - x) / X;
    const int y = (int)(row % Y), z = (int)(row / Y);
    const bool left = x > 0 && seg[p.i - 1] == v;
    const bool right = x + 1 < X && seg[p.i + 1] == v;
    if (lane == 0 && left) uf_union(L, (int)p.i, (int)p.i - 1);
    for (int k = 0; k < 4; ++k) {
      // k = 0: (z, y-1); 1..3: (z-1, y-1 .. y+1)
      const int zz = (k == 0) ? z : z - 1;
      const int yy = (k == 0) ? y - 1 : y + (k - 2);
      if (zz < 0 || yy < 0 || yy >= Y) continue;
      const long long r = ((long long)zz * Y + yy) * X + x;
      const bool b = seg[r] == v;
      const bool a = x > 0 && seg[r - 1] == v;
      const bool c = x + 1 < X && seg[r + 1] == v;
      if (b && !(left && a)) uf_union(L, (int)p.i, (int)r);
      if (a && !b && !left) uf_union(L, (int)p.i, (int)r - 1);
      if (c && !b && !right) uf_union(L, (int)p.i, (int)r + 1);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 2-D fast path for passes 1 + 2: a wavefront walks a strip of STRIP_ROWS rows of its 64-pixel
// column segment TOP-DOWN, carrying the previous row's values and labels in registers.  A run takes
// the smallest label of the runs it touches in the row above (segmented min inside the wave), a run
// that touches nothing starts a new label (its first pixel), and a global union is needed only when
// a run touches two different labels — so inside a strip an object costs no atomic at all instead
// of one per row.  What the strips do not see — the row above a strip, the column left of a
// segment — is linked afterwards by cc_link_borders (1/32 + 1/64 of the pixels).
// ---------------------------------------------------------------------------------------------
constexpr int STRIP_ROWS = 32;

__global__ __launch_bounds__(256) void cc_strip_kernel(const int* __restrict__ seg, int* L, int* __restrict__ size,
                                                       int Y, int X, int nseg, int nstrips) {
  const int lane = threadIdx.x & 63;
  const long long nwaves = (long long)nstrips * nseg;
  for (long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwaves;
       w += ((long long)gridDim.x * blockDim.x) >> 6) {
    const int strip = (int)(w / nseg), sg = (int)(w - (long long)strip * nseg);
    const int x = sg * 64 + lane;
    const bool in_x = x < X;
    const int y0 = strip * STRIP_ROWS, y1 = min(y0 + STRIP_ROWS, Y);
    int pv = 0, pl = -1;
    for (int y = y0; y < y1; ++y) {
      const long long i = (long long)y * X + x;
      const int v = in_x ? seg[i] : 0;
      unsigned long long starts;
      const int sl = run_start_lane(v, lane, &starts);
      // labels of the (up to three) touching pixels of the row above.  If the pixel straight above
      // matches, its two neighbours belong to the same run; otherwise up-left and up-right are two
      // DIFFERENT runs (the pixel between them differs), and both labels count.
      int c[3];
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int src = min(max(lane + dx, 0), 63);
        const int nv = __shfl(pv, src, 64), nl = __shfl(pl, src, 64);
        c[dx + 1] = (v != 0 && nv == v && (lane + dx) == src) ? nl : 0x7fffffff;
      }
      if (c[1] != 0x7fffffff) c[0] = c[2] = 0x7fffffff;
      int cand = min(c[0], min(c[1], c[2]));
      // segmented inclusive min-scan towards higher lanes, then everyone takes the run's last lane
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(cand, d, 64);
        if (lane - d >= sl) cand = min(cand, t);
      }
      const unsigned long long above = (lane == 63) ? 0ull : (starts >> (lane + 1));
      const int el = above ? lane + __builtin_ctzll(above) : 63;
      const int runmin = __shfl(cand, el, 64);
      int label = -1;
      if (v != 0) {
        label = (runmin == 0x7fffffff) ? (int)(i - lane + sl) : runmin;
        if (runmin == 0x7fffffff && sl == lane) size[i] = 0;       // a new root
        L[i] = label;
      } else if (in_x) {
        L[i] = -1;
      }
      // every other label the run touches is joined to the one it took (the L entries involved were
      // written by this wavefront in earlier rows: fence; the links themselves are atomics)
      const bool j0 = c[0] != 0x7fffffff && c[0] != label, j1 = c[1] != 0x7fffffff && c[1] != label,
                 j2 = c[2] != 0x7fffffff && c[2] != label;
      if (j0 || j1 || j2) {
        __threadfence();
        if (j0) uf_union(L, c[0], label);
        if (j1) uf_union(L, c[1], label);
        if (j2) uf_union(L, c[2], label);
      }
      pv = v;
      pl = label;
    }
  }
}

// links across the borders the strips ignore: for the first row of every strip the three pixels above,
// for the first column of every segment the three pixels to the left
__global__ void cc_link_borders(const int* __restrict__ seg, int* L, int Y, int X, int nseg, int nstrips) {
  const long long n_rows = (long long)(nstrips - 1) * X;          // pixels of the strips' first rows (not strip 0)
  const long long n_cols = (long long)(nseg - 1) * Y;             // pixels of the segments' first columns
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n_rows + n_cols;
       k += (long long)gridDim.x * blockDim.x) {
    if (k < n_rows) {
      const int y = (int)(k / X + 1) * STRIP_ROWS, x = (int)(k % X);
      const long long i = (long long)y * X + x;
      const int v = seg[i];
      if (v == 0) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= X) continue;
        const long long j = i - X + dx;
        if (seg[j] == v) uf_union(L, (int)i, (int)j);
      }
    } else {
      const long long kk = k - n_rows;
      const int x = (int)(kk / Y + 1) * 64, y = (int)(kk % Y);
      const long long i = (long long)y * X + x;
      const int v = seg[i];
      if (v == 0) continue;
      for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= Y) continue;
        const long long j = (long long)yy * X + x - 1;
        if (seg[j] == v) uf_union(L, (int)i, (int)j);
      }
    }
  }
}

// pass 3: every pixel learns its root (one find per run, broadcast inside the wave) and every
// run adds its length to the root's size (one atomic per run)
__global__ __launch_bounds__(256) void cc_flatten_count(const int* __restrict__ seg, int* L, int* size,
                                                        int X, int nseg, long long nwaves) {
  const int lane = threadIdx.x & 63;
  for (long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwaves;
       w += ((long long)gridDim.x * blockDim.x) >> 6) {
    const SegPos p = seg_pos(w, nseg, X, lane);
    const int v = p.valid ? seg[p.i] : 0;
    unsigned long long starts;
    const int s = run_start_lane(v, lane, &starts);
    int root = -1;
    if (v != 0 && s == lane) {
      root = uf_find(L, (int)p.i);
      // run length = distance to the next run start (or the end of the segment)
      const unsigned long long above = (lane == 63) ? 0ull : (starts >> (lane + 1));
      const int len = above ? __builtin_ctzll(above) + 1 : 64 - lane;
      atomicAdd(&size[root], len);
    }
    root = __shfl(root, s, 64);
    if (v != 0) L[p.i] = root;    // roots keep pointing at themselves
  }
}

// Raster-order numbering of the surviving roots: count per block of SCAN_BLOCK pixels, exclusive
// scan of the (few thousand) counts by one block, numbering.  (A single-pass variant with decoupled
// look-back was measured at 344 us against 307 us for these three launches at 8192^2: with a few KB
// of work per tile the walk through the ~2000 in-flight predecessors' aggregates costs more than
// reading L twice; the larger SCAN_BLOCK is what shortens the middle launch.)
constexpr int SCAN_BLOCK = 8192;

// per block: number of surviving roots (raster order)
__global__ __launch_bounds__(256) void cc_count_roots(const int* __restrict__ seg,
                                                      const int* __restrict__ L,
                                                      const int* __restrict__ size, int min_size,
                                                      long long npix, int* __restrict__ counts) {
  __shared__ int wsum[4];
  const long long base = (long long)blockIdx.x * SCAN_BLOCK;
  int local = 0;
  for (int k = 0; k < SCAN_BLOCK / 256; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i < npix && L[i] == (int)i && size[i] >= min_size) ++local;
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(1024) void cc_scan_counts(int* __restrict__ counts, int nblocks,
                                                       int* __restrict__ total_out) {
  // exclusive scan in place: per-thread runs of `per` entries, wave scans by lane shuffles, the 16 wave
  // totals by the first wave — two block barriers (the 1024-wide Hillis-Steele scan it replaces had twenty)
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // (runs of a multiple of four entries: a thread's run is read and written in 16-byte groups, all loads in flight
  //  at once — eight dependent 4-byte round trips each way were 8 us for 8192 entries)
  const int per = ((nblocks + 1023) / 1024 + 3) & ~3;
  const int lo = min(tid * per, nblocks), hi = min(lo + per, nblocks);
  const bool wide = (((uintptr_t)counts) & 15) == 0 && hi - lo == per && per <= 16;
  int4 q[4] = {};
  int s = 0;
  if (wide) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (g * 4 < per) q[g] = *reinterpret_cast<const int4*>(counts + lo + g * 4);
#pragma unroll
    for (int g = 0; g < 4; ++g) s += q[g].x + q[g].y + q[g].z + q[g].w;
  } else {
    for (int i = lo; i < hi; ++i) s += counts[i];
  }
  int incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  if (wid == 0) {
    int w = lane < 16 ? wsum[lane] : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int v = __shfl_up(w, o, 64);
      if (lane >= o) w += v;
    }
    if (lane < 16) wsum[lane] = w;
  }
  __syncthreads();
  int run = incl - s + (wid ? wsum[wid - 1] : 0);
  if (wide) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (g * 4 < per) {
        int4 o;
        o.x = run; run += q[g].x;
        o.y = run; run += q[g].y;
        o.z = run; run += q[g].z;
        o.w = run; run += q[g].w;
        *reinterpret_cast<int4*>(counts + lo + g * 4) = o;
      }
  } else {
    for (int i = lo; i < hi; ++i) {
      const int c = counts[i];
      counts[i] = run;
      run += c;
    }
  }
  if (tid == 1023 && total_out) *total_out = wsum[15];
}

// surviving root r gets id = 1 + (number of surviving roots before r); stored in size[r] as -id
__global__ __launch_bounds__(256) void cc_number_roots(const int* __restrict__ seg,
                                                       const int* __restrict__ L, int* size,
                                                       int min_size, long long npix,
                                                       const int* __restrict__ offsets) {
  __shared__ int wcount[4];
  __shared__ int running;
  if (threadIdx.x == 0) running = offsets[blockIdx.x];
  __syncthreads();
  const long long base = (long long)blockIdx.x * SCAN_BLOCK;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int k = 0; k < SCAN_BLOCK / 256; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    const bool root = i < npix && L[i] == (int)i && size[i] >= min_size;
    const unsigned long long ball = __ballot(root);
    const int before = __popcll(ball & ((1ull << lane) - 1ull));
    if (lane == 0) wcount[wid] = __popcll(ball);
    __syncthreads();
    int woff = running;
    for (int w = 0; w < wid; ++w) woff += wcount[w];
    if (root) size[i] = -(woff + before + 1);
    __syncthreads();
    if (threadIdx.x == 0) running += wcount[0] + wcount[1] + wcount[2] + wcount[3];
    __syncthreads();
  }
}

__global__ void cc_write(const int* __restrict__ seg, const int* __restrict__ L,
                         const int* __restrict__ size, int* __restrict__ out, long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    int v = 0;
    const int r = L[i];
    if (r >= 0) {
      const int s = size[r];
      v = (s < 0) ? -s : 0;   // positive sizes = removed (too small) components
    }
    out[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// 2-D path, round 3: four byte-moving passes instead of nine.  The union-find parent array IS the
// output image (entries = parent index + 1, background 0 — so background pixels are written once
// and never touched again), and everything between the strip pass and the final rewrite works on
// the (few thousand) strip-local LABELS instead of on pixels:
//   1. cc_strip1   seg -> out (label + 1 | 0); every run adds its length to its provisional label's
//                  size; per (row, segment) one 64-bit mask of the labels the row creates and one of its
//                  foreground pixels; rows without foreground take a wave-uniform branch around the lane work
//   2. cc_link1    unions across strip / segment borders, one per pair of touching runs
//   3. cc_fold     per label: root = find(label); size[root] += size[label]; out[label] = root + 1
//   4. cc_mark     per label: surviving roots set their bit in a pixel bitmap
//   5. cc_word_prefix  survivors in front of each 32-pixel bitmap word inside its chunk (>= 2048 pixels) + per chunk
//   6. cc_scan_counts over the (<= 8192) chunks
//   7. cc_rank     per label: id of its root = chunk prefix + word prefix + popcount inside the word + 1, 0 if
//                  removed; stored in size[label]
//   8. cc_rewrite_masked  out[i] = out[i] ? size[out[i] - 1] : 0, in place, 16 bytes per lane; groups without
//                  foreground (by the masks of pass 1) are neither read nor written
// Bytes per pixel: 4 read + 4 written (1), 4 read + 4 written at foreground groups only (8); passes 2-7
// touch borders and labels (a few microseconds each).  Round 2 read or wrote every pixel nine times
// (36 B per pixel).
// Round 4, measured at 4096^2 (PMC: profiles/r04_cc_pmc.txt): pass 1 is bound by instruction ISSUE, not by bytes — its
// access pattern alone (load a row, store it) runs in 23 us, the pass in 48: 71 vector + 45 scalar instructions per
// row of 64 pixels, one instruction per wavefront per four cycles, 256 rows per SIMD.  A tile form of the pass (all
// rows of a 16- or 32-row tile loaded at once, runs linked through a union-find in LDS, one size store per root
// instead of an atomic per run) was written and measured: the same instruction count per row, lower occupancy
// (LDS + 130 registers), 63 / 89 us — dropped.  What did pay: run-based border links (cc_link1 43 -> 10 us: a
// 26-pixel-wide object crossing a border cost 78 unions for one useful link), word prefixes for the ranks
// (cc_rank 17 -> 5 + 4.6 us), 16-byte accesses in the scan (8 -> 4.4 us).
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ int uf1_find(const int* L, int a) {
  int p = L[a] - 1;
  while (p != a) { a = p; p = L[a] - 1; }
  return a;
}
__device__ __forceinline__ int uf1_find_halve(int* L, int a) {
  int p = L[a] - 1;
  while (p != a) {
    const int gp = L[p] - 1;
    if (gp != p) L[a] = gp + 1;
    a = p;
    p = gp;
  }
  return a;
}
__device__ __forceinline__ void uf1_union(int* L, int a, int b) {
  bool done;
  do {
    a = uf1_find_halve(L, a);
    b = uf1_find_halve(L, b);
    if (a < b) {
      const int old = atomicMin(&L[b], a + 1) - 1;
      done = (old == b);
      b = old;
    } else if (b < a) {
      const int old = atomicMin(&L[a], b + 1) - 1;
      done = (old == a);
      a = old;
    } else {
      done = true;
    }
  } while (!done);
}

// whole-wave shifts by one lane as DPP moves (no LDS crossbar trip): lane i takes lane i -/+ 1, the end lane `fill`
__device__ __forceinline__ int wave_shr1(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_shl1(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x130, 0xf, 0xf, false); }

__global__ __launch_bounds__(256) void cc_strip1(const int* __restrict__ seg, int* L, int* size,
                                                 unsigned long long* __restrict__ labelmask,
                                                 unsigned long long* __restrict__ fgmask, int* __restrict__ colL,
                                                 int* __restrict__ colR, int Y, int X, int nseg, int nstrips) {
  __shared__ int runlab[4][64];
  const int lane = threadIdx.x & 63;
  int* rl = runlab[threadIdx.x >> 6];
  constexpr int INF = 0x7fffffff;
  const long long nwaves = (long long)nstrips * nseg;
  for (long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwaves;
       w += ((long long)gridDim.x * blockDim.x) >> 6) {
    const int strip = (int)(w / nseg), sg = (int)(w - (long long)strip * nseg);
    const int x = sg * 64 + lane;
    const bool in_x = x < X;
    const int xc = min(x, X - 1);                           // loads are unconditional (clamped), values masked after
    const int y0 = strip * STRIP_ROWS, y1 = min(y0 + STRIP_ROWS, Y);
    int pv = 0, pl = -1;
    int nx[4];                                              // the next four rows: loaded while the current four are walked
#pragma unroll
    for (int k = 0; k < 4; ++k) nx[k] = seg[min(y0 + k, Y - 1) * X + xc];      // (npix < 2^31: 32-bit indices)
    for (int yb = y0; yb < y1; yb += 4) {
      int vv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) vv[k] = (in_x && yb + k < y1) ? nx[k] : 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) nx[k] = seg[min(yb + 4 + k, Y - 1) * X + xc];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int y = yb + k;
        if (y >= y1) break;
        const int i = y * X + x;
        const int mrow = y * nseg + sg, crow = sg * Y + y;
        const int v = vv[k];
        // The pass is bound by instruction issue (a SIMD issues one wavefront instruction per four cycles, 256 rows
        // of 64 pixels per SIMD at 4096^2: a hundred instructions per row are 43 us), and a third to a half of the
        // rows of a cell image hold no foreground in a 64-pixel segment: those take this wave-uniform branch.
        const unsigned long long fgrow = __builtin_amdgcn_ballot_w64(v != 0);
        if (fgrow == 0ull) {
          if (in_x) L[i] = 0;
          if (lane == 0) {
            fgmask[mrow] = 0ull;
            labelmask[mrow] = 0ull;
          }
          if (colL != nullptr) {
            if (lane == 0) colL[crow] = 0;
            if (lane == 63) colR[crow] = 0;
          }
          pv = 0;
          pl = -1;
          continue;
        }
        // runs of equal non-zero values inside the segment (background lanes: runs of their own)
        // (every DPP move is issued with all lanes active, before any lane-dependent condition: a move under a
        //  partial EXEC mask treats the inactive source lanes as out of range)
        const int v_left = wave_shr1(v, 0), v_right = wave_shl1(v, 0);
        const int pv_left = wave_shr1(pv, 0), pl_left = wave_shr1(pl, INF);
        const int pv_right = wave_shl1(pv, 0), pl_right = wave_shl1(pl, INF);
        const bool same_left = v != 0 && v_left == v;
        const unsigned long long starts = __builtin_amdgcn_ballot_w64(!same_left);
        const int sl = 63 - __builtin_clzll(starts & ((2ull << lane) - 1ull));
        // labels of the (up to three) touching pixels of the row above.  If the pixel straight above
        // matches, its two neighbours belong to the same run; otherwise up-left and up-right are two
        // DIFFERENT runs (the pixel between them differs), and both labels count.
        int c0 = (v != 0 && pv_left == v) ? pl_left : INF;
        const int c1 = (v != 0 && pv == v) ? pl : INF;
        int c2 = (v != 0 && pv_right == v) ? pl_right : INF;
        if (c1 != INF) c0 = c2 = INF;
        const int cand = min(c0, min(c1, c2));
        // the run's smallest candidate through one LDS word per run (the wave's LDS operations execute in order):
        // the start lane resets it, touching lanes atomic-min into it, everyone reads it back
        if (sl == lane) rl[lane] = INF;
        if (cand != INF) atomicMin(&rl[sl], cand);
        const int runmin = rl[sl];
        const bool fresh = runmin == INF;
        const int label = v != 0 ? (fresh ? i - lane + sl : runmin) : -1;
        if (in_x) L[i] = label + 1;
        if (v != 0 && v_right != v) {                       // once per run, at its last pixel: its length goes to its label
          const int len = lane - sl + 1;
          if (fresh) atomicExch(&size[label], len);         // creates the label (no zero-filled array needed)
          else atomicAdd(&size[label], len);
        }
        // the labels this row created, one 64-bit word per (row, segment): what the label passes iterate over
        // (a single global list would serialise its appends on one counter: 162 us of a 4096^2 image)
        const unsigned long long created = __builtin_amdgcn_ballot_w64(v != 0 && fresh && sl == lane);
        if (lane == 0) {
          fgmask[mrow] = fgrow;
          labelmask[mrow] = created;
        }
        // the segment's edge columns, transposed (contiguous in y): what the border pass compares instead of
        // column-strided reads of the image (one 4-byte value per 64-byte sector: 0.21 GB at 8192^2)
        if (colL != nullptr) {
          if (lane == 0) colL[crow] = v;
          if (lane == 63) colR[crow] = v;
        }
        const bool j0 = c0 != INF && c0 != label, j1 = c1 != INF && c1 != label, j2 = c2 != INF && c2 != label;
        if (j0 || j1 || j2) {
          __threadfence();
          if (j0) uf1_union(L, c0, label);
          if (j1) uf1_union(L, c1, label);
          if (j2) uf1_union(L, c2, label);
        }
        pv = v;
        pl = label;
      }
    }
  }
}

__global__ void cc_link1(const int* __restrict__ seg, int* L, const int* __restrict__ colL,
                         const int* __restrict__ colR, int Y, int X, int nseg, int nstrips, int strip_rows,
                         unsigned int* __restrict__ zero, long long nzero) {
  // (also clears the survivor bitmap and the chunk counters of the later label passes: a fill beside this
  //  latency-bound pass instead of a launch of its own)
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nzero; k += (long long)gridDim.x * blockDim.x)
    zero[k] = 0u;
  // One link per pair of touching runs, as inside the strips (a 26-pixel-wide object crossing a border used to cost
  // 78 unions — three per pixel, each a chain of dependent global accesses — for one useful link): the pixel straight
  // across links unless the previous pixel of the border run already did; a diagonal pixel links only where neither
  // the pixel straight across nor the neighbour along the border covers it.
  const long long n_rows = (long long)(nstrips - 1) * X;
  const long long n_cols = (long long)(nseg - 1) * Y;
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n_rows + n_cols;
       k += (long long)gridDim.x * blockDim.x) {
    if (k < n_rows) {
      const int y = (int)(k / X + 1) * strip_rows, x = (int)(k % X);
      const long long i = (long long)y * X + x;
      const int v = seg[i];
      if (v == 0) continue;
      const long long j = i - X;
      // (left / right: the neighbour along the border IN THE SAME 64 x 32 BLOCK of the strip pass — a link is left to a neighbour only if
      //  that pass has joined this pixel to it)
      const bool left = (x & 63) != 0 && seg[i - 1] == v, right = ((x + 1) & 63) != 0 && x + 1 < X && seg[i + 1] == v;
      const bool a = x > 0 && seg[j - 1] == v, b = seg[j] == v, c = x + 1 < X && seg[j + 1] == v;
      if (b && !(left && a)) uf1_union(L, (int)i, (int)j);
      if (a && !b && !left) uf1_union(L, (int)i, (int)(j - 1));
      if (c && !b && !right) uf1_union(L, (int)i, (int)(j + 1));
    } else {
      const long long kk = k - n_rows;
      const int sb = (int)(kk / Y + 1), x = sb * 64, y = (int)(kk % Y);
      const long long i = (long long)y * X + x;
      const int* mine = colL != nullptr ? colL + (long long)sb * Y : nullptr;          // this column, contiguous in y
      const int* other = colR != nullptr ? colR + (long long)(sb - 1) * Y : nullptr;   // the column left of it
      const int v = mine ? mine[y] : seg[i];
      if (v == 0) continue;
      const bool up = y % strip_rows != 0 && (mine ? mine[y - 1] : seg[i - X]) == v;
      const bool down = (y + 1) % strip_rows != 0 && y + 1 < Y && (mine ? mine[y + 1] : seg[i + X]) == v;
      const bool a = y > 0 && (other ? other[y - 1] : seg[i - X - 1]) == v;
      const bool b = (other ? other[y] : seg[i - 1]) == v;
      const bool c = y + 1 < Y && (other ? other[y + 1] : seg[i + X - 1]) == v;
      if (b && !(up && a)) uf1_union(L, (int)i, (int)(i - 1));
      if (a && !b && !up) uf1_union(L, (int)i, (int)(i - X - 1));
      if (c && !b && !down) uf1_union(L, (int)i, (int)(i + X - 1));
    }
  }
}

#define CC_FOR_EACH_LABEL(l)                                                                        \
  for (long long wd = (long long)blockIdx.x * blockDim.x + threadIdx.x; wd < nwords;                \
       wd += (long long)gridDim.x * blockDim.x)                                                     \
    for (unsigned long long m_ = labelmask[wd]; m_ != 0ull; m_ &= m_ - 1ull)                        \
      if (const int l = (int)((wd / nseg) * X + (wd % nseg) * 64 + __builtin_ctzll(m_)); true)

// (Measured and not kept: the three label passes as ONE launch with two software grid barriers —
// 137 us instead of 3 x 5 us + gaps: every block's agent-scope fence writes back / invalidates its
// XCD's L2; and the bitmap / counter zeroing folded into the strip pass — 67 -> 84 us for the strip
// kernel against a 3-us fill.)
constexpr int MAX_RANK_CHUNKS = 8192;

__global__ void cc_fold(int* L, int* size, const unsigned long long* __restrict__ labelmask, long long nwords,
                        int nseg, int X) {
  CC_FOR_EACH_LABEL(l) {
    const int r = uf1_find(L, l);
    if (r != l) {
      atomicAdd(&size[r], size[l]);     // size[l] is final (written by the strip pass only)
      L[l] = r + 1;                     // any ancestor is a valid parent; the root is the best one
    }
  }
}

__global__ void cc_mark(const int* __restrict__ L, const int* __restrict__ size,
                        const unsigned long long* __restrict__ labelmask, long long nwords, int nseg, int X,
                        int min_size, unsigned int* bitmap) {
  CC_FOR_EACH_LABEL(l) {
    if (L[l] == l + 1 && size[l] >= min_size) atomicOr(&bitmap[l >> 5], 1u << (l & 31));
  }
}

// survivors before each 32-pixel word of the bitmap inside its rank chunk + the chunk's total: one wavefront per
// chunk (cc_rank used to count the words in front of a root one by one: up to 63 loads per label, 17 us)
__global__ __launch_bounds__(256) void cc_word_prefix(const unsigned int* __restrict__ bitmap, long long nbitwords,
                                                      int words_per_chunk, int nchunks, int* __restrict__ wprefix,
                                                      int* __restrict__ chunk_count) {
  const int lane = threadIdx.x & 63;
  for (long long c = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < nchunks;
       c += ((long long)gridDim.x * blockDim.x) >> 6) {
    int running = 0;
    for (int w0 = 0; w0 < words_per_chunk; w0 += 64) {
      const long long wd = c * words_per_chunk + w0 + lane;
      const bool in = w0 + lane < words_per_chunk && wd < nbitwords;
      const int n = in ? __popc(bitmap[wd]) : 0;
      int incl = n;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      if (in) wprefix[wd] = running + incl - n;
      running += __shfl(incl, 63, 64);
    }
    if (lane == 0) chunk_count[c] = running;
  }
}

__global__ void cc_rank(const int* __restrict__ L, int* size, const unsigned long long* __restrict__ labelmask,
                        long long nwords, int nseg, int X, const unsigned int* __restrict__ bitmap,
                        const int* __restrict__ chunk_prefix, const int* __restrict__ wprefix, int rank_chunk) {
  CC_FOR_EACH_LABEL(l) {
    const int r = L[l] - 1;             // cc_fold left the root here
    int id = 0;
    const unsigned int word = bitmap[r >> 5];
    if ((word >> (r & 31)) & 1u)
      id = chunk_prefix[r / rank_chunk] + wprefix[r >> 5] + __popc(word & ((1u << (r & 31)) - 1u)) + 1;
    size[l] = id;
  }
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// the four labels of a 16-byte group through as few look-ups as they need: a group inside an object holds one label
// (the pass was bound by the texture path — 4 gathers per group, 15 us whether a group was read or skipped)
__device__ __forceinline__ i32x4 relabel4(i32x4 v, const int* __restrict__ size) {
  const int first = v[0] ? v[0] : (v[1] ? v[1] : (v[2] ? v[2] : v[3]));
  const int id = first ? size[first - 1] : 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (v[e] == first) v[e] = v[e] ? id : 0;
    else if (v[e]) v[e] = size[v[e] - 1];
  }
  return v;
}

__global__ void cc_rewrite(int* out, const int* __restrict__ size, long long npix) {
  const long long nquads = npix >> 2;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nquads; q += (long long)gridDim.x * blockDim.x) {
    i32x4 v = *reinterpret_cast<const i32x4*>(out + 4 * q);
    if ((v[0] | v[1] | v[2] | v[3]) == 0) continue;
    *reinterpret_cast<i32x4*>(out + 4 * q) = relabel4(v, size);
  }
  if (blockIdx.x == 0 && threadIdx.x < (npix & 3)) {        // the last 1-3 pixels
    const long long i = (nquads << 2) + threadIdx.x;
    const int p = out[i];
    if (p) out[i] = size[p - 1];
  }
}

// the same with the foreground masks of the first pass ([row][segment], X % 4 == 0): a 16-byte group of the label
// image is read only if one of its four pixels is foreground — the background of `out` was final after the first
// pass.  Block (bx, by): 1024 pixels of RY rows (+ multiples of the grid's rows); a thread's mask words, then its
// groups, then its look-ups are each in flight together.  (Ablation at 4096^2, 24 % foreground: masks + groups alone
// 7.8 us — two dependent round trips and the launch —, + stores 12.4, + look-ups 16.1; the unmasked pass 16.9 us but
// 64 MB instead of 20 MB read.)
__global__ __launch_bounds__(256) void cc_rewrite_masked(int* out, const int* __restrict__ size,
                                                         const unsigned long long* __restrict__ fgmask, int Y, int X,
                                                         int nseg) {
  constexpr int RY = 8;
  const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x >= X) return;
  for (int y0 = blockIdx.y * RY; y0 < Y; y0 += gridDim.y * RY) {
    bool act[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      const int y = min(y0 + r, Y - 1);
      act[r] = y0 + r < Y && ((fgmask[(long long)y * nseg + (x >> 6)] >> (x & 63)) & 15ull) != 0ull;
    }
    i32x4 v[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      v[r] = i32x4{0, 0, 0, 0};
      if (act[r]) v[r] = *reinterpret_cast<const i32x4*>(out + (long long)(y0 + r) * X + x);
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) v[r] = relabel4(v[r], size);
#pragma unroll
    for (int r = 0; r < RY; ++r)
      if (act[r]) *reinterpret_cast<i32x4*>(out + (long long)(y0 + r) * X + x) = v[r];
  }
}

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" size_t clx_cc_workspace(long long npix) {
  // the larger of the two layouts: [L | size | block counts] (3-D) and
  // [size | label masks (one 64-bit word per row and 64-pixel segment, <= npix / 32 + 2 Y ints) | bitmap | chunk counts] (2-D)
  const long long nblocks = (npix + SCAN_BLOCK - 1) / SCAN_BLOCK;
  return (size_t)(3 * npix + npix / 16 + MAX_RANK_CHUNKS + nblocks + 128) * sizeof(int);
}

extern "C" int clx_cc_label_filter(const int* seg, int* out, int Z, int Y, int X, int min_size,
                                   int* ncomp_out, void* workspace, clx_stream stream) {
  CLX_REQUIRE(seg && out && workspace, "clx_cc_label_filter: null pointer");
  CLX_REQUIRE(Z > 0 && Y > 0 && X > 0, "clx_cc_label_filter: bad extents");
  const long long npix = (long long)Z * Y * X;
  CLX_REQUIRE(npix < (1ll << 31), "clx_cc_label_filter: too many pixels");
  CLX_REQUIRE(seg != out, "clx_cc_label_filter: in-place operation is not supported");
  int* L = (int*)workspace;
  int* size = L + npix;
  int* counts = size + npix;
  const int nblocks = (int)((npix + SCAN_BLOCK - 1) / SCAN_BLOCK);
  const int grid = grid_for(npix, 256);
  hipStream_t st = (hipStream_t)stream;
  if (min_size < 1) min_size = 1;   // every component has >= 1 pixel: keep all
  const int nseg = (X + 63) / 64;
  const long long nwaves = (long long)Z * Y * nseg;
  const int wgrid = grid_for(nwaves * 64, 256);
  static const bool strips = getenv("CLX_CC_STRIPS") == nullptr || atoi(getenv("CLX_CC_STRIPS")) != 0;
  static const bool labels = getenv("CLX_CC_LABELS") == nullptr || atoi(getenv("CLX_CC_LABELS")) != 0;
  // label-list path (2-D): `out` is the parent array.  Workspace: [size | label masks | foreground masks |
  // survivor bitmap | survivors before each bitmap word | chunk counts | edge columns left, right]
  static const bool masked = getenv("CLX_CC_MASKED_REWRITE") == nullptr || atoi(getenv("CLX_CC_MASKED_REWRITE")) != 0;
  const long long nwords = (npix + 31) / 32;
  // rank chunks: >= 2048 pixels, a multiple of 32, at most MAX_RANK_CHUNKS of them
  long long rank_chunk = (npix + MAX_RANK_CHUNKS - 1) / MAX_RANK_CHUNKS;
  rank_chunk = ((rank_chunk < 2048 ? 2048 : rank_chunk) + 31) / 32 * 32;
  const int nchunks = (int)((npix + rank_chunk - 1) / rank_chunk);
  const long long nmask = (long long)Y * nseg;
  int* sz = (int*)workspace;
  unsigned long long* labelmask = (unsigned long long*)(sz + npix + (npix & 1));
  unsigned long long* fgmask = labelmask + nmask;
  unsigned int* bitmap = (unsigned int*)(fgmask + nmask);
  int* wprefix = (int*)(bitmap + nwords);
  int* chunk = wprefix + nwords;
  chunk += (4 - (((uintptr_t)chunk >> 2) & 3)) & 3;                           // 16-byte groups in cc_scan_counts
  int* colL = chunk + nchunks;                       // edge columns of the 64-pixel segments, [segment][y]
  int* colR = colL + (long long)nseg * Y;
  // (images a few pixels wide and very tall: the edge columns do not fit the workspace — the border pass then reads
  //  the image; if the masks do not fit either — one 64-bit word per row for a handful of pixels — the pixel-list path below)
  if ((size_t)((unsigned char*)(colR + (long long)nseg * Y) - (unsigned char*)workspace) > clx_cc_workspace(npix))
    colL = colR = nullptr;
  const bool fits = (size_t)((unsigned char*)(chunk + nchunks) - (unsigned char*)workspace) <= clx_cc_workspace(npix);
  if (Z == 1 && strips && labels && fits && ((uintptr_t)out & 15) == 0) {
    const int nstrips = (Y + STRIP_ROWS - 1) / STRIP_ROWS;
    CLX_LAUNCH_KIND(CLX_PROF_CC, cc_strip1, dim3(grid_for((long long)nstrips * nseg * 64, 256)), dim3(256), 0, st, seg, out, sz, labelmask, fgmask, colL, colR, Y, X, nseg, nstrips);
    const long long nb = (long long)(nstrips - 1) * X + (long long)(nseg - 1) * Y;
    CLX_LAUNCH_KIND(CLX_PROF_CC, cc_link1, dim3(grid_for(nb > nwords ? nb : nwords, 256)), dim3(256), 0, st, seg, out, colL, colR, Y, X, nseg, nstrips, STRIP_ROWS, bitmap, nwords);
    const int lgrid = grid_for(nmask, 256);
    CLX_LAUNCH_KIND(CLX_PROF_CC, cc_fold, dim3(lgrid), dim3(256), 0, st, out, sz, labelmask, nmask, nseg, X);
    CLX_LAUNCH_KIND(CLX_PROF_CC, cc_mark, dim3(lgrid), dim3(256), 0, st, out, sz, labelmask, nmask, nseg, X, min_size, bitmap);
    CLX_LAUNCH_KIND(CLX_PROF_CC, cc_word_prefix, dim3(grid_for((long long)nchunks * 64, 256)), dim3(256), 0, st, bitmap, nwords, (int)(rank_chunk / 32), nchunks, wprefix, chunk);
    CLX_LAUNCH_KIND(CLX_PROF_CC, cc_scan_counts, dim3(1), dim3(1024), 0, st, chunk, nchunks, ncomp_out);
    CLX_LAUNCH_KIND(CLX_PROF_CC, cc_rank, dim3(lgrid), dim3(256), 0, st, out, sz, labelmask, nmask, nseg, X, bitmap, chunk, wprefix, (int)rank_chunk);
    if (masked && X % 4 == 0)
      CLX_LAUNCH_KIND(CLX_PROF_CC, cc_rewrite_masked, dim3((X / 4 + 255) / 256, (Y + 7) / 8 < 32768 ? (Y + 7) / 8 : 32768), dim3(256), 0, st, out, sz, fgmask, Y, X, nseg);
    else
      CLX_LAUNCH_KIND(CLX_PROF_CC, cc_rewrite, dim3(grid_for(npix / 4 + 1, 256)), dim3(256), 0, st, out, sz, npix);
    CLX_CHECK_LAUNCH("clx_cc_label_filter");
    return CLX_OK;
  }
  if (Z == 1 && strips) {
    const int nstrips = (Y + STRIP_ROWS - 1) / STRIP_ROWS;
    CLX_LAUNCH_KIND(CLX_PROF_CC, cc_strip_kernel, dim3(grid_for((long long)nstrips * nseg * 64, 256)), dim3(256), 0, st, seg, L, size, Y, X, nseg, nstrips);
    const long long nb = (long long)(nstrips - 1) * X + (long long)(nseg - 1) * Y;
    if (nb > 0) CLX_LAUNCH_KIND(CLX_PROF_CC, cc_link_borders, dim3(grid_for(nb, 256)), dim3(256), 0, st, seg, L, Y, X, nseg, nstrips);
  } else {
    CLX_LAUNCH_KIND(CLX_PROF_CC, cc_init_runs, dim3(wgrid), dim3(256), 0, st, seg, L, size, X, nseg, nwaves);
    CLX_LAUNCH_KIND(CLX_PROF_CC, cc_merge_runs, dim3(wgrid), dim3(256), 0, st, seg, L, Z, Y, X, nseg, nwaves);
  }
  CLX_LAUNCH_KIND(CLX_PROF_CC, cc_flatten_count, dim3(wgrid), dim3(256), 0, st, seg, L, size, X, nseg, nwaves);
  CLX_LAUNCH_KIND(CLX_PROF_CC, cc_count_roots, dim3(nblocks), dim3(256), 0, st, seg, L, size, min_size, npix, counts);
  CLX_LAUNCH_KIND(CLX_PROF_CC, cc_scan_counts, dim3(1), dim3(1024), 0, st, counts, nblocks, ncomp_out);
  CLX_LAUNCH_KIND(CLX_PROF_CC, cc_number_roots, dim3(nblocks), dim3(256), 0, st, seg, L, size, min_size, npix, counts);
  CLX_LAUNCH_KIND(CLX_PROF_CC, cc_write, dim3(grid), dim3(256), 0, st, seg, L, size, out, npix);
  CLX_CHECK_LAUNCH("clx_cc_label_filter");
  return CLX_OK;
}
