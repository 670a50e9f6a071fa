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

// Raster-order numbering of the surviving roots in ONE pass over L: every block takes the next
// tile of NUM_TILE pixels (atomic ticket), counts its surviving roots, learns how many precede its
// tile by decoupled look-back over the predecessors' published counts / prefixes (one 64-bit word
// each — status in the high half, value in the low half — so relaxed atomics suffice), and numbers
// its own: surviving root r gets id = 1 + (number of surviving roots before r), stored as -id
// in size[r].
constexpr int NUM_TILE = 8192;    // big tiles: the look-back costs per tile, the scan of L does not

__global__ __launch_bounds__(256) void cc_number_roots(const int* __restrict__ L, int* size, int min_size,
                                                       long long npix, int ntiles, unsigned int* __restrict__ ticket,
                                                       unsigned long long* __restrict__ desc,
                                                       int* __restrict__ total_out) {
  __shared__ int s_tile, s_excl;
  __shared__ int wcount[NUM_TILE / 256][4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) s_tile = (int)atomicAdd(ticket, 1u);
  __syncthreads();
  const int tile = s_tile;
  const long long base = (long long)tile * NUM_TILE;
  const unsigned long long lower = (1ull << lane) - 1ull;
  constexpr int K = NUM_TILE / 256;
  bool root[K];
  int before[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const long long i = base + k * 256 + tid;
    root[k] = i < npix && L[i] == (int)i && size[i] >= min_size;
    const unsigned long long ball = __ballot(root[k]);
    before[k] = __popcll(ball & lower);
    if (lane == 0) wcount[k][wid] = __popcll(ball);
  }
  __syncthreads();
  int total = 0, mine[K];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w == wid) mine[k] = total;
      total += wcount[k][w];
    }
  if (wid == 0) {
    if (lane == 0)
      __hip_atomic_store(&desc[tile], ((tile == 0 ? 2ull : 1ull) << 32) | (unsigned int)total, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    int excl = 0;
    for (int hi = tile - 1; hi >= 0; hi -= 64) {
      const int j = hi - lane;
      unsigned long long d = 2ull << 32;
      if (j >= 0) {
        do {
          d = __hip_atomic_load(&desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((unsigned int)(d >> 32) == 0);
      }
      const unsigned long long has_prefix = __ballot((unsigned int)(d >> 32) == 2u);
      const int stop = has_prefix ? __builtin_ctzll(has_prefix) : 64;
      int v = (lane <= stop) ? (int)(unsigned int)d : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      excl += v;
      if (has_prefix) break;
    }
    if (lane == 0) {
      if (tile > 0)
        __hip_atomic_store(&desc[tile], (2ull << 32) | (unsigned int)(excl + total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      s_excl = excl;
      if (tile == ntiles - 1 && total_out) *total_out = excl + total;
    }
  }
  __syncthreads();
  const int excl = s_excl;
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (root[k]) size[base + k * 256 + tid] = -(excl + mine[k] + before[k] + 1);
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

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" size_t clx_cc_workspace(long long npix) {
  const long long ntiles = (npix + NUM_TILE - 1) / NUM_TILE;
  // L and size (npix ints each), then — 8-byte aligned — the ticket and the look-back descriptors
  return (size_t)(2 * npix + (npix & 1)) * sizeof(int) + (size_t)(ntiles + 2) * sizeof(unsigned long long);
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
  CLX_REQUIRE(((uintptr_t)workspace & 7) == 0, "clx_cc_label_filter: workspace must be 8-byte aligned");
  unsigned long long* lb = (unsigned long long*)(size + npix + (npix & 1));
  unsigned int* ticket = (unsigned int*)lb;
  unsigned long long* desc = lb + 1;
  const int ntiles = (int)((npix + NUM_TILE - 1) / NUM_TILE);
  const int grid = grid_for(npix, 256);
  hipStream_t st = (hipStream_t)stream;
  if (min_size < 1) min_size = 1;   // every component has >= 1 pixel: keep all
  const int nseg = (X + 63) / 64;
  const long long nwaves = (long long)Z * Y * nseg;
  const int wgrid = grid_for(nwaves * 64, 256);
  cc_init_runs<<<wgrid, 256, 0, st>>>(seg, L, size, X, nseg, nwaves);
  cc_merge_runs<<<wgrid, 256, 0, st>>>(seg, L, Z, Y, X, nseg, nwaves);
  cc_flatten_count<<<wgrid, 256, 0, st>>>(seg, L, size, X, nseg, nwaves);
  if (hipMemsetAsync(lb, 0, (size_t)(ntiles + 1) * sizeof(unsigned long long), st) != hipSuccess) {
    clx_set_error("clx_cc_label_filter: memset failed");
    return CLX_ERR_LAUNCH;
  }
  cc_number_roots<<<ntiles, 256, 0, st>>>(L, size, min_size, npix, ntiles, ticket, desc, ncomp_out);
  cc_write<<<grid, 256, 0, st>>>(seg, L, size, out, npix);
  CLX_CHECK_LAUNCH("clx_cc_label_filter");
  return CLX_OK;
}
