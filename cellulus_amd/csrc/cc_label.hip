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

constexpr int SCAN_BLOCK = 1024;

__device__ __forceinline__ int uf_find(const int* L, int a) {
  int p = L[a];
  while (p != a) { a = p; p = L[a]; }
  return a;
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  bool done;
  do {
    a = uf_find(L, a);
    b = uf_find(L, b);
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

__global__ void cc_init(const int* __restrict__ seg, int* __restrict__ L, int* __restrict__ size,
                        long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    L[i] = (int)i;
    size[i] = 0;
  }
}

__global__ void cc_merge(const int* __restrict__ seg, int* L, int Z, int Y, int X, long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    const int v = seg[i];
    if (v == 0) continue;
    const int x = (int)(i % X);
    const long long t = i / X;
    const int y = (int)(t % Y);
    const int z = (int)(t / Y);
    // the 13 (3-D) / 4 (2-D) neighbours that precede i in raster order
    for (int dz = -1; dz <= 0; ++dz) {
      const int zz = z + dz;
      if (zz < 0) continue;
      for (int dy = -1; dy <= 1; ++dy) {
        if (dz == 0 && dy > 0) break;
        const int yy = y + dy;
        if (yy < 0 || yy >= Y) continue;
        for (int dx = -1; dx <= 1; ++dx) {
          if (dz == 0 && dy == 0 && dx >= 0) break;
          const int xx = x + dx;
          if (xx < 0 || xx >= X) continue;
          const long long j = ((long long)zz * Y + yy) * X + xx;
          if (seg[j] == v) uf_union(L, (int)i, (int)j);
        }
      }
    }
  }
}

__global__ void cc_flatten_count(const int* __restrict__ seg, int* L, int* size, long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    if (seg[i] == 0) continue;
    const int r = uf_find(L, (int)i);
    L[i] = r;   // only shortens i's own chain; roots are never rewritten
    atomicAdd(&size[r], 1);
  }
}

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
    if (i < npix && seg[i] != 0 && L[i] == (int)i && size[i] >= min_size) ++local;
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(1024) void cc_scan_counts(int* __restrict__ counts, int nblocks,
                                                       int* __restrict__ total_out) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (nblocks + 1023) / 1024;
  const int lo = tid * per, hi = min(lo + per, nblocks);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += counts[i];
  part[tid] = s;
  __syncthreads();
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
    const bool root = i < npix && seg[i] != 0 && L[i] == (int)i && size[i] >= min_size;
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
    if (seg[i] != 0) {
      const int s = size[L[i]];
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
  const long long nblocks = (npix + SCAN_BLOCK - 1) / SCAN_BLOCK;
  return (size_t)(2 * npix + nblocks + 1) * sizeof(int);
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
  cc_init<<<grid, 256, 0, st>>>(seg, L, size, npix);
  cc_merge<<<grid, 256, 0, st>>>(seg, L, Z, Y, X, npix);
  cc_flatten_count<<<grid, 256, 0, st>>>(seg, L, size, npix);
  cc_count_roots<<<nblocks, 256, 0, st>>>(seg, L, size, min_size, npix, counts);
  cc_scan_counts<<<1, 1024, 0, st>>>(counts, nblocks, ncomp_out);
  cc_number_roots<<<nblocks, 256, 0, st>>>(seg, L, size, min_size, npix, counts);
  cc_write<<<grid, 256, 0, st>>>(seg, L, size, out, npix);
  CLX_CHECK_LAUNCH("clx_cc_label_filter");
  return CLX_OK;
}
