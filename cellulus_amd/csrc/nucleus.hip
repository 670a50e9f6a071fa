// "nucleus" post-processing (cellulus/segment.py:52-101): for every instance id
//   threshold = otsu(raw[seg == id]);  mask = (seg == id) & (raw > threshold)
//   mask[bbox] = binary_fill_holes(mask[bbox]);  out[mask] = id      (ascending id)
// as three passes over the whole label image instead of a Python loop over ids:
//   1. clx_inst_stats      bounding box + min/max raw value of every id (integer atomics on
//                          order-preserving keys, so float min/max need no CAS loop)
//   2. clx_inst_histogram  all per-instance Otsu histograms at once; numpy.histogram's index
//                          arithmetic is reproduced in the raw dtype (f32 stays f32), integer
//                          images get one bin per value as skimage does
//   3. clx_inst_refine     one workgroup per instance: threshold, flood the background of the
//                          bounding box from its border (face connectivity = scipy's default
//                          structure), everything not reached is instance; atomicMax into the
//                          output reproduces "later ids overwrite earlier ones"
// The 256-bin between-class-variance scan per instance stays on the host (numpy, a few µs).
#include "clx_common.h"

namespace {

enum { RAW_F32 = 0, RAW_F64 = 1, RAW_I32 = 2 };

__device__ __forceinline__ unsigned long long order_key(float v) {
  const unsigned int b = __float_as_uint(v);
  return (unsigned long long)((b & 0x80000000u) ? ~b : (b | 0x80000000u));
}
__device__ __forceinline__ unsigned long long order_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ unsigned long long order_key(int v) {
  return (unsigned long long)((unsigned int)v ^ 0x80000000u);
}

__global__ void stats_init(int* __restrict__ bbox, unsigned long long* __restrict__ vkey, int nid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nid) return;
  for (int k = 0; k < 3; ++k) { bbox[i * 6 + k] = 0x7fffffff; bbox[i * 6 + 3 + k] = -1; }
  vkey[i * 2] = ~0ull;
  vkey[i * 2 + 1] = 0ull;
}

// the plain reads only filter: min/max are monotonic, so a stale value can cost an extra atomic,
// never skip a needed one
template <typename T>
__global__ __launch_bounds__(256) void stats_kernel(const int* __restrict__ seg, const T* __restrict__ raw,
                                                    int Z, int Y, int X, int nid, int* __restrict__ bbox,
                                                    unsigned long long* __restrict__ vkey) {
  const long long npix = (long long)Z * Y * X;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
       p += (long long)gridDim.x * blockDim.x) {
    const int id = seg[p];
    if (id <= 0 || id >= nid) continue;
    const int x = (int)(p % X), y = (int)((p / X) % Y), z = (int)(p / ((long long)X * Y));
    const unsigned long long k = order_key(raw[p]);
    int* b = bbox + (size_t)id * 6;
    if (z < b[0]) atomicMin(b + 0, z);
    if (y < b[1]) atomicMin(b + 1, y);
    if (x < b[2]) atomicMin(b + 2, x);
    if (z > b[3]) atomicMax(b + 3, z);
    if (y > b[4]) atomicMax(b + 4, y);
    if (x > b[5]) atomicMax(b + 5, x);
    unsigned long long* v = vkey + (size_t)id * 2;
    if (k < v[0]) atomicMin(v + 0, k);
    if (k > v[1]) atomicMax(v + 1, k);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void inst_hist_kernel(const int* __restrict__ seg, const T* __restrict__ raw,
                                                        long long npix, const int* __restrict__ slot, int nid,
                                                        const T* __restrict__ edges, int nbins,
                                                        unsigned int* __restrict__ counts) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
       p += (long long)gridDim.x * blockDim.x) {
    const int id = seg[p];
    if (id <= 0 || id >= nid) continue;
    const int s = slot[id];
    if (s < 0) continue;
    const T* e = edges + (size_t)s * (nbins + 1);
    const T v = raw[p];
    const T first = e[0], last = e[nbins];
    const T denom = last - first;
    const T f = ((v - first) / denom) * (T)nbins;      // numpy: (a - first_edge) / norm_denom * n_bins
    int idx = (int)f;
    if (idx == nbins) idx -= 1;
    if (v < e[idx]) idx -= 1;
    if (v >= e[idx + 1] && idx != nbins - 1) idx += 1;
    atomicAdd(&counts[(size_t)s * nbins + idx], 1u);
  }
}

__global__ __launch_bounds__(256) void inst_hist_int_kernel(const int* __restrict__ seg, const int* __restrict__ raw,
                                                            long long npix, const int* __restrict__ slot, int nid,
                                                            const int* __restrict__ vmin, int nbins,
                                                            unsigned int* __restrict__ counts) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
       p += (long long)gridDim.x * blockDim.x) {
    const int id = seg[p];
    if (id <= 0 || id >= nid) continue;
    const int s = slot[id];
    if (s < 0) continue;
    const long long idx = (long long)raw[p] - vmin[s];
    if (idx >= 0 && idx < nbins) atomicAdd(&counts[(size_t)s * nbins + idx], 1u);
  }
}

// state per bounding-box voxel: 0 = instance (foreground), 1 = background not reached yet,
// 2 = background connected to the outside of the box
template <typename T>
__global__ __launch_bounds__(256) void refine_kernel(const int* __restrict__ seg, const T* __restrict__ raw,
                                                     int zborder, int Y, int X, const int* __restrict__ ids,
                                                     const int* __restrict__ bbox, const double* __restrict__ thr,
                                                     const long long* __restrict__ scratch_off,
                                                     unsigned char* __restrict__ scratch, int* __restrict__ out) {
  const int inst = blockIdx.x;
  const int id = ids[inst];
  const int* b = bbox + (size_t)id * 6;
  const int z0 = b[0], y0 = b[1], x0 = b[2];
  const int bz = b[3] - z0 + 1, by = b[4] - y0 + 1, bx = b[5] - x0 + 1;
  const int vol = bz * by * bx;
  unsigned char* st = scratch + scratch_off[inst];
  const double t = thr[inst];
  __shared__ int changed;

  for (int i = threadIdx.x; i < vol; i += blockDim.x) {
    const int x = i % bx, y = (i / bx) % by, z = i / (bx * by);
    const long long p = ((long long)(z0 + z) * Y + (y0 + y)) * X + (x0 + x);
    const bool fg = seg[p] == id && (double)raw[p] > t;
    const bool edge = x == 0 || x == bx - 1 || y == 0 || y == by - 1 || (zborder && (z == 0 || z == bz - 1));
    st[i] = fg ? 0 : (edge ? 2 : 1);
  }
  __syncthreads();

  const int sy = bx, sz = bx * by;
  for (;;) {
    if (threadIdx.x == 0) changed = 0;
    __syncthreads();
    int local = 0;
    // sweep every x-line forward and backward, also looking at the y/z neighbours
    for (int line = threadIdx.x; line < by * bz; line += blockDim.x) {
      const int y = line % by, z = line / by;
      unsigned char* row = st + z * sz + y * sy;
      for (int dir = 0; dir < 2; ++dir) {
        bool prev = false;
        for (int k = 0; k < bx; ++k) {
          const int x = dir ? bx - 1 - k : k;
          unsigned char s = row[x];
          if (s == 1) {
            bool reach = prev;
            if (!reach && y > 0) reach = row[x - sy] == 2;
            if (!reach && y < by - 1) reach = row[x + sy] == 2;
            if (!reach && z > 0) reach = row[x - sz] == 2;
            if (!reach && z < bz - 1) reach = row[x + sz] == 2;
            if (reach) { row[x] = s = 2; local = 1; }
          }
          prev = s == 2;
        }
      }
    }
    __syncthreads();
    // ... and every y-line, so that the front crosses the box in a few rounds in both axes
    for (int line = threadIdx.x; line < bx * bz; line += blockDim.x) {
      const int x = line % bx, z = line / bx;
      unsigned char* col = st + z * sz + x;
      for (int dir = 0; dir < 2; ++dir) {
        bool prev = false;
        for (int k = 0; k < by; ++k) {
          const int y = dir ? by - 1 - k : k;
          unsigned char s = col[y * sy];
          if (s == 1 && prev) { col[y * sy] = s = 2; local = 1; }
          prev = s == 2;
        }
      }
    }
    if (local) changed = 1;
    __syncthreads();
    if (!changed) break;
    __syncthreads();
  }

  for (int i = threadIdx.x; i < vol; i += blockDim.x) {
    if (st[i] == 2) continue;
    const int x = i % bx, y = (i / bx) % by, z = i / (bx * by);
    atomicMax(&out[((long long)(z0 + z) * Y + (y0 + y)) * X + (x0 + x)], id);
  }
}

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int clx_inst_stats(const int32_t* seg, const void* raw, int raw_type, int Z, int Y, int X, int nid,
                              int32_t* bbox, unsigned long long* vkey, clx_stream stream) {
  CLX_REQUIRE(seg && raw && bbox && vkey, "clx_inst_stats: null pointer");
  CLX_REQUIRE(Z > 0 && Y > 0 && X > 0 && nid > 0, "clx_inst_stats: bad shape");
  CLX_REQUIRE(raw_type >= RAW_F32 && raw_type <= RAW_I32, "clx_inst_stats: raw_type must be 0 (f32), 1 (f64) or 2 (i32)");
  hipStream_t st = (hipStream_t)stream;
  stats_init<<<(nid + 255) / 256, 256, 0, st>>>(bbox, vkey, nid);
  const long long npix = (long long)Z * Y * X;
  const int grid = grid_for(npix, 256);
  if (raw_type == RAW_F32) stats_kernel<float><<<grid, 256, 0, st>>>(seg, (const float*)raw, Z, Y, X, nid, bbox, vkey);
  else if (raw_type == RAW_F64) stats_kernel<double><<<grid, 256, 0, st>>>(seg, (const double*)raw, Z, Y, X, nid, bbox, vkey);
  else stats_kernel<int><<<grid, 256, 0, st>>>(seg, (const int*)raw, Z, Y, X, nid, bbox, vkey);
  CLX_CHECK_LAUNCH("clx_inst_stats");
  return CLX_OK;
}

extern "C" int clx_inst_histogram(const int32_t* seg, const void* raw, int raw_type, long long npix,
                                  const int32_t* slot, int nid, const void* edges_or_min, int nbins,
                                  uint32_t* counts, clx_stream stream) {
  CLX_REQUIRE(seg && raw && slot && edges_or_min && counts, "clx_inst_histogram: null pointer");
  CLX_REQUIRE(npix > 0 && nid > 0 && nbins > 0, "clx_inst_histogram: bad sizes");
  CLX_REQUIRE(raw_type >= RAW_F32 && raw_type <= RAW_I32, "clx_inst_histogram: raw_type must be 0 (f32), 1 (f64) or 2 (i32)");
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for(npix, 256);
  if (raw_type == RAW_F32)
    inst_hist_kernel<float><<<grid, 256, 0, st>>>(seg, (const float*)raw, npix, slot, nid, (const float*)edges_or_min, nbins, counts);
  else if (raw_type == RAW_F64)
    inst_hist_kernel<double><<<grid, 256, 0, st>>>(seg, (const double*)raw, npix, slot, nid, (const double*)edges_or_min, nbins, counts);
  else
    inst_hist_int_kernel<<<grid, 256, 0, st>>>(seg, (const int*)raw, npix, slot, nid, (const int*)edges_or_min, nbins, counts);
  CLX_CHECK_LAUNCH("clx_inst_histogram");
  return CLX_OK;
}

extern "C" int clx_inst_refine(const int32_t* seg, const void* raw, int raw_type, int ndim, int Z, int Y, int X,
                               const int32_t* ids, const int32_t* bbox, const double* thr,
                               const long long* scratch_off, unsigned char* scratch, int n, int32_t* out,
                               clx_stream stream) {
  CLX_REQUIRE(seg && raw && out, "clx_inst_refine: null pointer");
  CLX_REQUIRE(Z > 0 && Y > 0 && X > 0 && n >= 0, "clx_inst_refine: bad shape");
  CLX_REQUIRE(ndim == 3 || (ndim == 2 && Z == 1), "clx_inst_refine: ndim must be 2 (Z == 1) or 3");
  CLX_REQUIRE(raw_type >= RAW_F32 && raw_type <= RAW_I32, "clx_inst_refine: raw_type must be 0 (f32), 1 (f64) or 2 (i32)");
  if (n == 0) return CLX_OK;
  CLX_REQUIRE(ids && bbox && thr && scratch_off && scratch, "clx_inst_refine: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (raw_type == RAW_F32)
    refine_kernel<float><<<n, 256, 0, st>>>(seg, (const float*)raw, ndim == 3, Y, X, ids, bbox, thr, scratch_off, scratch, out);
  else if (raw_type == RAW_F64)
    refine_kernel<double><<<n, 256, 0, st>>>(seg, (const double*)raw, ndim == 3, Y, X, ids, bbox, thr, scratch_off, scratch, out);
  else
    refine_kernel<int><<<n, 256, 0, st>>>(seg, (const int*)raw, ndim == 3, Y, X, ids, bbox, thr, scratch_off, scratch, out);
  CLX_CHECK_LAUNCH("clx_inst_refine");
  return CLX_OK;
}
