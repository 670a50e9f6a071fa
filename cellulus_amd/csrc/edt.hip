// Exact squared Euclidean distance transform (integer arithmetic) and the
// grow/shrink post-processing built on it.
//
// scipy.ndimage.distance_transform_edt(a) returns, for every non-zero element
// of `a`, the distance to the nearest zero element.  cellulus/segment.py:41-51
// only compares those distances with small integers (grow_distance,
// shrink_distance), and d < g  <=>  d^2 < g^2 for the exact integer d^2, so the
// whole post-processing is done on squared distances — bit-exact.
//
// Separable min-plus: pass X finds the squared distance to the nearest zero in
// the same row by an outward search; passes Y and Z take
// min_{q}( (p-q)^2 + g[q] ) searching outward and stopping as soon as
// (p-q)^2 >= best (no farther candidate can win), which is exact.  The callers only
// compare distances with a small bound, so every search is additionally capped at
// `cap` steps: results < cap^2 are exact, larger ones are reported as >= cap^2.
#include "clx_common.h"

namespace {

constexpr int EDT_INF = 1 << 29;

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

__global__ void edt_pass_x(const unsigned char* __restrict__ in, int* __restrict__ g, int X,
                           long long npix, int cap, int* __restrict__ any_zero) {
  bool seen = false;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % X);
    const long long row = i - x;
    int best = EDT_INF;
    seen = seen || (in[i] == 0);
    for (int d = 0; d < cap; ++d) {
      const int xl = x - d, xr = x + d;
      if (xl < 0 && xr >= X) break;
      if ((xl >= 0 && in[row + xl] == 0) || (xr < X && in[row + xr] == 0)) {
        best = d * d;
        break;
      }
    }
    g[i] = best;
  }
  if (__any(seen) && (threadIdx.x & 63) == 0 && __hip_atomic_load(any_zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
    atomicOr(any_zero, 1);
}

// min over the axis with stride `stride` and extent `n` (axis index = (i / stride) % n).
// final != 0 (last pass): an image without any zero element has no finite
// distance; scipy then reports the distance to a phantom zero at index -1 of the
// FIRST axis (0 on the others) — reproduced here so the result stays bit-exact.
__global__ void edt_pass_axis(const int* __restrict__ g, int* __restrict__ out, int n,
                              long long stride, long long npix, int final, int Z, int Y, int X,
                              int cap, const int* __restrict__ any_zero) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)((i / stride) % n);
    int best = g[i];
    for (int d = 1; d < cap; ++d) {
      const int d2 = d * d;
      if (d2 >= best) break;
      const int lo = p - d, hi = p + d;
      if (lo < 0 && hi >= n) break;
      if (lo >= 0) best = min(best, d2 + g[i - (long long)d * stride]);
      if (hi < n) best = min(best, d2 + g[i + (long long)d * stride]);
    }
    best = min(best, EDT_INF);
    if (final && best >= EDT_INF && *any_zero == 0) {
      const int x = (int)(i % X);
      const long long t = i / X;
      const int y = (int)(t % Y);
      const int z = (int)(t / Y);
      best = (Z > 1) ? (z + 1) * (z + 1) + y * y + x * x : (y + 1) * (y + 1) + x * x;
    }
    out[i] = best;
  }
}

__global__ void mask_eq_zero(const int* __restrict__ seg, unsigned char* __restrict__ m, long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x)
    m[i] = (seg[i] == 0) ? 1 : 0;
}

__global__ void mask_lt(const int* __restrict__ d, int bound, unsigned char* __restrict__ m, long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x)
    m[i] = (d[i] < bound) ? 1 : 0;
}

__global__ void zero_where_lt(int* __restrict__ seg, const int* __restrict__ d, int bound, long long npix) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x)
    if (d[i] < bound) seg[i] = 0;
}

// tmp: npix ints + 1 flag int.  cap <= 0: unlimited search.
int edt_run(const unsigned char* in, int* out, int Z, int Y, int X, int* tmp, int cap, hipStream_t st) {
  const long long npix = (long long)Z * Y * X;
  const int grid = grid_for(npix, 256);
  int* any_zero = tmp + npix;
  const int big = 1 << 20;
  const int cx = cap > 0 ? (cap < X ? cap : X) : X;
  const int cy = cap > 0 ? cap : big, cz = cap > 0 ? cap : big;
  if (hipMemsetAsync(any_zero, 0, sizeof(int), st) != hipSuccess) return CLX_ERR_LAUNCH;
  // X pass -> tmp ; Y pass tmp -> out ; Z pass out -> tmp -> copy (3-D only)
  edt_pass_x<<<grid, 256, 0, st>>>(in, tmp, X, npix, cx, any_zero);
  edt_pass_axis<<<grid, 256, 0, st>>>(tmp, out, Y, (long long)X, npix, Z > 1 ? 0 : 1, Z, Y, X,
                                      cy < Y ? cy : Y, any_zero);
  if (Z > 1) {
    edt_pass_axis<<<grid, 256, 0, st>>>(out, tmp, Z, (long long)X * Y, npix, 1, Z, Y, X,
                                        cz < Z ? cz : Z, any_zero);
    if (hipMemcpyAsync(out, tmp, (size_t)npix * sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess)
      return CLX_ERR_LAUNCH;
  }
  return CLX_OK;
}


// ---------------------------------------------------------------------------------------------
// grow / shrink in ONE tile kernel.  Both distance tests have a small integer bound (d1 < grow,
// d2 < shrink), so a pixel's fate depends only on the foreground within grow - 1 + shrink - 1
// pixels of it: a block loads the byte mask of its tile plus that halo into LDS, runs both capped
// separable transforms there (x pass, y pass [, z pass] on 16-bit partial squared distances) and
// writes the zeros back.  HBM traffic: one 1-byte mask pass (4 B read + 1 B write per pixel),
// then ~1.4 B read per pixel + 4 B per pixel that is actually cleared — instead of the seven
// full-image int32 passes of the generic transform above.
//
// scipy's phantom zero: an EDT input without any zero element measures distances to index -1 of
// the first axis.  (1) No foreground at all: seg is all zero and stays so, whatever the masks
// say.  (2) Every pixel is within `grow` of foreground (nothing "non-expanded" anywhere): the
// tile kernel clears nothing and raises no flag; grow_shrink_phantom then clears the pixels with
// (first-axis + 1)^2 + other^2 < shrink^2, exactly what the generic path computes.
// ---------------------------------------------------------------------------------------------
constexpr unsigned short GS_INF = 0xffff;

template <int ND>
struct GsTile;
template <> struct GsTile<2> { static constexpr int TZ = 1, TY = 32, TX = 64; };
template <> struct GsTile<3> { static constexpr int TZ = 8, TY = 8, TX = 32; };

__global__ void fg_mask_kernel(const int* __restrict__ seg, unsigned char* __restrict__ m, long long npix) {
  // four pixels per thread: one 16-byte load, one 4-byte store
  const long long n4 = npix >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    const int4 v = reinterpret_cast<const int4*>(seg)[i];
    const unsigned int b = (v.x != 0 ? 1u : 0u) | (v.y != 0 ? 0x100u : 0u) | (v.z != 0 ? 0x10000u : 0u) |
                           (v.w != 0 ? 0x1000000u : 0u);
    reinterpret_cast<unsigned int*>(m)[i] = b;
  }
  if (blockIdx.x == 0 && threadIdx.x < (npix & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    m[i] = seg[i] != 0;
  }
}

// capped 1-D pass along the fastest axis of a byte image in LDS: out = d^2 of the nearest set
// byte within |d| < cap, GS_INF if none.  in: rows x win, out: rows x wout, out column c reads
// in column c + cap - 1 (so all taps are inside the input row).
__device__ __forceinline__ void gs_row_pass(const unsigned char* in, int win, unsigned short* out, int wout,
                                            int rows, int cap, int tid) {
  const int c1 = cap > 0 ? cap - 1 : 0;
  for (int idx = tid; idx < rows * wout; idx += 256) {
    const int r = idx / wout, c = idx - r * wout;
    const unsigned char* row = in + r * win + c + c1;
    unsigned short best = GS_INF;
    for (int d = 0; d < cap; ++d)
      if (row[-d] | row[d]) { best = (unsigned short)(d * d); break; }
    out[idx] = best;
  }
}

// capped min-plus pass along an outer axis: out[o][i] = min_{|d| < cap} d^2 + in[o + c1 + d][i]
// for `nout` output planes of `inner` elements each (input has nout + 2 c1 planes)
__device__ __forceinline__ void gs_axis_pass(const unsigned short* in, unsigned short* out, int nout, int inner,
                                             int cap, int tid) {
  const int c1 = cap > 0 ? cap - 1 : 0;
  for (int idx = tid; idx < nout * inner; idx += 256) {
    const int o = idx / inner, i = idx - o * inner;
    unsigned int best = GS_INF;
    for (int d = -c1; d <= c1; ++d) {
      const unsigned int v = in[(o + c1 + d) * inner + i];
      if (v != GS_INF) best = min(best, v + (unsigned int)(d * d));
    }
    out[idx] = (unsigned short)best;
  }
}

// GC / SC >= 0: grow / shrink known at compile time (the reference's defaults 3 and 6 are
// instantiated) — every region extent is then a constant, the index divisions become
// multiply-shifts and the tap loops unroll; -1: taken from the arguments.
template <int ND, int GC, int SC>
__global__ __launch_bounds__(256) void grow_shrink_tile_kernel(int* __restrict__ seg,
                                                               const unsigned char* __restrict__ fgm,
                                                               int Z, int Y, int X, int grow_rt, int shrink_rt,
                                                               int* __restrict__ flag_nonexp) {
  using T = GsTile<ND>;
  extern __shared__ unsigned char gs_smem[];
  const int tid = threadIdx.x;
  const int grow = GC >= 0 ? GC : grow_rt, shrink = SC >= 0 ? SC : shrink_rt;
  const int g1 = grow > 0 ? grow - 1 : 0, s1 = shrink > 0 ? shrink - 1 : 0, H = g1 + s1;
  const int hz = (ND == 3) ? 1 : 0;        // no halo / pass along z in 2-D
  // region 0 (foreground mask): tile + H; region 1 (expanded mask): tile + s1
  const int Z0 = T::TZ + 2 * H * hz, Y0 = T::TY + 2 * H, X0 = T::TX + 2 * H;
  const int Z1 = T::TZ + 2 * s1 * hz, Y1 = T::TY + 2 * s1, X1 = T::TX + 2 * s1;
  const int tz0 = blockIdx.z * T::TZ, ty0 = blockIdx.y * T::TY, tx0 = blockIdx.x * T::TX;
  // LDS carve-up (sizes in the launcher): byte masks, then two 16-bit ping-pong planes
  unsigned char* fg = gs_smem;                                   // Z0*Y0*X0
  unsigned char* ne = fg + ((Z0 * Y0 * X0 + 15) & ~15);          // Z1*Y1*X1
  unsigned short* pa = reinterpret_cast<unsigned short*>(ne + ((Z1 * Y1 * X1 + 15) & ~15));
  unsigned short* pb = pa + ((Z0 * Y0 * X1 + 7) & ~7);           // pa: Z0*Y0*X1, pb: Z0*Y1*X1

  int any_fg = 0;
  for (int idx = tid; idx < Z0 * Y0 * X0; idx += 256) {
    const int rx = idx % X0, t = idx / X0, ry = t % Y0, rz = t / Y0;
    const int z = tz0 - H * hz + rz, y = ty0 - H + ry, x = tx0 - H + rx;
    const bool in = (unsigned)z < (unsigned)Z && (unsigned)y < (unsigned)Y && (unsigned)x < (unsigned)X;
    const unsigned char v = in ? fgm[((long long)z * Y + y) * X + x] : 0;
    fg[idx] = v;
    any_fg |= v;
  }
  if (!__syncthreads_or(any_fg)) {
    // no foreground within reach: nothing is expanded, nothing can be cleared; every pixel of
    // the (never empty) tile is a zero of the second transform
    if (tid == 0 && __hip_atomic_load(flag_nonexp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(flag_nonexp, 1);
    return;
  }
  // ---- d1 < grow^2: x pass (rows of region 0, columns of region 1), y pass, z pass
  gs_row_pass(fg, X0, pa, X1, Z0 * Y0, grow, tid);      // out col c <-> region-0 col c + g1
  __syncthreads();
  for (int rz = 0; rz < Z0; ++rz)
    gs_axis_pass(pa + rz * Y0 * X1, pb + rz * Y1 * X1, Y1, X1, grow, tid);
  __syncthreads();
  const unsigned short* d1 = pb;
  if (ND == 3) {
    gs_axis_pass(pb, pa, Z1, Y1 * X1, grow, tid);
    __syncthreads();
    d1 = pa;
  }
  const unsigned int g2 = grow > 0 ? (unsigned)(grow * grow) : 0u;
  int any = 0;
  for (int idx = tid; idx < Z1 * Y1 * X1; idx += 256) {
    const int cx = idx % X1, t = idx / X1, cy = t % Y1, cz = t / Y1;
    const int z = tz0 - s1 * hz + cz, y = ty0 - s1 + cy, x = tx0 - s1 + cx;
    const bool in = (unsigned)z < (unsigned)Z && (unsigned)y < (unsigned)Y && (unsigned)x < (unsigned)X;
    const unsigned int d = d1[idx];
    const bool expanded = d != GS_INF && d < g2;
    const unsigned char v = (in && !expanded) ? 1 : 0;     // outside the image: not a zero of the 2nd EDT
    ne[idx] = v;
    const bool interior = cx >= s1 && cx < s1 + T::TX && cy >= s1 && cy < s1 + T::TY &&
                          cz >= s1 * hz && cz < s1 * hz + T::TZ;
    any |= (v && interior) ? 1 : 0;
  }
  if (__any(any) && (tid & 63) == 0 && __hip_atomic_load(flag_nonexp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
    atomicOr(flag_nonexp, 1);
  __syncthreads();
  // ---- d2 < shrink^2 on the tile
  gs_row_pass(ne, X1, pa, T::TX, Z1 * Y1, shrink, tid);
  __syncthreads();
  for (int cz = 0; cz < Z1; ++cz)
    gs_axis_pass(pa + cz * Y1 * T::TX, pb + cz * T::TY * T::TX, T::TY, T::TX, shrink, tid);
  __syncthreads();
  const unsigned short* d2 = pb;
  if (ND == 3) {
    gs_axis_pass(pb, pa, T::TZ, T::TY * T::TX, shrink, tid);
    __syncthreads();
    d2 = pa;
  }
  const unsigned int s2 = shrink > 0 ? (unsigned)(shrink * shrink) : 0u;
  for (int idx = tid; idx < T::TZ * T::TY * T::TX; idx += 256) {
    const int ix = idx % T::TX, t = idx / T::TX, iy = t % T::TY, iz = t / T::TY;
    const int z = tz0 + iz, y = ty0 + iy, x = tx0 + ix;
    if (z >= Z || y >= Y || x >= X) continue;
    const unsigned int d = d2[idx];
    if (d != GS_INF && d < s2 && fg[((iz + H * hz) * Y0 + iy + H) * X0 + ix + H])
      seg[((long long)z * Y + y) * X + x] = 0;
  }
}

// case (2) of the header comment: runs only if no tile saw a non-expanded pixel
__global__ void grow_shrink_phantom(int* __restrict__ seg, int Z, int Y, int X, int shrink,
                                    const int* __restrict__ flag_nonexp) {
  if (*flag_nonexp != 0 || shrink <= 0) return;
  const long long npix = (long long)Z * Y * X;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % X);
    const long long t = i / X;
    const int y = (int)(t % Y), z = (int)(t / Y);
    const int d = (Z > 1) ? (z + 1) * (z + 1) + y * y + x * x : (y + 1) * (y + 1) + x * x;
    if (d < shrink * shrink) seg[i] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// 2-D grow / shrink on BIT rows.  "d1 < grow" is a dilation of the foreground by the disc
// {dx^2 + dy^2 < grow^2}, "d2 < shrink" a dilation of the non-expanded set by the disc of the shrink
// bound; on a row packed 64 pixels per word a horizontal dilation by r is 2r shift-ORs, and with one
// image row per LANE the vertical part is cross-lane shuffles — no LDS, no per-pixel work at all:
//   gs_bits_kernel : seg -> 1 bit per pixel (wave ballot), 4 B read + 1/8 B written per pixel
//   gs_rows_kernel : a wavefront owns 64 rows (H = grow - 1 + shrink - 1 halo rows on either side)
//                    of one 64-pixel word column, reads the word and its two neighbours per row
//                    (a 128-bit window: 32 halo pixels each side), computes the pixels to clear
//                    and stores zeros there — the only 4-byte traffic after the first pass.
// The phantom-zero cases are the ones of the tile kernel above.
// ---------------------------------------------------------------------------------------------
typedef unsigned __int128 u128;

__global__ __launch_bounds__(256) void gs_bits_kernel(const int* __restrict__ seg, unsigned long long* __restrict__ bits,
                                                      int Y, int X, int W64) {
  const int lane = threadIdx.x & 63;
  const long long nwaves = (long long)Y * W64;
  for (long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwaves;
       w += ((long long)gridDim.x * blockDim.x) >> 6) {
    const int y = (int)(w / W64), wx = (int)(w - (long long)y * W64);
    const int x = wx * 64 + lane;
    const bool f = x < X && seg[(long long)y * X + x] != 0;
    const unsigned long long b = __ballot(f);
    if (lane == 0) bits[w] = b;
  }
}

// The same with 16-byte loads (row length a multiple of 4, aligned image): a wavefront takes 256 pixels of a row = four
// words; a lane's four pixels are one nibble, sixteen lanes' nibbles one word (OR over xor-shuffles).  One 4-byte load
// per lane ran at 2.3 TB/s (round 4: 119 us at 8192^2 for 268 MB).
typedef int gs_i32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gs_bits4_kernel(const int* __restrict__ seg, unsigned long long* __restrict__ bits,
                                                       int Y, int X, int W64) {
  const int lane = threadIdx.x & 63;
  const int G4 = (W64 + 3) >> 2;
  const long long nwaves = (long long)Y * G4;
  for (long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwaves;
       w += ((long long)gridDim.x * blockDim.x) >> 6) {
    const int y = (int)(w / G4), wg = (int)(w - (long long)y * G4);
    const int x = wg * 256 + 4 * lane;
    gs_i32x4 v = {0, 0, 0, 0};
    if (x < X) v = *reinterpret_cast<const gs_i32x4*>(seg + (long long)y * X + x);       // X % 4 == 0: x + 3 < X
    const unsigned int nib = (v[0] != 0 ? 1u : 0u) | (v[1] != 0 ? 2u : 0u) | (v[2] != 0 ? 4u : 0u) | (v[3] != 0 ? 8u : 0u);
    unsigned long long word = (unsigned long long)nib << (4 * (lane & 15));
    word |= __shfl_xor(word, 1, 64);
    word |= __shfl_xor(word, 2, 64);
    word |= __shfl_xor(word, 4, 64);
    word |= __shfl_xor(word, 8, 64);
    const int wi = wg * 4 + (lane >> 4);
    if ((lane & 15) == 0 && wi < W64) bits[(long long)y * W64 + wi] = word;
  }
}

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
  const int lo = __shfl((int)(unsigned int)v, src, 64), hi = __shfl((int)(unsigned int)(v >> 32), src, 64);
  return ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
}

// largest dx >= 0 with dx^2 + dy^2 < bound^2, -1 if none (dy >= bound)
__device__ __forceinline__ int disc_halfwidth(int dy, int bound) {
  if (bound <= 0) return -1;          // "distance < bound" never holds
  int r = -1;
  while ((r + 1) * (r + 1) + dy * dy < bound * bound) ++r;
  return r;
}

__global__ __launch_bounds__(256) void gs_rows_kernel(int* __restrict__ seg, const unsigned long long* __restrict__ bits,
                                                      int Y, int X, int W64, int grow, int shrink, int rows_per_wave,
                                                      int ntiles_y, int* __restrict__ flag_nonexp) {
  const int lane = threadIdx.x & 63;
  const int g1 = grow > 0 ? grow - 1 : 0, s1 = shrink > 0 ? shrink - 1 : 0, H = g1 + s1;
  const long long nwaves = (long long)ntiles_y * W64;
  for (long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwaves;
       w += ((long long)gridDim.x * blockDim.x) >> 6) {
    const int ty = (int)(w / W64), wx = (int)(w - (long long)ty * W64);
    const int y = ty * rows_per_wave - H + lane;               // this lane's image row
    const bool row_in = (unsigned)y < (unsigned)Y;
    unsigned long long wl = 0, wm = 0, wr = 0;
    if (row_in) {
      const unsigned long long* row = bits + (long long)y * W64;
      wm = row[wx];
      if (wx > 0) wl = row[wx - 1];
      if (wx + 1 < W64) wr = row[wx + 1];
    }
    // window of pixels [64 wx - 32, 64 wx + 96): bit 32 + k is pixel 64 wx + k
    const u128 fg = ((u128)wr << 96) | ((u128)wm << 32) | (u128)(wl >> 32);
    // pixels of the window that exist
    u128 inimg = 0;
    if (row_in) {
      const int x_lo = wx * 64 - 32;
      const int first = x_lo < 0 ? -x_lo : 0;
      const int last = min(128, X - x_lo);                       // exclusive
      inimg = (last >= 128 ? ~(u128)0 : (((u128)1 << last) - 1)) & ~(((u128)1 << first) - 1);
    }
    // ---- expanded = foreground dilated by the disc of `grow`
    u128 expanded = 0;
    {
      u128 D = 0;
      int cur = -1;
      for (int ady = g1; ady >= 0; --ady) {
        const int R = disc_halfwidth(ady, grow);
        if (R < 0) continue;
        while (cur < R) { ++cur; D |= (fg << cur) | (fg >> cur); }
        const unsigned long long dlo = (unsigned long long)D, dhi = (unsigned long long)(D >> 64);
        const u128 up = ((u128)shfl_u64(dhi, lane - ady) << 64) | shfl_u64(dlo, lane - ady);
        expanded |= up;
        if (ady) expanded |= ((u128)shfl_u64(dhi, lane + ady) << 64) | shfl_u64(dlo, lane + ady);
      }
    }
    const u128 nonexp = ~expanded & inimg;
    const bool interior = lane >= H && lane < H + rows_per_wave && row_in;
    const unsigned long long ne_mid = (unsigned long long)(nonexp >> 32);
    // (one flag for the whole image: raise it only while it is still down — thousands of wavefronts would
    //  otherwise queue their atomics on one address: 71 us of a 4096^2 image)
    if (__any(interior && ne_mid != 0) && lane == 0 &&
        __hip_atomic_load(flag_nonexp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
      atomicOr(flag_nonexp, 1);
    // ---- cleared = non-expanded set dilated by the disc of `shrink` (middle word only)
    unsigned long long clear = 0;
    {
      u128 D = 0;
      int cur = -1;
      for (int ady = s1; ady >= 0; --ady) {
        const int R = disc_halfwidth(ady, shrink);
        if (R < 0) continue;
        while (cur < R) { ++cur; D |= (nonexp << cur) | (nonexp >> cur); }
        const unsigned long long dm = (unsigned long long)(D >> 32);
        clear |= shfl_u64(dm, lane - ady);
        if (ady) clear |= shfl_u64(dm, lane + ady);
      }
    }
    clear &= wm;                                  // only foreground pixels hold anything to clear
    if (!interior) clear = 0;
    // ---- apply: every lane walks the set bits of ITS row's word (a rim of ~10 pixels per row that crosses objects: as many
    // trips, all rows at once; the round-4 form took one trip — two shuffles and a store of a few lanes — per ROW with
    // anything to clear, ~50 per wavefront)
    if (clear) {
      int* rowp = seg + (long long)y * X + wx * 64;
      do {
        const int b = __builtin_ctzll(clear);
        clear &= clear - 1;
        rowp[b] = 0;
      } while (clear);
    }
  }
}

int grow_shrink_bitrows(int* seg, int Y, int X, int grow, int shrink, void* workspace, hipStream_t st) {
  const int g1 = grow > 0 ? grow - 1 : 0, s1 = shrink > 0 ? shrink - 1 : 0, H = g1 + s1;
  const int W64 = (X + 63) / 64;
  const int rows_per_wave = 64 - 2 * H;
  const int ntiles_y = (Y + rows_per_wave - 1) / rows_per_wave;
  int* flag = (int*)workspace;
  unsigned long long* bits = (unsigned long long*)((unsigned char*)workspace + 16);
  if (hipMemsetAsync(flag, 0, sizeof(int), st) != hipSuccess) return CLX_ERR_LAUNCH;
  const long long w1 = (long long)Y * W64, w2 = (long long)ntiles_y * W64;
  if (X % 4 == 0 && ((uintptr_t)seg & 15) == 0)
    CLX_LAUNCH_KIND(CLX_PROF_GROW_SHRINK, gs_bits4_kernel, dim3(grid_for((long long)Y * ((W64 + 3) / 4) * 64, 256)), dim3(256), 0, st, seg, bits, Y, X, W64);
  else
    CLX_LAUNCH_KIND(CLX_PROF_GROW_SHRINK, gs_bits_kernel, dim3(grid_for(w1 * 64, 256)), dim3(256), 0, st, seg, bits, Y, X, W64);
  CLX_LAUNCH_KIND(CLX_PROF_GROW_SHRINK, gs_rows_kernel, dim3(grid_for(w2 * 64, 256)), dim3(256), 0, st, seg, bits, Y, X, W64, grow, shrink, rows_per_wave, ntiles_y,
                                                         flag);
  const long long npix = (long long)Y * X;
  CLX_LAUNCH_KIND(CLX_PROF_GROW_SHRINK, grow_shrink_phantom, dim3(grid_for(npix, 256) < 1024 ? grid_for(npix, 256) : 1024), dim3(256), 0, st, seg, 1, Y, X, shrink, flag);
  return CLX_OK;
}

template <int ND>
size_t gs_smem_bytes(int grow, int shrink) {
  using T = GsTile<ND>;
  const int g1 = grow > 0 ? grow - 1 : 0, s1 = shrink > 0 ? shrink - 1 : 0, H = g1 + s1;
  const int hz = (ND == 3) ? 1 : 0;
  const int Z0 = T::TZ + 2 * H * hz, Y0 = T::TY + 2 * H, X0 = T::TX + 2 * H;
  const int Z1 = T::TZ + 2 * s1 * hz, Y1 = T::TY + 2 * s1, X1 = T::TX + 2 * s1;
  return (size_t)((Z0 * Y0 * X0 + 15) & ~15) + ((Z1 * Y1 * X1 + 15) & ~15) +
         2 * (size_t)(((Z0 * Y0 * X1 + 7) & ~7) + Z0 * Y1 * X1 + 8);
}

template <int ND, int GC, int SC>
int gs_launch(dim3 grid, size_t smem, int* seg, const unsigned char* mask, int Z, int Y, int X, int grow, int shrink,
              int* flag, hipStream_t st) {
  if (smem > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(&grow_shrink_tile_kernel<ND, GC, SC>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return CLX_ERR_LAUNCH;
  CLX_LAUNCH_KIND(CLX_PROF_GROW_SHRINK, (grow_shrink_tile_kernel<ND, GC, SC>), dim3(grid), dim3(256), smem, st, seg, mask, Z, Y, X, grow, shrink, flag);
  return CLX_OK;
}

template <int ND>
int grow_shrink_tiled(int* seg, int Z, int Y, int X, int grow, int shrink, void* workspace, hipStream_t st) {
  using T = GsTile<ND>;
  const long long npix = (long long)Z * Y * X;
  int* flag = (int*)workspace;
  unsigned char* mask = (unsigned char*)workspace + 16;
  if (hipMemsetAsync(flag, 0, sizeof(int), st) != hipSuccess) return CLX_ERR_LAUNCH;
  CLX_LAUNCH_KIND(CLX_PROF_GROW_SHRINK, fg_mask_kernel, dim3(grid_for((npix + 3) / 4, 256)), dim3(256), 0, st, seg, mask, npix);
  const dim3 grid((X + T::TX - 1) / T::TX, (Y + T::TY - 1) / T::TY, (Z + T::TZ - 1) / T::TZ);
  const size_t smem = gs_smem_bytes<ND>(grow, shrink);
  const int rc = (grow == 3 && shrink == 6)      // segment.py's defaults (inference_config.py:158-159)
                     ? gs_launch<ND, 3, 6>(grid, smem, seg, mask, Z, Y, X, grow, shrink, flag, st)
                     : gs_launch<ND, -1, -1>(grid, smem, seg, mask, Z, Y, X, grow, shrink, flag, st);
  if (rc) return rc;
  CLX_LAUNCH_KIND(CLX_PROF_GROW_SHRINK, grow_shrink_phantom, dim3(grid_for(npix, 256) < 1024 ? grid_for(npix, 256) : 1024), dim3(256), 0, st, seg, Z, Y, X,
                                                                                               shrink, flag);
  return CLX_OK;
}

}  // namespace

extern "C" size_t clx_edt_workspace(long long npix) { return (size_t)(npix + 4) * sizeof(int); }

extern "C" int clx_edt_sq(const unsigned char* in, int* out, int Z, int Y, int X, int cap,
                          void* workspace, clx_stream stream) {
  CLX_REQUIRE(in && out && workspace, "clx_edt_sq: null pointer");
  CLX_REQUIRE(Z > 0 && Y > 0 && X > 0, "clx_edt_sq: bad extents");
  CLX_REQUIRE((long long)Z * Y * X < (1ll << 31), "clx_edt_sq: too many pixels");
  CLX_REQUIRE(Z < 16384 && Y < 16384 && X < 16384, "clx_edt_sq: extent too large for int32 distances");
  const int rc = edt_run(in, out, Z, Y, X, (int*)workspace, cap, (hipStream_t)stream);
  if (rc) { clx_set_error("clx_edt_sq: copy failed"); return rc; }
  CLX_CHECK_LAUNCH("clx_edt_sq");
  return CLX_OK;
}

extern "C" int clx_grow_shrink(int* seg, int Z, int Y, int X, int grow, int shrink,
                               void* workspace, clx_stream stream) {
  CLX_REQUIRE(seg && workspace, "clx_grow_shrink: null pointer");
  CLX_REQUIRE(Z > 0 && Y > 0 && X > 0, "clx_grow_shrink: bad extents");
  CLX_REQUIRE(Z < 16384 && Y < 16384 && X < 16384, "clx_grow_shrink: extent too large");
  const long long npix = (long long)Z * Y * X;
  CLX_REQUIRE(npix < (1ll << 31), "clx_grow_shrink: too many pixels");
  hipStream_t st0 = (hipStream_t)stream;
  // small bounds (always the case for cellulus: 3 and 6): the whole post-processing in one tile kernel
  {
    const int g1 = grow > 0 ? grow - 1 : 0, s1 = shrink > 0 ? shrink - 1 : 0;
    const bool three_d = Z > 1;
    if (!three_d && g1 + s1 <= 16 && ((uintptr_t)workspace & 15) == 0) {
      const int rc = grow_shrink_bitrows(seg, Y, X, grow, shrink, workspace, st0);
      if (rc) { clx_set_error("clx_grow_shrink: memset failed"); return rc; }
      CLX_CHECK_LAUNCH("clx_grow_shrink(bit rows)");
      return CLX_OK;
    }
    const size_t smem = three_d ? gs_smem_bytes<3>(grow, shrink) : gs_smem_bytes<2>(grow, shrink);
    if (g1 + s1 <= 24 && smem <= 150 * 1024 && ((uintptr_t)seg & 15) == 0 && ((uintptr_t)workspace & 15) == 0) {
      const int rc = three_d ? grow_shrink_tiled<3>(seg, Z, Y, X, grow, shrink, workspace, st0)
                             : grow_shrink_tiled<2>(seg, Z, Y, X, grow, shrink, workspace, st0);
      if (rc) { clx_set_error("clx_grow_shrink: memset failed"); return rc; }
      CLX_CHECK_LAUNCH("clx_grow_shrink(tiled)");
      return CLX_OK;
    }
  }
  int* tmp = (int*)workspace;
  int* dist = tmp + npix + 4;
  unsigned char* mask = (unsigned char*)(dist + npix);
  const int grid = grid_for(npix, 256);
  hipStream_t st = (hipStream_t)stream;
  // d1 = edt(seg == 0); expanded = d1 < grow
  mask_eq_zero<<<grid, 256, 0, st>>>(seg, mask, npix);
  // only "d < grow" / "d < shrink" are needed: cap the searches at those radii
  int rc = edt_run(mask, dist, Z, Y, X, tmp, grow > 0 ? grow : 1, st);
  if (rc) { clx_set_error("clx_grow_shrink: copy failed"); return rc; }
  // sqrt(d1) < grow  <=>  d1 < grow^2   (grow <= 0: nothing is expanded)
  mask_lt<<<grid, 256, 0, st>>>(dist, grow > 0 ? grow * grow : 0, mask, npix);
  // d2 = edt(expanded); seg[d2 < shrink] = 0
  rc = edt_run(mask, dist, Z, Y, X, tmp, shrink > 0 ? shrink : 1, st);
  if (rc) { clx_set_error("clx_grow_shrink: copy failed"); return rc; }
  zero_where_lt<<<grid, 256, 0, st>>>(seg, dist, shrink > 0 ? shrink * shrink : 0, npix);
  CLX_CHECK_LAUNCH("clx_grow_shrink");
  return CLX_OK;
}
