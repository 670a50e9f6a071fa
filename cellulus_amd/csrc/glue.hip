// HBM-bound streaming kernels around the convolutions: max-pool forward,
// the fused (max-pool backward + skip-gradient add + ReLU gate), the fused
// (nearest-upsample backward + crop + ReLU gate), inference statistics.
// All tensors are dense channels-last f32; one thread handles 4 channels of
// one pixel (16-byte accesses, channel-contiguous => fully coalesced).
#include "clx_common.h"

namespace {

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

// Linear index -> (4-channel group, x, y, z, batch) with host-prepared multiply-shift divisors: the
// 64-bit `%` / `/` chain this replaces cost more VALU time than the kernels' memory traffic
// (launchers require the element count / 4 to stay below 2^31).
struct Dec {
  FastDiv c4, w, h, d;
};
inline Dec make_dec(int C4, int W, int H, int D) {
  return Dec{make_fastdiv((uint32_t)C4), make_fastdiv((uint32_t)W), make_fastdiv((uint32_t)H), make_fastdiv((uint32_t)D)};
}
__device__ __forceinline__ void decode(uint32_t i, const Dec& dc, uint32_t& pixel, int& c4, int& x, int& y, int& z, int& b) {
  pixel = fdiv(i, dc.c4);
  c4 = (int)(i - pixel * dc.c4.d);
  const uint32_t q1 = fdiv(pixel, dc.w);
  x = (int)(pixel - q1 * dc.w.d);
  const uint32_t q2 = fdiv(q1, dc.h);
  y = (int)(q1 - q2 * dc.h.d);
  const uint32_t q3 = fdiv(q2, dc.d);
  z = (int)(q2 - q3 * dc.d.d);
  b = (int)q3;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// replaces funlib Downsample = nn.MaxPool{2,3}d(f, stride=f) [unet.py:24-51]
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                   int D, int H, int W, int C4, int fz, int fy, int fx,
                                   int OD, int OH, int OW, Dec dc, uint32_t total) {
  const int C = C4 * 4;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    uint32_t opix;
    int c4, ox, oy, oz, bi;
    decode(i, dc, opix, c4, ox, oy, oz, bi);
    const int c = c4 * 4;
    const long long b = bi;
    f32x4 m;
    bool first = true;
    for (int dz = 0; dz < fz; ++dz)
      for (int dy = 0; dy < fy; ++dy)
        for (int dx = 0; dx < fx; ++dx) {
          const long long pix = ((b * D + oz * fz + dz) * H + oy * fy + dy) * W + ox * fx + dx;
          const f32x4 v = ld4(x + pix * C + c);
          if (first) { m = v; first = false; }
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = (v[e] > m[e]) ? v[e] : m[e];
          }
        }
    st4(y + (size_t)opix * C + c, m);
  }
}

// max-pool backward + skip gradient + ReLU gate, one thread per WINDOW (x 4 channels): the window's values
// are read once, the first maximum is found in registers (no pooled tensor, no re-reads of the earlier window positions),
// the pooled gradient is read once per window instead of once per pixel
__global__ void maxpool_bwd_window_kernel(const float* __restrict__ x, const float* __restrict__ dyp,
                                          const float* __restrict__ dskip, int ld_skip, int SD, int SH, int SW,
                                          int cz, int cy, int cx, float* __restrict__ dxo, int D, int H, int W,
                                          int C4, int fz, int fy, int fx, Dec dcw, uint32_t total) {
  const int C = C4 * 4;
  const int OD = D / fz, OH = H / fy, OW = W / fx;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    uint32_t win;
    int c4, wx, wy, wz, bi;
    decode(i, dcw, win, c4, wx, wy, wz, bi);         // decoded over the POOLED grid (OW, OH, OD)
    const int c = c4 * 4;
    const f32x4 gv = ld4(dyp + (size_t)win * C + c);
    // pass 1: maximum and the scan position of its first occurrence (torch's rule), per channel
    f32x4 best = {0.f, 0.f, 0.f, 0.f};
    int arg[4] = {0, 0, 0, 0};
    int k = 0;
    for (int dz = 0; dz < fz; ++dz)
      for (int dy = 0; dy < fy; ++dy)
        for (int dx = 0; dx < fx; ++dx, ++k) {
          const size_t pix = (((size_t)bi * D + wz * fz + dz) * H + wy * fy + dy) * W + wx * fx + dx;
          const f32x4 v = ld4(x + pix * C + c);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k == 0 || v[e] > best[e]) { best[e] = v[e]; arg[e] = k; }
        }
    // pass 2 (the window is in cache): route the gradient, add the skip gradient, gate by the ReLU
    k = 0;
    for (int dz = 0; dz < fz; ++dz)
      for (int dy = 0; dy < fy; ++dy)
        for (int dx = 0; dx < fx; ++dx, ++k) {
          const int pz = wz * fz + dz, py = wy * fy + dy, px = wx * fx + dx;
          const size_t pix = (((size_t)bi * D + pz) * H + py) * W + px;
          const f32x4 xv = ld4(x + pix * C + c);
          f32x4 g;
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = (arg[e] == k) ? gv[e] : 0.f;
          if (dskip) {
            const int sz = pz - cz, sy = py - cy, sx = px - cx;
            if ((unsigned)sz < (unsigned)SD && (unsigned)sy < (unsigned)SH && (unsigned)sx < (unsigned)SW)
              g += ld4(dskip + ((((size_t)bi * SD + sz) * SH + sy) * SW + sx) * ld_skip + c);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = (xv[e] > 0.f) ? g[e] : 0.f;
          st4(dxo + pix * C + c, g);
        }
  }
}

__global__ void upsample_bwd_kernel(const float* __restrict__ dcat, int ld_cat, int coff,
                                    int LD, int LH, int LW, int oz, int oy, int ox,
                                    const float* __restrict__ y, float* __restrict__ dy,
                                    int D, int H, int W, int C4, int fz, int fy, int fx,
                                    Dec dc, uint32_t total) {
  const int C = C4 * 4;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    uint32_t pixel;
    int c4, qx, qy, qz, bi;
    decode(i, dc, pixel, c4, qx, qy, qz, bi);
    const int c = c4 * 4;
    const long long b = bi;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    for (int dz = 0; dz < fz; ++dz) {
      const int lz = qz * fz + dz - oz;
      if ((unsigned)lz >= (unsigned)LD) continue;
      for (int dyy = 0; dyy < fy; ++dyy) {
        const int ly = qy * fy + dyy - oy;
        if ((unsigned)ly >= (unsigned)LH) continue;
        for (int dx = 0; dx < fx; ++dx) {
          const int lx = qx * fx + dx - ox;
          if ((unsigned)lx >= (unsigned)LW) continue;
          const long long pix = ((b * LD + lz) * LH + ly) * LW + lx;
          g += ld4(dcat + pix * ld_cat + coff + c);
        }
      }
    }
    const f32x4 yv = ld4(y + (size_t)pixel * C + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = (yv[e] > 0.f) ? g[e] : 0.f;
    st4(dy + (size_t)pixel * C + c, g);
  }
}

// lo[(b,z,y,x)][phase*N + n] <-> hi[(b, z*fz+a, y*fy+bb, x*fx+c)][n], phase = (a*fy+bb)*fx+c
template <bool TO_SPACE>
__global__ void subpixel_kernel(const float* __restrict__ src, float* __restrict__ dst, int ld_lo,
                                int ld_hi, int D, int H, int W, int N4, int fz, int fy, int fx,
                                long long total) {
  const int P = fz * fy * fx;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N4) * 4;
    long long q = i / N4;
    const int ph = (int)(q % P); q /= P;
    const int x = (int)(q % W); q /= W;
    const int y = (int)(q % H); q /= H;
    const int z = (int)(q % D);
    const long long b = q / D;
    const int c = ph % fx, bb = (ph / fx) % fy, a = ph / (fx * fy);
    const long long lo_pix = ((b * D + z) * H + y) * W + x;
    const long long hi_pix = ((b * D * fz + z * fz + a) * (H * fy) + y * fy + bb) * (long long)(W * fx) + x * fx + c;
    const long long lo_off = lo_pix * ld_lo + (long long)ph * N4 * 4 + n;
    const long long hi_off = hi_pix * ld_hi + n;
    if (TO_SPACE) st4(dst + hi_off, ld4(src + lo_off));
    else st4(dst + lo_off, ld4(src + hi_off));
  }
}

// torch.std_mean(stack(preds), dim=0, unbiased=False); std summed over channels
// [cellulus/models/unet.py:90-98]
__global__ void noise_stats_kernel(const float* __restrict__ preds, float* __restrict__ out,
                                   int T, int C, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float std_sum = 0.f;
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      for (int t = 0; t < T; ++t) s += preds[((long long)t * C + c) * n + i];
      const float mean = s / (float)T;
      float v = 0.f;
      for (int t = 0; t < T; ++t) {
        const float d = preds[((long long)t * C + c) * n + i] - mean;
        v += d * d;
      }
      out[(long long)c * n + i] = mean;
      std_sum += sqrtf(v / (float)T);
    }
    out[(long long)C * n + i] = std_sum;
  }
}

constexpr int NOISE_MM_BLOCKS = 1024;     // == (CLX_NOISE_MINMAX_FLOATS - 2) / 2

// std_minmax: [0] min, [1] max of the std plane so far, [2 + 2 b], [3 + 2 b] the partials of block b of the last launch.
// Bit patterns of non-negative floats order like the numbers, so the reductions are unsigned-integer min / max; one
// small block folds the partials into [0..1] (no two blocks ever meet on an address: same-address atomics are served
// one after the other, ~12 ns each — csrc/otsu.hip).
__global__ __launch_bounds__(256) void noise_minmax_final(unsigned int* mm, int nblocks, int init) {
  __shared__ unsigned int slo[4], shi[4];
  unsigned int lo = 0xffffffffu, hi = 0u;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    const unsigned int l = mm[2 + 2 * b], h = mm[3 + 2 * b];
    lo = l < lo ? l : lo;
    hi = h > hi ? h : hi;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 0; w < 4; ++w) { lo = slo[w] < lo ? slo[w] : lo; hi = shi[w] > hi ? shi[w] : hi; }
    if (!init) { lo = mm[0] < lo ? mm[0] : lo; hi = mm[1] > hi ? mm[1] : hi; }
    mm[0] = lo;
    mm[1] = hi;
  }
}

// same arithmetic, the T samples of a pixel are read ONCE and kept in registers.  minmax (optional): the block's
// minimum / maximum of the std plane as float bit patterns (a NaN, whose pattern lies above +inf, ends up as the maximum)
template <int TCAP>
__global__ void noise_stats_reg_kernel(const float* __restrict__ preds, float* __restrict__ out,
                                       int T, int C, long long n, unsigned int* __restrict__ minmax) {
  unsigned int lo = 0xffffffffu, hi = 0u;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float std_sum = 0.f;
    for (int c = 0; c < C; ++c) {
      float v[TCAP];
#pragma unroll
      for (int t = 0; t < TCAP; ++t) v[t] = (t < T) ? preds[((long long)t * C + c) * n + i] : 0.f;
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < TCAP; ++t) s += (t < T) ? v[t] : 0.f;
      const float mean = s / (float)T;
      float var = 0.f;
#pragma unroll
      for (int t = 0; t < TCAP; ++t) {
        const float d = v[t] - mean;
        var += (t < T) ? d * d : 0.f;
      }
      out[(long long)c * n + i] = mean;
      std_sum += sqrtf(var / (float)T);
    }
    out[(long long)C * n + i] = std_sum;
    const unsigned int bits = __float_as_uint(std_sum + 0.f);      // (-0 + 0 = +0)
    lo = bits < lo ? bits : lo;
    hi = bits > hi ? bits : hi;
  }
  if (minmax != nullptr) {
    __shared__ unsigned int slo[4], shi[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 0; w < 4; ++w) { lo = slo[w] < lo ? slo[w] : lo; hi = shi[w] > hi ? shi[w] : hi; }
      minmax[2 + 2 * blockIdx.x] = lo;          // (a block without pixels: lo = all ones, hi = 0 — neutral)
      minmax[3 + 2 * blockIdx.x] = hi;
    }
  }
}

}  // namespace

extern "C" int clx_maxpool_fwd(const float* x, float* y, int B, int D, int H, int W, int C,
                               int fz, int fy, int fx, clx_stream stream) {
  CLX_REQUIRE(x && y, "clx_maxpool_fwd: null pointer");
  CLX_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "clx_maxpool_fwd: bad extents");
  CLX_REQUIRE(fz >= 1 && fy >= 1 && fx >= 1, "clx_maxpool_fwd: bad factors");
  CLX_REQUIRE(D % fz == 0 && H % fy == 0 && W % fx == 0,
              "clx_maxpool_fwd: extent (%d,%d,%d) not divisible by factor (%d,%d,%d)", D, H, W, fz, fy, fx);
  const int OD = D / fz, OH = H / fy, OW = W / fx;
  const long long total = (long long)B * OD * OH * OW * (C / 4);
  CLX_REQUIRE((long long)B * D * H * W * (C / 4) < (1ll << 31), "clx_maxpool_fwd: tensor too large");
  maxpool_fwd_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      x, y, D, H, W, C / 4, fz, fy, fx, OD, OH, OW, make_dec(C / 4, OW, OH, OD), (uint32_t)total);
  CLX_CHECK_LAUNCH("clx_maxpool_fwd");
  return CLX_OK;
}

extern "C" int clx_maxpool_bwd(const float* x, const float* y, const float* dy_pool,
                               const float* dskip, int ld_skip, int SD, int SH, int SW,
                               int cz, int cy, int cx, float* dx, int B, int D, int H,
                               int W, int C, int fz, int fy, int fx, clx_stream stream) {
  CLX_REQUIRE(x && y && dy_pool && dx, "clx_maxpool_bwd: null pointer");
  CLX_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "clx_maxpool_bwd: bad extents");
  CLX_REQUIRE(D % fz == 0 && H % fy == 0 && W % fx == 0, "clx_maxpool_bwd: extent not divisible");
  CLX_REQUIRE(dskip == nullptr || (ld_skip % 4 == 0 && ld_skip >= C), "clx_maxpool_bwd: bad ld_skip");
  const long long total = (long long)B * D * H * W * (C / 4);
  CLX_REQUIRE(total < (1ll << 31), "clx_maxpool_bwd: tensor too large");
  // extents divide (checked above): every pixel belongs to exactly one window
  const long long wtotal = total / ((long long)fz * fy * fx);
  maxpool_bwd_window_kernel<<<grid_for(wtotal, 256), 256, 0, (hipStream_t)stream>>>(
      x, dy_pool, dskip, ld_skip, SD, SH, SW, cz, cy, cx, dx, D, H, W, C / 4, fz, fy, fx,
      make_dec(C / 4, W / fx, H / fy, D / fz), (uint32_t)wtotal);
  CLX_CHECK_LAUNCH("clx_maxpool_bwd");
  return CLX_OK;
}

extern "C" int clx_upsample_bwd(const float* dcat, int ld_cat, int coff, int LD, int LH,
                                int LW, int oz, int oy, int ox, const float* y, float* dy,
                                int B, int D, int H, int W, int C, int fz, int fy, int fx,
                                clx_stream stream) {
  CLX_REQUIRE(dcat && y && dy, "clx_upsample_bwd: null pointer");
  CLX_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "clx_upsample_bwd: bad extents");
  CLX_REQUIRE(ld_cat % 4 == 0 && coff % 4 == 0 && coff + C <= ld_cat, "clx_upsample_bwd: bad ld/coff");
  const long long total = (long long)B * D * H * W * (C / 4);
  CLX_REQUIRE(total < (1ll << 31), "clx_upsample_bwd: tensor too large");
  upsample_bwd_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      dcat, ld_cat, coff, LD, LH, LW, oz, oy, ox, y, dy, D, H, W, C / 4, fz, fy, fx, make_dec(C / 4, W, H, D),
      (uint32_t)total);
  CLX_CHECK_LAUNCH("clx_upsample_bwd");
  return CLX_OK;
}

static int noise_stats_launch(const float* preds, float* out, int T, int C, long long n, unsigned int* minmax,
                              hipStream_t st, int* grid_out = nullptr) {
  int g = grid_for(n, 256);
  if (minmax != nullptr && g > NOISE_MM_BLOCKS) g = NOISE_MM_BLOCKS;
  if (grid_out) *grid_out = g;
  if (T <= 32)
    CLX_LAUNCH_KIND(CLX_PROF_NOISE_STATS, (noise_stats_reg_kernel<32>), dim3(g), dim3(256), 0, st, preds, out, T, C, n, minmax);
  else if (T <= 64)
    CLX_LAUNCH_KIND(CLX_PROF_NOISE_STATS, (noise_stats_reg_kernel<64>), dim3(g), dim3(256), 0, st, preds, out, T, C, n, minmax);
  else
    CLX_LAUNCH_KIND(CLX_PROF_NOISE_STATS, noise_stats_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, preds, out, T, C, n);
  return CLX_OK;
}

extern "C" int clx_noise_stats(const float* preds, float* out, int T, int C, long long n,
                               clx_stream stream) {
  CLX_REQUIRE(preds && out && T > 0 && C > 0 && n > 0, "clx_noise_stats: bad arguments");
  noise_stats_launch(preds, out, T, C, n, nullptr, (hipStream_t)stream);
  CLX_CHECK_LAUNCH("clx_noise_stats");
  return CLX_OK;
}

extern "C" int clx_noise_stats_minmax(const float* preds, float* out, int T, int C, long long n, float* std_minmax,
                                      int init, clx_stream stream) {
  CLX_REQUIRE(preds && out && std_minmax && T > 0 && C > 0 && n > 0, "clx_noise_stats_minmax: bad arguments");
  CLX_REQUIRE(T <= 64, "clx_noise_stats_minmax: at most 64 predictions per pixel (32 noise iterations)");
  hipStream_t st = (hipStream_t)stream;
  int g = 0;
  noise_stats_launch(preds, out, T, C, n, (unsigned int*)std_minmax, st, &g);
  CLX_LAUNCH_KIND(CLX_PROF_NOISE_STATS, noise_minmax_final, dim3(1), dim3(256), 0, st, (unsigned int*)std_minmax, g, init);
  CLX_CHECK_LAUNCH("clx_noise_stats_minmax");
  return CLX_OK;
}

// ---------------------------------------------------------------------------------------------
// The launches PyTorch used to stand in for on the hot path (round 5): salt / pepper injection, the zero fills of the
// accumulators a step adds into, the row gather of the mean-shift subsample.
// ---------------------------------------------------------------------------------------------
namespace {

// out[t][i] = rnd[t][i] <= p ? (t < n_half ? 0.5 : 1.0) : raw[i]      [cellulus/models/unet.py:75-88]
template <int V>
__global__ __launch_bounds__(256) void noise_inject_kernel(const float* __restrict__ rnd, const float* __restrict__ raw,
                                                           float* __restrict__ out, int T, int n_half, long long nv,
                                                           float p) {
  typedef float vec __attribute__((ext_vector_type(V)));
  const long long total = (long long)T * nv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int t = (int)(i / nv);
    const long long j = i - (long long)t * nv;
    const float val = t < n_half ? 0.5f : 1.0f;
    const vec r = *reinterpret_cast<const vec*>(rnd + i * V);
    vec x = *reinterpret_cast<const vec*>(raw + j * V);
#pragma unroll
    for (int e = 0; e < V; ++e)
      if (r[e] <= p) x[e] = val;
    *reinterpret_cast<vec*>(out + i * V) = x;
  }
}

struct ZeroManyP {
  unsigned long long base[8];
  unsigned long long first16[9];   // prefix of the buffers' 16-byte units
  unsigned int tail_words[8];      // 4-byte words behind a buffer's last whole unit
  int count;
};

__global__ __launch_bounds__(256) void zero_many_kernel(ZeroManyP p) {
  const unsigned long long total = p.first16[p.count];
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * 256) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) k += (q < p.count && i >= p.first16[q]) ? 1 : 0;
    reinterpret_cast<f32x4*>(p.base[k])[i - p.first16[k]] = z;
  }
  if (blockIdx.x == 0 && threadIdx.x < 32) {
    const int k = threadIdx.x >> 2, w = threadIdx.x & 3;
    if (k < p.count && (unsigned int)w < p.tail_words[k])
      reinterpret_cast<float*>(p.base[k])[(p.first16[k + 1] - p.first16[k]) * 4 + w] = 0.f;
  }
}

__global__ __launch_bounds__(256) void gather_rows_f64_kernel(const double* __restrict__ src, const int* __restrict__ sel,
                                                              long long n, int nd, double* __restrict__ dst) {
  const long long total = n * nd;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / nd;
    dst[i] = src[(long long)sel[r] * nd + (i - r * nd)];
  }
}

}  // namespace

namespace {
constexpr int EXTENT_BLOCKS = 256;

// per-column minimum and maximum of a (n, ND) float64 array: block partials [block][2 * ND], then one block over them
template <int ND>
__global__ __launch_bounds__(256) void rows_extent_kernel(const double* __restrict__ src, long long n, double* __restrict__ partial) {
  __shared__ double red[4][2 * ND];
  double lo[ND], hi[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) { lo[d] = __builtin_huge_val(); hi[d] = -__builtin_huge_val(); }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const double v = src[i * ND + d];
      lo[d] = fmin(lo[d], v);
      hi[d] = fmax(hi[d], v);
    }
#pragma unroll
  for (int d = 0; d < ND; ++d)
    for (int o = 32; o > 0; o >>= 1) {
      lo[d] = fmin(lo[d], __shfl_xor(lo[d], o, 64));
      hi[d] = fmax(hi[d], __shfl_xor(hi[d], o, 64));
    }
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int d = 0; d < ND; ++d) { red[threadIdx.x >> 6][d] = lo[d]; red[threadIdx.x >> 6][ND + d] = hi[d]; }
  __syncthreads();
  if (threadIdx.x < 2 * ND) {
    const int d = threadIdx.x;
    double v = red[0][d];
    for (int w = 1; w < 4; ++w) v = d < ND ? fmin(v, red[w][d]) : fmax(v, red[w][d]);
    partial[(size_t)blockIdx.x * 2 * ND + d] = v;
  }
}

template <int ND>
__global__ __launch_bounds__(64) void rows_extent_final(const double* __restrict__ partial, int nblocks, double* __restrict__ out) {
  const int d = threadIdx.x;
  if (d >= 2 * ND) return;
  double v = partial[d];
  for (int b = 1; b < nblocks; ++b) v = d < ND ? fmin(v, partial[(size_t)b * 2 * ND + d]) : fmax(v, partial[(size_t)b * 2 * ND + d]);
  out[d] = v;
}
}  // namespace

extern "C" int clx_rows_extent_f64(const double* src, long long n, int width, double* extent, clx_stream stream) {
  CLX_REQUIRE(src && extent && n > 0 && (width == 2 || width == 3), "clx_rows_extent_f64: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  long long g = (n + 255) / 256;
  g = g > EXTENT_BLOCKS ? EXTENT_BLOCKS : g;
  // one block: its partial IS the result
  double* partial = g == 1 ? extent : extent + 2 * width;
  if (width == 2) rows_extent_kernel<2><<<(int)g, 256, 0, st>>>(src, n, partial);
  else rows_extent_kernel<3><<<(int)g, 256, 0, st>>>(src, n, partial);
  if (g > 1) {
    if (width == 2) rows_extent_final<2><<<1, 64, 0, st>>>(partial, (int)g, extent);
    else rows_extent_final<3><<<1, 64, 0, st>>>(partial, (int)g, extent);
  }
  CLX_CHECK_LAUNCH("clx_rows_extent_f64");
  return CLX_OK;
}

extern "C" int clx_noise_inject(const float* rnd, const float* raw, float* out, int T, int n_half, long long n,
                                float p, clx_stream stream) {
  CLX_REQUIRE(rnd && raw && out && T > 0 && n > 0 && n_half >= 0 && n_half <= T, "clx_noise_inject: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = n % 4 == 0 && ((((uintptr_t)rnd | (uintptr_t)raw | (uintptr_t)out) & 15) == 0);
  if (vec)
    noise_inject_kernel<4><<<grid_for((long long)T * (n / 4), 256), 256, 0, st>>>(rnd, raw, out, T, n_half, n / 4, p);
  else
    noise_inject_kernel<1><<<grid_for((long long)T * n, 256), 256, 0, st>>>(rnd, raw, out, T, n_half, n, p);
  CLX_CHECK_LAUNCH("clx_noise_inject");
  return CLX_OK;
}

extern "C" int clx_zero_many(void* const* buffers, const long long* nbytes, int count, clx_stream stream) {
  CLX_REQUIRE(count >= 0 && (count == 0 || (buffers && nbytes)), "clx_zero_many: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  for (int first = 0; first < count; first += 8) {
    ZeroManyP p;
    p.count = 0;
    p.first16[0] = 0;
    for (int k = first; k < count && k < first + 8; ++k) {
      CLX_REQUIRE(nbytes[k] >= 0 && nbytes[k] % 4 == 0, "clx_zero_many: sizes are multiples of 4 bytes");
      if (nbytes[k] == 0) continue;
      CLX_REQUIRE(buffers[k] && ((uintptr_t)buffers[k] & 15) == 0, "clx_zero_many: buffers are 16-byte aligned");
      const int q = p.count++;
      p.base[q] = (unsigned long long)(uintptr_t)buffers[k];
      p.first16[q + 1] = p.first16[q] + (unsigned long long)nbytes[k] / 16;
      p.tail_words[q] = (unsigned int)(nbytes[k] % 16) / 4;
    }
    if (p.count == 0) continue;
    for (int q = p.count; q < 8; ++q) { p.base[q] = 0; p.first16[q + 1] = p.first16[p.count]; p.tail_words[q] = 0; }
    long long g = (long long)((p.first16[p.count] + 255) / 256);
    g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
    zero_many_kernel<<<(int)g, 256, 0, st>>>(p);
  }
  CLX_CHECK_LAUNCH("clx_zero_many");
  return CLX_OK;
}

extern "C" int clx_gather_rows_f64(const double* src, const int* rows, long long n, int width, double* dst,
                                   clx_stream stream) {
  CLX_REQUIRE(n >= 0 && width > 0 && (n == 0 || (src && rows && dst)), "clx_gather_rows_f64: bad arguments");
  if (n == 0) return CLX_OK;
  gather_rows_f64_kernel<<<grid_for(n * width, 256), 256, 0, (hipStream_t)stream>>>(src, rows, n, width, dst);
  CLX_CHECK_LAUNCH("clx_gather_rows_f64");
  return CLX_OK;
}

static int subpixel_launch(bool to_space, const float* src, float* dst, int ld_lo, int ld_hi, int B,
                           int D, int H, int W, int N, int fz, int fy, int fx, hipStream_t st,
                           const char* who) {
  CLX_REQUIRE(src && dst, "%s: null pointer", who);
  CLX_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && N > 0 && N % 4 == 0, "%s: bad extents", who);
  CLX_REQUIRE(fz >= 1 && fy >= 1 && fx >= 1, "%s: bad factors", who);
  CLX_REQUIRE(ld_lo % 4 == 0 && ld_lo >= fz * fy * fx * N && ld_hi % 4 == 0 && ld_hi >= N, "%s: bad strides", who);
  const long long total = (long long)B * D * H * W * fz * fy * fx * (N / 4);
  if (to_space)
    subpixel_kernel<true><<<grid_for(total, 256), 256, 0, st>>>(src, dst, ld_lo, ld_hi, D, H, W, N / 4, fz, fy, fx, total);
  else
    subpixel_kernel<false><<<grid_for(total, 256), 256, 0, st>>>(src, dst, ld_lo, ld_hi, D, H, W, N / 4, fz, fy, fx, total);
  CLX_CHECK_LAUNCH(who);
  return CLX_OK;
}

extern "C" int clx_depth_to_space(const float* lo, int ld_lo, float* hi, int ld_hi, int B, int D, int H,
                                  int W, int N, int fz, int fy, int fx, clx_stream stream) {
  return subpixel_launch(true, lo, hi, ld_lo, ld_hi, B, D, H, W, N, fz, fy, fx, (hipStream_t)stream,
                         "clx_depth_to_space");
}

extern "C" int clx_space_to_depth(const float* hi, int ld_hi, float* lo, int ld_lo, int B, int D, int H,
                                  int W, int N, int fz, int fy, int fx, clx_stream stream) {
  return subpixel_launch(false, hi, lo, ld_lo, ld_hi, B, D, H, W, N, fz, fy, fx, (hipStream_t)stream,
                         "clx_space_to_depth");
}


// ---------------------------------------------------------------------------------------------
// Reproducible column sums (the bias gradient of a convolution, db[n] = sum_m dY[m][n]) — the
// fused forms accumulate it with LDS and global float atomics, whose order changes from run to run.
// Stage 1: block b sums a contiguous range of rows (lane = 4 channels, the block's rows strided
// over its 256 / (N/4) row groups, combined through LDS in a fixed loop) into partial[b][N];
// stage 2: one block adds the partials in block order.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int COLSUM_BLOCKS = 512;

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int ld, long long M, int N4,
                                                             float* __restrict__ partial) {
  extern __shared__ float cs_red[];                 // [groups][N4 * 4]
  const int groups = 256 / N4 > 0 ? 256 / N4 : 1;   // row groups per block (N4 <= 256)
  const int g = threadIdx.x / N4, c4 = threadIdx.x % N4;
  const long long rows_per_block = (M + gridDim.x - 1) / gridDim.x;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (g < groups)
    for (long long r = r0 + g; r < r1; r += groups) acc += *reinterpret_cast<const f32x4*>(x + r * ld + 4 * c4);
  if (g < groups) *reinterpret_cast<f32x4*>(cs_red + ((size_t)g * N4 + c4) * 4) = acc;
  __syncthreads();
  for (int n = threadIdx.x; n < N4 * 4; n += blockDim.x) {
    float sum = 0.f;
    for (int k = 0; k < groups; ++k) sum += cs_red[(size_t)k * N4 * 4 + n];
    partial[(size_t)blockIdx.x * N4 * 4 + n] = sum;
  }
}

__global__ void colsum_final_kernel(const float* __restrict__ partial, int nblocks, int Np, int N, float* __restrict__ out) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    float sum = 0.f;
    for (int b = 0; b < nblocks; ++b) sum += partial[(size_t)b * Np + n];
    out[n] += sum;      // adds, like the atomic bias path of clx_conv_wgrad (the caller zeroes the gradient once per step)
  }
}
}  // namespace

extern "C" size_t clx_colsum_scratch_bytes(int N) {
  return (size_t)COLSUM_BLOCKS * (size_t)((N + 3) / 4 * 4) * sizeof(float);
}

extern "C" int clx_colsum_ordered(const float* x, int ld, long long M, int N, float* out, void* scratch,
                                  clx_stream stream) {
  CLX_REQUIRE(x && out && scratch, "clx_colsum_ordered: null pointer");
  CLX_REQUIRE(M > 0 && N > 0, "clx_colsum_ordered: bad extents");
  const int N4 = (N + 3) / 4;
  CLX_REQUIRE(N4 <= 256 * 4, "clx_colsum_ordered: more than 4096 channels are not supported");
  CLX_REQUIRE(ld % 4 == 0 && ld >= N4 * 4 && ((uintptr_t)x & 15) == 0, "clx_colsum_ordered: x must be 16-byte aligned with ld %% 4 == 0 covering pad4(N)");
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)((M + 255) / 256);
  if (blocks > COLSUM_BLOCKS) blocks = COLSUM_BLOCKS;
  float* partial = (float*)scratch;
  if (N4 <= 256) {
    const int groups = 256 / N4;
    colsum_partial_kernel<<<blocks, 256, (size_t)groups * N4 * 4 * sizeof(float), st>>>(x, ld, M, N4, partial);
    colsum_final_kernel<<<(N + 255) / 256, 256, 0, st>>>(partial, blocks, N4 * 4, N, out);
  } else {
    // wide tensors: 1024 channels at a time
    for (int c0 = 0; c0 < N4; c0 += 256) {
      const int n4 = N4 - c0 < 256 ? N4 - c0 : 256;
      const int groups = 256 / n4;
      colsum_partial_kernel<<<blocks, 256, (size_t)groups * n4 * 4 * sizeof(float), st>>>(x + 4 * c0, ld, M, n4, partial);
      const int nreal = N - 4 * c0 < n4 * 4 ? N - 4 * c0 : n4 * 4;
      colsum_final_kernel<<<(nreal + 255) / 256, 256, 0, st>>>(partial, blocks, n4 * 4, nreal, out + 4 * c0);
    }
  }
  CLX_CHECK_LAUNCH("clx_colsum_ordered");
  return CLX_OK;
}
