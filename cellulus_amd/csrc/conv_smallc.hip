// Convolution with <= 4 input channels (the network's first layer: raw image ->
// num_fmaps): K = taps * 4 is far too short for an MFMA tile, and the layer is bound by
// writing (forward) / reading (weight gradient) the [M][N] activation once.  Two families:
//   conv_smallc_*  any 1-4 channels, any padding: the forward kernel stages the gathered patch
//                  [pixels][taps*4] and the packed weights in LDS, the weight gradient streams dy
//                  with lane-broadcast tap loads; 16-byte channel runs throughout;
//   conv_grey_*    ONE real channel (clx_conv_desc.c_real == 1) and a valid 3x3 / 3x3x3 kernel — every
//                  BASELINE configuration: one 4-byte patch load per lane, taps by ds_bpermute,
//                  weights / gradient sums in registers (below).
//
// Selected by clx_conv_fwd / clx_conv_wgrad when nsrc == 1, C == 4, no upsampling, no
// accumulate / mask epilogue (those go to the implicit-GEMM kernel whatever the width).
// Replaces nn.Conv{2,3}d(in_channels -> num_fmaps, 3) of l_conv.0.conv_pass.0
// (cellulus/models/unet.py:24-51) and its weight/bias gradient.
#include "clx_common.h"

namespace {

struct SmallP {
  const float* x;        // [pixels][ld_x], 4 channels used
  int ld_x, B, D, H, W;  // stored grid
  int oz, oy, ox;        // crop offset
  int ID, IH, IW, KD, KH, KW, PD, PH, PW, OD, OH, OW;
  int N, M, taps;
  FastDiv dOW, dOH, dOD;
  const float* wpack;    // fwd: [N][taps][4]
  const float* bias;
  float* out;            // fwd: [M][ld_out]
  int ld_out, relu;
  const float* dy;       // wgrad: [M][ld_dy]
  int ld_dy;
  float* dwp;            // wgrad: [taps][N][4]
  float* dbias;
  int tiles_per_block;
  int RX, items;         // grey kernels: runs of 4 output pixels per row, runs in total
  FastDiv dRX;
};

constexpr int PT = 64;          // pixels per tile
constexpr int MAX_TAPS = 27;

// gathers the input patch of PT output pixels into Xs[PT][taps*4] (zero outside the image)
template <int PTILE>
__device__ __forceinline__ void stage_patch(const SmallP& p, int m0, float* Xs) {
  static_assert(256 % PTILE == 0, "a thread keeps one pixel of the tile");
  const int K4 = p.taps;                 // float4s per pixel
  // 256 % PTILE == 0: every thread serves ONE pixel (decoded once) and walks over the taps
  const int pl = threadIdx.x % PTILE;
  const uint32_t m = (uint32_t)(m0 + pl);
  const bool live = m < (uint32_t)p.M;
  const uint32_t q1 = fdiv(m, p.dOW);
  const int ox = (int)(m - q1 * p.OW);
  const uint32_t q2 = fdiv(q1, p.dOH);
  const int oy = (int)(q1 - q2 * p.OH);
  const uint32_t q3 = fdiv(q2, p.dOD);
  const int oz = (int)(q2 - q3 * p.OD);
  const int b = (int)q3;
  int tx = 0, ty = 0, tz = 0;
  {
    const int tap = threadIdx.x / PTILE;
    tx = tap % p.KW; ty = (tap / p.KW) % p.KH; tz = tap / (p.KW * p.KH);
  }
  constexpr int STEP = 256 / PTILE;      // taps advanced per iteration
  for (int tap = threadIdx.x / PTILE; tap < K4; tap += STEP) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int lz = oz + tz - p.PD, ly = oy + ty - p.PH, lx = ox + tx - p.PW;
    if (live && (unsigned)lz < (unsigned)p.ID && (unsigned)ly < (unsigned)p.IH && (unsigned)lx < (unsigned)p.IW) {
      const long long pix = (((long long)b * p.D + lz + p.oz) * p.H + ly + p.oy) * p.W + lx + p.ox;
      v = *reinterpret_cast<const f32x4*>(p.x + pix * p.ld_x);
    }
    *reinterpret_cast<f32x4*>(&Xs[(pl * K4 + tap) * 4]) = v;
    tx += STEP;                           // advance (tz, ty, tx) by STEP taps without divisions
    while (tx >= p.KW) { tx -= p.KW; if (++ty == p.KH) { ty = 0; ++tz; } }
  }
}

// NG = N-tile / 4 channel groups per block (power of two <= 64); each thread owns 4 output
// channels of PT / (256 / NG) pixels.
template <int NG>
__global__ __launch_bounds__(256) void conv_smallc_fwd_kernel(const SmallP p) {
  constexpr int PL = 256 / NG;          // pixel lanes
  constexpr int PPT = PT / PL;          // pixels per thread
  extern __shared__ float smem[];
  const int K = p.taps * 4;
  float* Ws = smem;                     // [K][NG*4]
  float* Xs = smem + K * NG * 4;        // [PT][K]
  const int n0 = blockIdx.y * NG * 4;
  const int ng = threadIdx.x % NG, pl = threadIdx.x / NG;
  // Channels whose packed weights are all zero (the padding of a 1-, 2- or 3-channel image up to 4)
  // are skipped in the contraction below: 4x less LDS traffic and FMAs for grey-scale input.
  __shared__ int used_channels;
  if (threadIdx.x == 0) used_channels = 0;
  __syncthreads();
  int mine_used = 0;
  for (int idx = threadIdx.x; idx < K * NG * 4; idx += 256) {
    const int n = idx / K, k = idx % K;
    const float wv = (n0 + n < p.N) ? p.wpack[(size_t)(n0 + n) * K + k] : 0.f;
    Ws[k * NG * 4 + n] = wv;
    if (wv != 0.f) mine_used |= 1 << (k & 3);
  }
  if (mine_used) atomicOr(&used_channels, mine_used);
  const int n = n0 + ng * 4;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (p.bias && n < p.N) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = (n + e < p.N) ? p.bias[n + e] : 0.f;
  }
  for (int t = 0; t < p.tiles_per_block; ++t) {
    const int m0 = (blockIdx.x * p.tiles_per_block + t) * PT;
    if (m0 >= p.M) break;
    __syncthreads();
    stage_patch<PT>(p, m0, Xs);
    __syncthreads();
    f32x4 acc[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) acc[i] = bv;
    const int used = used_channels;
    for (int c = 0; c < 4; ++c) {
      if (!((used >> c) & 1)) continue;
      for (int k = c; k < K; k += 4) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(&Ws[k * NG * 4 + ng * 4]);
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
          const float xv = Xs[(pl + i * PL) * K + k];
          acc[i] += xv * w;
        }
      }
    }
    if (n < p.N) {
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int m = m0 + pl + i * PL;
        if (m >= p.M) continue;
        f32x4 v = acc[i];
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        float* dst = p.out + (size_t)m * p.ld_out + n;
        if (n + 3 < p.N) *reinterpret_cast<f32x4*>(dst) = v;
        else
          for (int e = 0; e < 4 && n + e < p.N; ++e) dst[e] = v[e];
      }
    }
  }
}

// Weight gradient, streaming form: the layer is bound
// by ONE read of dy [M][N], so no LDS staging and no barriers in the pixel loop — lane = 4 output
// channels, the G lanes of a pixel read its dy row as one coalesced run and the (<= 9) input
// patch values as same-address loads; partial sums meet in LDS once per block, then one global
// atomic per (tap, n, c) and block.
template <int G>
__global__ __launch_bounds__(256) void conv_smallc_wgrad_stream_kernel(const SmallP p) {
  constexpr int Q = 256 / G;            // pixels in flight per block
  constexpr int MT = 9;
  extern __shared__ float red[];        // [taps][4][G*4] + [G*4] bias
  const int g = threadIdx.x % G, q = threadIdx.x / G;
  const int NT = G * 4;
  const int n0 = blockIdx.y * NT, n = n0 + g * 4;
  // taps are handled in groups of <= 9 (blockIdx.z): a 3x3x3 kernel makes three passes over dy,
  // still a third of the time of staging everything through LDS
  const int tap0 = blockIdx.z * MT;
  const int ntap = p.taps - tap0 < MT ? p.taps - tap0 : MT;
  const int nred = ntap * 4 * NT + NT;
  for (int i = threadIdx.x; i < nred; i += 256) red[i] = 0.f;
  f32x4 acc[MT][4];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const long long per = (long long)p.tiles_per_block;         // pixels per block
  const long long m_begin = (long long)blockIdx.x * per;
  const long long m_end = m_begin + per < p.M ? m_begin + per : p.M;
  const bool live = n < p.N;
  // this lane's tap (lane g < taps loads tap g's input vector for its pixel)
  const int gtap = tap0 + (g < ntap ? g : 0);
  const int my_tx = gtap % p.KW - p.PW, my_ty = (gtap / p.KW) % p.KH - p.PH, my_tz = gtap / (p.KW * p.KH) - p.PD;
  for (long long mm = m_begin + q; mm < m_end; mm += Q) {
    const uint32_t m = (uint32_t)mm;
    const uint32_t q1 = fdiv(m, p.dOW);
    const int ox = (int)(m - q1 * p.OW);
    const uint32_t q2 = fdiv(q1, p.dOH);
    const int oy = (int)(q1 - q2 * p.OH);
    const uint32_t q3 = fdiv(q2, p.dOD);
    const int oz = (int)(q2 - q3 * p.OD);
    const int b = (int)q3;
    f32x4 dyv = {0.f, 0.f, 0.f, 0.f};
    if (live) dyv = *reinterpret_cast<const f32x4*>(p.dy + (size_t)m * p.ld_dy + n);
    if (tap0 == 0) bsum += dyv;
    if constexpr (G >= 16) {
      // lane a (< taps) of the pixel's G lanes fetches tap a's input vector: ONE vector load with
      // <= 9 active lanes per pixel instead of 9 same-address loads occupying all 64 lanes
      // (measured: 0.55 ms of the 1.1 ms kernel), then broadcast across the pixel's lanes.
      f32x4 mine = {0.f, 0.f, 0.f, 0.f};
      if (g < ntap) {
        const int lz = oz + my_tz, ly = oy + my_ty, lx = ox + my_tx;
        if ((unsigned)lz < (unsigned)p.ID && (unsigned)ly < (unsigned)p.IH && (unsigned)lx < (unsigned)p.IW) {
          const long long pix = (((long long)b * p.D + lz + p.oz) * p.H + ly + p.oy) * p.W + lx + p.ox;
          mine = *reinterpret_cast<const f32x4*>(p.x + pix * p.ld_x);
        }
      }
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        if (a < ntap) {
          f32x4 xv;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            xv[c] = __shfl(mine[c], (int)(threadIdx.x & 63 & ~(G - 1)) + a, 64);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[a][c] += xv[c] * dyv;
        }
      }
    } else {
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        if (a < ntap) {
          const int tap = tap0 + a;
          const int lz = oz + tap / (p.KW * p.KH) - p.PD, ly = oy + (tap / p.KW) % p.KH - p.PH, lx = ox + tap % p.KW - p.PW;
          f32x4 xv = {0.f, 0.f, 0.f, 0.f};
          if ((unsigned)lz < (unsigned)p.ID && (unsigned)ly < (unsigned)p.IH && (unsigned)lx < (unsigned)p.IW) {
            const long long pix = (((long long)b * p.D + lz + p.oz) * p.H + ly + p.oy) * p.W + lx + p.ox;
            xv = *reinterpret_cast<const f32x4*>(p.x + pix * p.ld_x);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[a][c] += xv[c] * dyv;
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    if (a < ntap) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&red[(a * 4 + c) * NT + g * 4 + e], acc[a][c][e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) atomicAdd(&red[ntap * 4 * NT + g * 4 + e], bsum[e]);
  __syncthreads();
  for (int i = threadIdx.x; i < ntap * 4 * NT; i += 256) {
    const int nl = i % NT, c = (i / NT) & 3, tap = tap0 + i / (4 * NT);
    if (n0 + nl < p.N) atomicAdd(p.dwp + ((size_t)tap * p.N + n0 + nl) * 4 + c, red[i]);
  }
  if (p.dbias && tap0 == 0)
    for (int i = threadIdx.x; i < NT; i += 256)
      if (n0 + i < p.N) atomicAdd(p.dbias + n0 + i, red[ntap * 4 * NT + i]);
}

// ---------------------------------------------------------------------------------------------
// ONE real input channel (grey-scale raw image, c_real == 1) and a valid 3x3 / 3x3x3 kernel —
// the first layer of every configuration BASELINE.json names.  A wavefront works on a run of four
// consecutive output pixels along x: lane = (px, cq), pixel px of the run and channel quad cq
// (16 quads = 64 output channels).  The run's input patch is KD*3 rows of 6 values: lane L < KD*18
// fetches value L (ONE 4-byte load per lane and run instead of taps 16-byte loads per pixel, three
// quarters of them padding), and every lane picks its 3 x 3 (x 3) taps out of the patch with
// ds_bpermute.  Weights (forward) / weight-gradient sums (backward) live in registers for the
// whole kernel, so per run a wave issues 1 load, TAPS permutes, 4*TAPS FMAs per lane and one
// 16-byte-per-lane store (forward) or load (backward) of the [M][N] activation: HBM-bound on that
// tensor.  No LDS staging, no barriers in the loop, loads run two runs ahead; addresses of dead
// prefetches / of the patch columns past the row end are clamped instead of predicated (a guarded
// load in the loop costs a vmcnt(0)) — such columns only ever reach pixels past the
// row end, which are not stored (forward) or enter with dy = 0 (backward).
struct Run {
  int ox0, m0;           // first output pixel of the run: x position, linear index
  long long base;        // its input pixel (stored grid, crop applied)
};

__device__ __forceinline__ Run decode_run(const SmallP& p, uint32_t item) {
  const uint32_t q1 = fdiv(item, p.dRX);
  const int rx = (int)(item - q1 * p.RX);
  const uint32_t q2 = fdiv(q1, p.dOH);
  const int oy = (int)(q1 - q2 * p.OH);
  const uint32_t q3 = fdiv(q2, p.dOD);
  const int oz = (int)(q2 - q3 * p.OD);
  const int b = (int)q3;
  Run r;
  r.ox0 = rx * 4;
  r.m0 = ((b * p.OD + oz) * p.OH + oy) * p.OW + r.ox0;
  r.base = (((long long)b * p.D + oz + p.oz) * p.H + oy + p.oy) * p.W + r.ox0 + p.ox;
  return r;
}

template <int KD>
struct GreyLane {
  static constexpr int R = KD * 3, NP = R * 6, TAPS = R * 3;
  int px, cq, col;
  long long poff;
  __device__ __forceinline__ GreyLane(const SmallP& p) {
    const int lane = threadIdx.x & 63;
    px = lane >> 4; cq = lane & 15;
    const int l = lane < NP ? lane : 0;          // lanes past the patch re-read value 0 (unused)
    const int r = l / 6;
    col = l - r * 6;
    const int tz = r / 3, ty = r - tz * 3;
    poff = ((long long)tz * p.H + ty) * p.W;
  }
  __device__ __forceinline__ float load_patch(const SmallP& p, const Run& r) const {
    const int c = r.ox0 + col < p.IW ? col : p.IW - 1 - r.ox0;
    return p.x[(r.base + poff + c) * p.ld_x];
  }
  // input value under tap (row, tx) of this lane's pixel
  __device__ __forceinline__ float tap(float pv, int row, int tx) const {
    return __shfl(pv, row * 6 + tx + px, 64);
  }
};

template <int KD>
__global__ __launch_bounds__(256) void conv_grey_fwd_kernel(const SmallP p) {
  using L = GreyLane<KD>;
  const L ln(p);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.y * 64 + ln.cq * 4;
  const bool live_n = n < p.N;
  f32x4 w[L::TAPS];
#pragma unroll
  for (int t = 0; t < L::TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) w[t][e] = live_n ? p.wpack[((size_t)(n + e) * L::TAPS + t) * 4] : 0.f;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (p.bias && live_n) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
  const uint32_t last = (uint32_t)p.items - 1, stride = gridDim.x * 4;
  uint32_t item = blockIdx.x * 4 + wave;
  // three register sets in rotation, the loop unrolled by three so that no set is ever copied
  // (a copy of a pending load makes the wait-count pass drain the queue: prefetch distance 1)
  Run r[3];
  float pv[3];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    r[s] = decode_run(p, item + s * stride < last ? item + s * stride : last);
    pv[s] = ln.load_patch(p, r[s]);
  }
  while (true) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (item > last) return;
      const int s2 = (s + 2) % 3;
      const uint32_t nxt = item + 2 * stride;
      r[s2] = decode_run(p, nxt < last ? nxt : last);
      pv[s2] = ln.load_patch(p, r[s2]);
      f32x4 acc = bv;
#pragma unroll
      for (int row = 0; row < L::R; ++row)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) acc += ln.tap(pv[s], row, tx) * w[row * 3 + tx];
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.f);
      }
      if (live_n && r[s].ox0 + ln.px < p.OW)
        *reinterpret_cast<f32x4*>(p.out + (size_t)(r[s].m0 + ln.px) * p.ld_out + n) = acc;
      item += stride;
    }
  }
}

template <int KD>
__global__ __launch_bounds__(256) void conv_grey_wgrad_kernel(const SmallP p) {
  using L = GreyLane<KD>;
  __shared__ float red[(L::TAPS + 1) * 64];
  const L ln(p);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.y * 64, n = n0 + ln.cq * 4;
  const bool live_n = n < p.N;
  const int nld = live_n ? n : 0;
  for (int i = threadIdx.x; i < (L::TAPS + 1) * 64; i += 256) red[i] = 0.f;
  f32x4 acc[L::TAPS];
#pragma unroll
  for (int t = 0; t < L::TAPS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const uint32_t last = (uint32_t)p.items - 1, stride = gridDim.x * 4;
  uint32_t item = blockIdx.x * 4 + wave;
  auto load_dy = [&](const Run& r) {
    const int m = r.ox0 + ln.px < p.OW ? r.m0 + ln.px : r.m0;
    return *reinterpret_cast<const f32x4*>(p.dy + (size_t)m * p.ld_dy + nld);
  };
  Run r[3];
  float pv[3];
  f32x4 dy[3];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    r[s] = decode_run(p, item + s * stride < last ? item + s * stride : last);
    pv[s] = ln.load_patch(p, r[s]);
    dy[s] = load_dy(r[s]);
  }
  bool more = item <= last;
  while (more) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (more) {
        const int s2 = (s + 2) % 3;
        const uint32_t nxt = item + 2 * stride;
        r[s2] = decode_run(p, nxt < last ? nxt : last);
        pv[s2] = ln.load_patch(p, r[s2]);
        dy[s2] = load_dy(r[s2]);
        const bool ok = live_n && r[s].ox0 + ln.px < p.OW;
        f32x4 dyv;
#pragma unroll
        for (int e = 0; e < 4; ++e) dyv[e] = ok ? dy[s][e] : 0.f;
        bsum += dyv;
#pragma unroll
        for (int row = 0; row < L::R; ++row)
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) acc[row * 3 + tx] += ln.tap(pv[s], row, tx) * dyv;
        item += stride;
        more = item <= last;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < L::TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(&red[t * 64 + ln.cq * 4 + e], acc[t][e]);
#pragma unroll
  for (int e = 0; e < 4; ++e) atomicAdd(&red[L::TAPS * 64 + ln.cq * 4 + e], bsum[e]);
  __syncthreads();
  for (int i = threadIdx.x; i < L::TAPS * 64; i += 256) {
    const int nl = i & 63, t = i >> 6;
    if (n0 + nl < p.N) atomicAdd(p.dwp + ((size_t)t * p.N + n0 + nl) * 4, red[i]);
  }
  if (p.dbias && threadIdx.x < 64 && n0 + threadIdx.x < p.N)
    atomicAdd(p.dbias + n0 + threadIdx.x, red[L::TAPS * 64 + threadIdx.x]);
}

// grey kernels apply: one real channel, valid 3x3 or 3x3x3 kernel, channel quads stored whole
// conv_grey_fwd_kernel's arithmetic for a LIST of output pixels of a planar one-channel image batch (the changed
// rows of the noisy copies of infer mode): out[r][n] = act(bias[n] + sum over the taps in the order
// t = (dz * 3 + dy) * 3 + dx of x[...] * w[n][t]) with the same fused multiply-adds in the same order — the bits
// conv_grey_fwd_kernel writes for that pixel (tests/test_gpu_unet.py compares them).  lane = (px, cq): four listed
// pixels per wavefront, channel quad cq of the 64 channels of blockIdx.y.
template <int KD>
__global__ __launch_bounds__(256) void conv_grey_rows_kernel(const float* __restrict__ x, int D, int H, int W, int OD,
                                                             int OH, int OW, const int* __restrict__ rows, long long n,
                                                             const float* __restrict__ wpack,
                                                             const float* __restrict__ bias, int relu, int N,
                                                             float* __restrict__ out, int ld_out) {
  constexpr int R = KD * 3, TAPS = R * 3;
  const int lane = threadIdx.x & 63, px = lane >> 4, cq = lane & 15;
  const int nn = blockIdx.y * 64 + cq * 4;
  const bool live_n = nn < N;
  f32x4 w[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) w[t][e] = live_n ? wpack[((size_t)(nn + e) * TAPS + t) * 4] : 0.f;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias && live_n) bv = *reinterpret_cast<const f32x4*>(bias + nn);
  const long long npo = (long long)OD * OH * OW;
  const long long groups = (n + 3) / 4;
  for (long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); g < groups; g += (long long)gridDim.x * 4) {
    const long long r = g * 4 + px;
    const bool live = r < n;
    const long long row = rows[live ? r : n - 1];
    const long long b = row / npo, pix = row - b * npo;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), oz = (int)(pix / ((long long)OW * OH));
    const float* src = x + ((b * D + oz) * H + oy) * (long long)W + ox;
    float xv[TAPS];
#pragma unroll
    for (int rr = 0; rr < R; ++rr)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) xv[rr * 3 + tx] = src[((long long)(rr / 3) * H + (rr % 3)) * W + tx];
    f32x4 acc = bv;
#pragma unroll
    for (int rr = 0; rr < R; ++rr)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) acc += xv[rr * 3 + tx] * w[rr * 3 + tx];
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.f);
    }
    if (live && live_n) *reinterpret_cast<f32x4*>(out + (size_t)r * ld_out + nn) = acc;
  }
}

bool grey_applicable(const clx_conv_desc* d, int ld_act) {
  return d->c_real == 1 && d->KH == 3 && d->KW == 3 && (d->KD == 1 || d->KD == 3) && d->PD == 0 &&
         d->PH == 0 && d->PW == 0 && d->N % 4 == 0 && ld_act % 4 == 0;
}

dim3 grey_grid(SmallP& p) {
  p.RX = cdiv(p.OW, 4);
  p.items = p.B * p.OD * p.OH * p.RX;
  p.dRX = make_fastdiv(p.RX);
  const int ny = cdiv(p.N, 64);
  int nx = 768 / ny;                    // ~3 waves per SIMD over the whole chip
  if (nx < 1) nx = 1;
  if (nx > cdiv(p.items, 4)) nx = cdiv(p.items, 4);
  return dim3(nx, ny);
}

bool fill(const clx_conv_desc* d, SmallP& p) {
  const clx_src& S = d->src[0];
  p.x = S.ptr; p.ld_x = S.ld; p.B = d->B; p.D = S.D; p.H = S.H; p.W = S.W;
  p.oz = S.oz; p.oy = S.oy; p.ox = S.ox;
  p.ID = d->ID; p.IH = d->IH; p.IW = d->IW;
  p.KD = d->KD; p.KH = d->KH; p.KW = d->KW;
  p.PD = d->PD; p.PH = d->PH; p.PW = d->PW;
  p.OD = d->ID + 2 * d->PD - d->KD + 1;
  p.OH = d->IH + 2 * d->PH - d->KH + 1;
  p.OW = d->IW + 2 * d->PW - d->KW + 1;
  p.N = d->N;
  p.M = d->B * p.OD * p.OH * p.OW;
  p.taps = d->KD * d->KH * d->KW;
  p.dOW = make_fastdiv(p.OW); p.dOH = make_fastdiv(p.OH); p.dOD = make_fastdiv(p.OD);
  return true;
}

int pick_ng(int N) {
  int ng = 4;
  while (ng < 64 && ng * 4 < N) ng *= 2;
  return ng;
}

}  // namespace

// true if the small-channel path applies to this descriptor
bool clx_smallc_applicable(const clx_conv_desc* d) {
  if (d->nsrc != 1 || d->accumulate) return false;   // the kernels overwrite their output
  const clx_src& S = d->src[0];
  if (S.C != 4 || S.fz != 1 || S.fy != 1 || S.fx != 1) return false;
  const int taps = d->KD * d->KH * d->KW;
  return taps > 1 && taps <= MAX_TAPS && d->N >= 4;
}

int clx_smallc_fwd(const clx_conv_desc* d, hipStream_t st) {
  SmallP p{};
  fill(d, p);
  p.wpack = d->wpack; p.bias = d->bias; p.out = d->out; p.ld_out = d->ld_out; p.relu = d->relu;
  if (grey_applicable(d, d->ld_out) && ((uintptr_t)d->out & 15) == 0 && (!d->bias || ((uintptr_t)d->bias & 15) == 0)) {
    const dim3 grid = grey_grid(p);
    if (d->KD == 1) conv_grey_fwd_kernel<1><<<grid, 256, 0, st>>>(p);
    else conv_grey_fwd_kernel<3><<<grid, 256, 0, st>>>(p);
    return CLX_OK;
  }
  const int ng = pick_ng(p.N);
  const int K = p.taps * 4;
  const int tiles = cdiv(p.M, PT);
  p.tiles_per_block = tiles > 4096 ? 4 : 1;
  const dim3 grid(cdiv(tiles, p.tiles_per_block), cdiv(p.N, ng * 4));
  const size_t lds = (size_t)(K * ng * 4 + PT * K) * sizeof(float);
  switch (ng) {
    case 4: conv_smallc_fwd_kernel<4><<<grid, 256, lds, st>>>(p); break;
    case 8: conv_smallc_fwd_kernel<8><<<grid, 256, lds, st>>>(p); break;
    case 16: conv_smallc_fwd_kernel<16><<<grid, 256, lds, st>>>(p); break;
    case 32: conv_smallc_fwd_kernel<32><<<grid, 256, lds, st>>>(p); break;
    default: conv_smallc_fwd_kernel<64><<<grid, 256, lds, st>>>(p); break;
  }
  return CLX_OK;
}

int clx_smallc_wgrad(const clx_conv_desc* d, const float* dy, int ld_dy, float* dwpack,
                     float* dbias, hipStream_t st) {
  SmallP p{};
  fill(d, p);
  p.dy = dy; p.ld_dy = ld_dy; p.dwp = dwpack; p.dbias = dbias;
  if (grey_applicable(d, ld_dy) && ((uintptr_t)dy & 15) == 0) {
    const dim3 grid = grey_grid(p);
    if (d->KD == 1) conv_grey_wgrad_kernel<1><<<grid, 256, 0, st>>>(p);
    else conv_grey_wgrad_kernel<3><<<grid, 256, 0, st>>>(p);
    return CLX_OK;
  }
  const int ng = pick_ng(p.N);
  // 183 VGPRs -> 2 waves per SIMD = 2 blocks per CU resident: one round of 512 blocks per tap
  // group, so every block pays its final global atomics once (0.22 ms of 0.62 at 1024 blocks)
  const int blocks = p.M < 512 * 64 ? cdiv(p.M, 64) : 512;
  p.tiles_per_block = cdiv(p.M, blocks);
  const dim3 grid(cdiv(p.M, p.tiles_per_block), cdiv(p.N, ng * 4), cdiv(p.taps, 9));
  const size_t lds = (size_t)((p.taps < 9 ? p.taps : 9) * 4 * ng * 4 + ng * 4) * sizeof(float);
  switch (ng) {
    case 4: conv_smallc_wgrad_stream_kernel<4><<<grid, 256, lds, st>>>(p); break;
    case 8: conv_smallc_wgrad_stream_kernel<8><<<grid, 256, lds, st>>>(p); break;
    case 16: conv_smallc_wgrad_stream_kernel<16><<<grid, 256, lds, st>>>(p); break;
    case 32: conv_smallc_wgrad_stream_kernel<32><<<grid, 256, lds, st>>>(p); break;
    default: conv_smallc_wgrad_stream_kernel<64><<<grid, 256, lds, st>>>(p); break;
  }
  return CLX_OK;
}

extern "C" int clx_grey_rows(const float* x, int B, int ID, int IH, int IW, int KD, const int* rows, long long n,
                             const float* wpack, const float* bias, int relu, int N, float* out, int ld_out,
                             clx_stream stream) {
  CLX_REQUIRE(n >= 0, "clx_grey_rows: bad count");
  if (n == 0) return CLX_OK;
  CLX_REQUIRE(x && rows && wpack && out, "clx_grey_rows: null pointer");
  CLX_REQUIRE(B > 0 && (KD == 1 || KD == 3) && ID >= KD && IH >= 3 && IW >= 3, "clx_grey_rows: a 3x3 or 3x3x3 window");
  CLX_REQUIRE(N > 0 && N % 4 == 0 && ld_out % 4 == 0 && ld_out >= N && ((uintptr_t)out & 15) == 0 &&
                  (!bias || ((uintptr_t)bias & 15) == 0),
              "clx_grey_rows: N, ld_out multiples of 4, aligned out / bias");
  const int OD = ID - KD + 1, OH = IH - 2, OW = IW - 2;
  long long gx = (n + 15) / 16;
  if (gx > 4096) gx = 4096;
  const dim3 grid((unsigned)gx, (unsigned)((N + 63) / 64));
  hipStream_t st = (hipStream_t)stream;
  if (KD == 1) conv_grey_rows_kernel<1><<<grid, 256, 0, st>>>(x, ID, IH, IW, OD, OH, OW, rows, n, wpack, bias, relu, N, out, ld_out);
  else conv_grey_rows_kernel<3><<<grid, 256, 0, st>>>(x, ID, IH, IW, OD, OH, OW, rows, n, wpack, bias, relu, N, out, ld_out);
  CLX_CHECK_LAUNCH("clx_grey_rows");
  return CLX_OK;
}
