// Two consecutive 1x1 convolutions over 64 channels in ONE pass over the pixels (gfx950).
//
// The U-Net of the reference (funlib ConvPass as called from cellulus/models/unet.py:24-51)
// follows every 3x3 convolution by two 1x1 convolutions + ReLU, and the head is two more
// (unet.py:52-63).  Where those layers are 64 channels wide — the whole 3-D benchmark network's
// top level, the last level and the head of the 2-D one — each of them is HBM-bound as a launch
// of its own (K = 64: 0.3 of the matrix pipe, 3.5 TB/s): the intermediate tensor is written by
// one launch and read back by the next, and in the backward pass it is read FOUR times (two data
// gradients, two weight gradients) and its gradient written and read once more.
//
//   forward   y1 = relu(x W1^T + b1),  y2 = [relu](y1 W2^T + b2)        read x, write y1, y2
//   backward  dP1 = (dP2 W2) * (y1 > 0),  dP0 = (dP1 W1) * (x > 0)      read dP2, y1, x; write dP0
//             dW2 += dP2^T y1, db2 += sum dP2, dW1 += dP1^T x, db1 += sum dP1
//             (dP1 never exists in HBM)
//
// Formulation: every product is computed TRANSPOSED, Y^T = W X^T, with the weights as the MFMA's
// A operand (LDS-resident for the whole kernel, fragment order, one ds_read_b128 per four
// v_mfma_f32_32x32x2_f32) and the activations as its B operand straight from the registers the
// global loads filled: lane (pixel i, half h) holds channels {8q + 4h + j} of its pixel, which is
// (a) 16-byte runs of a pixel-major row in HBM, (b) the k order the A fragments are stored in and
// (c) exactly the accumulator layout of the transposed product (row n = 8g + 4h + j of lane
// (i, h), register 4g + j) — so layer 1's result, after bias + ReLU in registers, IS layer 2's B
// operand: no LDS round trip, no barrier, no transpose between the layers.  Only the weight
// gradients (contraction over pixels) need the tile in LDS, pixel-major as it arrives: every wave
// stages its own 32 pixels and accumulates the whole of dW2 and dW1 over them (128 accumulator
// registers — one wave per SIMD has them), so the tile loop has no block barrier at all; the
// waves' sums meet once, at the end of the kernel.
#include "clx_common.h"

#include <stdlib.h>

namespace {

constexpr int LDW = 68;                 // floats per staged pixel row: 64 + 4 (ds_write_b128 of 8 lanes: 8 x 4 banks apart)

// [rows][k] row-major matrix (k contiguous, leading dimension ld) -> LDS in A-fragment order:
// the float4 (row 32 rt + i, k = 8q + 4h .. + 3) goes to slot ((rt * 8 + q) * 64 + h * 32 + i);
// rows >= nrows and k >= ncols are zero.  rt < RT.
template <int RT>
__device__ __forceinline__ void load_matrix(float* lds, const float* __restrict__ w, int nrows, int ncols, int ld) {
  for (int idx = threadIdx.x; idx < RT * 512; idx += blockDim.x) {
    const int row = idx >> 4, c4 = idx & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < nrows && 4 * c4 < ncols) v = *reinterpret_cast<const f32x4*>(w + (size_t)row * ld + 4 * c4);
    const int rt = row >> 5, i = row & 31, q = c4 >> 1, h = c4 & 1;
    *reinterpret_cast<f32x4*>(lds + ((((rt * 8 + q) * 64) + h * 32 + i) << 2)) = v;
  }
}

struct ChainFwdP {
  const float* x; int ld_x; int M;
  const float* w1; const float* b1; float* y1; int ld_y1; unsigned int* gate1; int ld_gate1;
  const float* w2; const float* b2; int N2; int relu2; float* y2; int ld_y2; unsigned int* gate2; int ld_gate2;
  int ntiles;
};

// NT2: 32-row tiles of layer 2's output channels — 2 (64 channels) or 1 (N2 <= 32, the head's last layer)
//
// Global memory is touched in FULL LINES only: a wave instruction reads / writes four whole 256-byte pixel rows
// (lane l: row 4j + l / 16, 16-byte piece l % 16), and the wave's private LDS tile [32][LDW] turns that into the
// fragment layout of the transposed products and back.  (The first version loaded and stored fragment-shaped —
// 32 rows x 32 bytes per instruction — which keeps the texture addresser busy several times longer for the same
// bytes: DESIGN.md 8b.)  The LDS operations of ONE wave execute in order, so its write -> read hand-offs through
// the tile need no barrier.
template <int NT2>
__global__ __launch_bounds__(256, 2) void chain64_fwd_kernel(const ChainFwdP p) {
  __shared__ __attribute__((aligned(16))) float W1s[2 * 2048];
  __shared__ __attribute__((aligned(16))) float W2s[NT2 * 2048];
  __shared__ __attribute__((aligned(16))) float B1s[64];
  __shared__ __attribute__((aligned(16))) float B2s[64];
  __shared__ __attribute__((aligned(16))) float Stage[4][32 * LDW];
  load_matrix<2>(W1s, p.w1, 64, 64, 64);
  load_matrix<NT2>(W2s, p.w2, p.N2, 64, 64);
  if (threadIdx.x < 64) {
    B1s[threadIdx.x] = p.b1 ? p.b1[threadIdx.x] : 0.f;
    B2s[threadIdx.x] = (p.b2 && (int)threadIdx.x < p.N2) ? p.b2[threadIdx.x] : 0.f;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int lr = lane >> 4, lp = lane & 15;            // full-line role: row 4j + lr of the tile, piece lp
  float* stg = Stage[wid];
  const int n2p = (p.N2 + 3) & ~3;
  const int stride = gridDim.x * 4;
  auto load_rows = [&](int tile, f32x4* dst) {         // clamped into the tensor (stores are predicated)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int row = tile * 32 + 4 * j + lr;
      row = row < p.M ? row : p.M - 1;
      dst[j] = *reinterpret_cast<const f32x4*>(p.x + (size_t)row * p.ld_x + 4 * lp);
    }
  };
  // fragment layout (lane = pixel i, half h: channels 8q + 4h ..) -> full lines, through the wave's tile
  auto store_rows = [&](const f32x4* frag, float* dst, int ld, int tile) {
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x4*>(stg + i * LDW + 8 * q + 4 * h) = frag[q];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (4 * j + lr) * LDW + 4 * lp);
      const int row = tile * 32 + 4 * j + lr;
      if (row < p.M) *reinterpret_cast<f32x4*>(dst + (size_t)row * ld + 4 * lp) = v;
    }
    __builtin_amdgcn_wave_barrier();
  };
  f32x4 xl[8], xn[8];
  int tile = blockIdx.x * 4 + wid;
  if (tile < p.ntiles) load_rows(tile, xl);
  for (; tile < p.ntiles; tile += stride) {
    // the weights stay in LDS (a hoisted copy would be 128 VGPRs): nothing moves across this point
    asm volatile("" ::: "memory");
    load_rows(tile + stride < p.ntiles ? tile + stride : tile, xn);      // next tile's rows, in flight during this one
    const int row = tile * 32 + i;
    const bool ok = row < p.M;
    // full lines -> fragments
    f32x4 xb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(stg + (4 * j + lr) * LDW + 4 * lp) = xl[j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 8; ++q) xb[q] = *reinterpret_cast<const f32x4*>(stg + i * LDW + 8 * q + 4 * h);
    __builtin_amdgcn_wave_barrier();
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(W1s + (((t * 8 + q) * 64 + lane) << 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], xb[q][j], acc[t], 0, 0, 0);
      }
    // ---- layer 1 epilogue: bias + ReLU in registers; y[4t + g][j] = channel 32t + 8g + 4h + j
    f32x4 y[8];
    unsigned int bits[2] = {0u, 0u};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(B1s + 32 * t + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = fmaxf(acc[t][4 * g + j] + b[j], 0.f);
          y[4 * t + g][j] = v;
          bits[t] |= (v > 0.f ? 1u : 0u) << (8 * g + 4 * h + j);
        }
      }
    if (p.y1 != nullptr) store_rows(y, p.y1, p.ld_y1, tile);
    if (p.gate1 != nullptr) {
      const unsigned int w0 = bits[0] | (unsigned int)__shfl_xor((int)bits[0], 32, 64);
      const unsigned int w1 = bits[1] | (unsigned int)__shfl_xor((int)bits[1], 32, 64);
      if (ok) p.gate1[(size_t)row * p.ld_gate1 + h] = h ? w1 : w0;
    }
    // ---- layer 2: its B operand is y as it stands
    f32x16 acc2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(W2s + (((t * 8 + q) * 64 + lane) << 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], y[q][j], acc2[t], 0, 0, 0);
      }
    unsigned int bits2[NT2];
    f32x4 y2[4 * NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
      bits2[t] = 0u;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 32 * t + 8 * g + 4 * h;
        const f32x4 b = *reinterpret_cast<const f32x4*>(B2s + c0);
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float u = acc2[t][4 * g + j] + b[j];
          if (p.relu2) u = fmaxf(u, 0.f);
          v[j] = u;
          bits2[t] |= (u > 0.f ? 1u : 0u) << (8 * g + 4 * h + j);
        }
        y2[4 * t + g] = v;
        // the head's narrow last layer (a few channels per pixel) stores straight from the registers
        if (NT2 == 1 && ok && c0 < n2p) *reinterpret_cast<f32x4*>(p.y2 + (size_t)row * p.ld_y2 + c0) = v;
      }
    }
    if constexpr (NT2 == 2) store_rows(y2, p.y2, p.ld_y2, tile);
    if (p.gate2 != nullptr) {
      if constexpr (NT2 == 2) {
        const unsigned int w0 = bits2[0] | (unsigned int)__shfl_xor((int)bits2[0], 32, 64);
        const unsigned int w1 = bits2[1] | (unsigned int)__shfl_xor((int)bits2[1], 32, 64);
        if (ok) p.gate2[(size_t)row * p.ld_gate2 + h] = h ? w1 : w0;
      } else {
        const unsigned int w0 = bits2[0] | (unsigned int)__shfl_xor((int)bits2[0], 32, 64);
        if (ok && h == 0) p.gate2[(size_t)row * p.ld_gate2] = w0;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) xl[q] = xn[q];
  }
}

struct ChainBwdP {
  const float* dp2; int ld_dp2; int N2;
  const float* y1; int ld_y1;
  const float* x; int ld_x; int gate_x;
  int M;
  const float* w2t; const float* w1t;
  float* dp0; int ld_dp0;
  float* dw2; float* db2; float* dw1; float* db1;
  int ntiles;          // tiles of 128 pixels
};

// KQ2: k-steps of 8 over layer 2's output channels — 8 (64 channels) or 1 (N2 <= 8, the head's last layer)
// One block per CU: three staged tiles (dP2 then dP1 | y1 then dP0 | x) + both weight matrices are 134 KB of LDS.
// Global memory is touched in full lines only (see the forward kernel): the rows a wave loads go to LDS as they
// are, the fragments of the data-gradient products are read back from there — which the weight gradients need in
// LDS anyway — and dP0 leaves through the y1 tile once the layer-2 weight gradient has read it.  The next tile's
// rows are loaded right after the current tile's registers have been written to LDS: one register set, in flight
// for the whole tile.
template <int KQ2>
__global__ __launch_bounds__(256, 1) void chain64_bwd_kernel(const ChainBwdP p) {
  __shared__ __attribute__((aligned(16))) float W2Ts[2 * 2048];
  __shared__ __attribute__((aligned(16))) float W1Ts[2 * 2048];
  __shared__ __attribute__((aligned(16))) float bufA[128 * LDW];      // dP2 rows, later dP1 rows
  __shared__ __attribute__((aligned(16))) float bufB[128 * LDW];      // y1 rows, later dP0 rows on their way out
  __shared__ __attribute__((aligned(16))) float bufC[128 * LDW];      // x rows
  const int n2p = (p.N2 + 3) & ~3;
  load_matrix<2>(W2Ts, p.w2t, 64, n2p, n2p);
  load_matrix<2>(W1Ts, p.w1t, 64, 64, 64);
  __syncthreads();

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int lr = lane >> 4, lp = lane & 15;       // full-line role: row 4j + lr of the wave's 32, piece lp
  constexpr int NT2W = KQ2 == 8 ? 2 : 1;          // 32-row tiles of dW2 that hold rows at all
  const int bch = tid & 63;                       // bias sums: channel bch over this wave's 32 staged rows
  // every wave accumulates the WHOLE of dW2 and dW1 over its own 32 pixels of a tile (128 accumulator registers:
  // one wave per SIMD has them): no hand-off between waves, hence no block barrier anywhere in the tile loop
  f32x16 accW2[NT2W][2], accW1[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        accW1[a][b][r] = 0.f;
        if (a < NT2W) accW2[a][b][r] = 0.f;
      }
  float bsum2 = 0.f, bsum1 = 0.f;
  const int srow = wid * 32 + i;                  // this lane's pixel among the block's 128
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // rows of tile t in the full-line layout; rows beyond the tensor are loaded clamped and zeroed
  // (dP2 of the head's narrow last layer is 16 or 32 bytes per pixel: a lane takes its pixel's half)
  f32x4 l2[KQ2 == 8 ? 8 : 1], ly[8], lx[8];
  auto load_rows = [&](int tile) {
    if (KQ2 == 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = tile * 128 + wid * 32 + 4 * j + lr;
        const bool in = row < p.M;
        l2[j] = *reinterpret_cast<const f32x4*>(p.dp2 + (size_t)(in ? row : p.M - 1) * p.ld_dp2 + 4 * lp);
        if (!in) l2[j] = zero4;
      }
    } else {
      const int row = tile * 128 + srow;
      const bool in = row < p.M;
      const bool have = 4 * h < n2p;                 // N2 <= 4: lanes h = 1 hold zeros; 5..8: channels 4-7
      l2[0] = *reinterpret_cast<const f32x4*>(p.dp2 + (size_t)(in ? row : p.M - 1) * p.ld_dp2 + (have ? 4 * h : 0));
      if (!in || !have) l2[0] = zero4;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = tile * 128 + wid * 32 + 4 * j + lr;
      const bool in = row < p.M;
      const size_t rc = (size_t)(in ? row : p.M - 1);
      ly[j] = *reinterpret_cast<const f32x4*>(p.y1 + rc * p.ld_y1 + 4 * lp);
      lx[j] = *reinterpret_cast<const f32x4*>(p.x + rc * p.ld_x + 4 * lp);
      if (!in) { ly[j] = zero4; lx[j] = zero4; }
    }
  };
  float* rowsA = bufA + wid * 32 * LDW;
  float* rowsB = bufB + wid * 32 * LDW;
  float* rowsC = bufC + wid * 32 * LDW;

  int tile = blockIdx.x;
  if (tile < p.ntiles) load_rows(tile);
  for (; tile < p.ntiles; tile += gridDim.x) {
    // ---- the tile's rows into LDS (the previous tile's last readers passed the loop-end barrier)
    if (KQ2 == 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(rowsA + (4 * j + lr) * LDW + 4 * lp) = l2[j];
    } else {
      // 8 columns: the pixel's 4 channels (lanes h = 0) and 4 zeros (lanes h = 1)
      *reinterpret_cast<f32x4*>(rowsA + i * LDW + 4 * h) = l2[0];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      *reinterpret_cast<f32x4*>(rowsB + (4 * j + lr) * LDW + 4 * lp) = ly[j];
      *reinterpret_cast<f32x4*>(rowsC + (4 * j + lr) * LDW + 4 * lp) = lx[j];
    }
    __builtin_amdgcn_wave_barrier();
    load_rows(tile + (int)gridDim.x < p.ntiles ? tile + (int)gridDim.x : tile);     // in flight during this tile
    // ---- this wave's fragments (its own rows: the wave's LDS operations execute in order)
    f32x4 g2[KQ2], yv[8];
#pragma unroll
    for (int q = 0; q < KQ2; ++q) g2[q] = *reinterpret_cast<const f32x4*>(rowsA + i * LDW + 8 * q + 4 * h);
#pragma unroll
    for (int q = 0; q < 8; ++q) yv[q] = *reinterpret_cast<const f32x4*>(rowsB + i * LDW + 8 * q + 4 * h);
    // ---- data gradient through layer 2: dP1^T = W2^T dP2^T, gated by y1 > 0
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int q = 0; q < KQ2; ++q)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(W2Ts + (((t * 8 + q) * 64 + lane) << 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], g2[q][j], acc[t], 0, 0, 0);
      }
    f32x4 g1[8];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) g1[4 * t + g][j] = yv[4 * t + g][j] > 0.f ? acc[t][4 * g + j] : 0.f;
    // ---- weight gradient of layer 2 over this wave's pixels: dW2[n][c] += sum_p dP2[p][n] y1[p][c]
    {
      const float* ap = rowsA + h * LDW + i;
      const float* bp = rowsB + h * LDW + i;
#pragma unroll 4
      for (int s2 = 0; s2 < 16; ++s2) {
        const float b0 = bp[2 * s2 * LDW], b1 = bp[2 * s2 * LDW + 32];
#pragma unroll
        for (int a = 0; a < NT2W; ++a) {
          const float av = ap[2 * s2 * LDW + 32 * a];
          accW2[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, accW2[a][0], 0, 0, 0);
          accW2[a][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, accW2[a][1], 0, 0, 0);
        }
      }
    }
    if (p.db2 != nullptr && bch < n2p) {
      const float* cp = rowsA + bch;
      float t2 = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) t2 += cp[r * LDW];
      bsum2 += t2;
    }
    // ---- data gradient through layer 1: dP0^T = W1^T dP1^T, gated by x > 0 (x is the previous layer's ReLU output)
    f32x4 d0[8];
    if (p.dp0 != nullptr) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(W1Ts + (((t * 8 + q) * 64 + lane) << 2));
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], g1[q][j], acc[t], 0, 0, 0);
        }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(rowsC + i * LDW + 32 * t + 8 * g + 4 * h);
#pragma unroll
          for (int j = 0; j < 4; ++j) d0[4 * t + g][j] = (!p.gate_x || xv[j] > 0.f) ? acc[t][4 * g + j] : 0.f;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- dP1 rows for the layer-1 weight gradient; dP0 out through the y1 tile, in full lines
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x4*>(rowsA + i * LDW + 8 * q + 4 * h) = g1[q];
    if (p.dp0 != nullptr) {
#pragma unroll
      for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x4*>(rowsB + i * LDW + 8 * q + 4 * h) = d0[q];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(rowsB + (4 * j + lr) * LDW + 4 * lp);
        const int row = tile * 128 + wid * 32 + 4 * j + lr;
        if (row < p.M) *reinterpret_cast<f32x4*>(p.dp0 + (size_t)row * p.ld_dp0 + 4 * lp) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    {
      const float* ap = rowsA + h * LDW + i;
      const float* bp = rowsC + h * LDW + i;
#pragma unroll 4
      for (int s2 = 0; s2 < 16; ++s2) {
        const float b0 = bp[2 * s2 * LDW], b1 = bp[2 * s2 * LDW + 32];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const float av = ap[2 * s2 * LDW + 32 * a];
          accW1[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, accW1[a][0], 0, 0, 0);
          accW1[a][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, accW1[a][1], 0, 0, 0);
        }
      }
    }
    if (p.db1 != nullptr) {
      const float* cp = rowsA + bch;
      float t1 = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) t1 += cp[r * LDW];
      bsum1 += t1;
    }
    __builtin_amdgcn_wave_barrier();               // (the wave's own reads precede its next tile's writes in program order)
  }

  // ---- flush: the four waves' sums meet in LDS (the x tile is free now), then one atomic per element and block
  __syncthreads();
  float* red = bufC;                               // [dW2 64 x 64 | dW1 64 x 64]
  for (int k = tid; k < 2 * 4096; k += 256) red[k] = 0.f;
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int c = 32 * b + i;
        if (a < NT2W) atomicAdd(&red[n * 64 + c], accW2[a < NT2W ? a : 0][b][r]);
        atomicAdd(&red[4096 + n * 64 + c], accW1[a][b][r]);
      }
  __syncthreads();
  for (int k = tid; k < 4096; k += 256) {
    if ((k >> 6) < n2p) atomicAdd(p.dw2 + k, red[k]);
    atomicAdd(p.dw1 + k, red[4096 + k]);
  }
  if (p.db2 != nullptr && bch < p.N2) atomicAdd(p.db2 + bch, bsum2);
  if (p.db1 != nullptr) atomicAdd(p.db1 + bch, bsum1);
}

int chain_grid(const void* fn, int work_items) {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  int g = cus * per_cu;
  if (g > work_items) g = work_items;
  return g < 1 ? 1 : g;
}

}  // namespace

extern "C" int clx_chain64_fwd(const float* x, int ld_x, long long M, const float* w1, const float* b1, float* y1,
                               int ld_y1, unsigned int* gate1, int ld_gate1, const float* w2, const float* b2,
                               int N2, int relu2, float* y2, int ld_y2, unsigned int* gate2, int ld_gate2,
                               clx_stream stream) {
  CLX_REQUIRE(x && w1 && w2 && y2, "clx_chain64_fwd: null pointer");
  CLX_REQUIRE(M > 0 && M < (1ll << 31) - 256, "clx_chain64_fwd: bad pixel count %lld", M);
  CLX_REQUIRE(N2 >= 1 && N2 <= 64 && (N2 == 64 || N2 <= 32), "clx_chain64_fwd: N2 must be 64 or <= 32 (got %d)", N2);
  const int n2p = (N2 + 3) & ~3;
  CLX_REQUIRE(ld_x >= 64 && ld_x % 4 == 0 && ld_y2 >= n2p && ld_y2 % 4 == 0 && (y1 == nullptr || (ld_y1 >= 64 && ld_y1 % 4 == 0)),
              "clx_chain64_fwd: leading dimensions must be multiples of 4 and cover the channels");
  CLX_REQUIRE((((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)y1 | (uintptr_t)y2) & 15) == 0,
              "clx_chain64_fwd: pointers must be 16-byte aligned");
  CLX_REQUIRE(gate1 == nullptr || ld_gate1 >= 2, "clx_chain64_fwd: gate1 needs two words per pixel");
  CLX_REQUIRE(gate2 == nullptr || (relu2 && ld_gate2 >= (n2p + 31) / 32), "clx_chain64_fwd: gate2 needs relu2 and whole words");
  ChainFwdP p{x, ld_x, (int)M, w1, b1, y1, ld_y1, gate1, ld_gate1, w2, b2, N2, relu2, y2, ld_y2, gate2, ld_gate2,
              (int)((M + 31) / 32)};
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (clx_prof_enabled()) clx_prof_events(CLX_PROF_CHAIN64, 2.0 * M * 64 * (64 + N2), &e0, &e1);
  if (N2 == 64) {
    const int g = chain_grid((const void*)chain64_fwd_kernel<2>, (p.ntiles + 3) / 4);
    CLX_LAUNCH_TIMED((chain64_fwd_kernel<2>), dim3(g), dim3(256), st, e0, e1, p);
  } else {
    const int g = chain_grid((const void*)chain64_fwd_kernel<1>, (p.ntiles + 3) / 4);
    CLX_LAUNCH_TIMED((chain64_fwd_kernel<1>), dim3(g), dim3(256), st, e0, e1, p);
  }
  CLX_CHECK_LAUNCH("clx_chain64_fwd");
  return CLX_OK;
}

extern "C" int clx_chain64_bwd(const float* dp2, int ld_dp2, int N2, const float* y1, int ld_y1, const float* x,
                               int ld_x, int gate_x, long long M, const float* w2t, const float* w1t, float* dp0,
                               int ld_dp0, float* dw2, float* db2, float* dw1, float* db1, clx_stream stream) {
  CLX_REQUIRE(dp2 && y1 && x && w2t && w1t && dw2 && dw1, "clx_chain64_bwd: null pointer");
  CLX_REQUIRE(M > 0 && M < (1ll << 31) - 256, "clx_chain64_bwd: bad pixel count %lld", M);
  CLX_REQUIRE(N2 == 64 || (N2 >= 1 && N2 <= 8), "clx_chain64_bwd: N2 must be 64 or <= 8 (got %d)", N2);
  const int n2p = (N2 + 3) & ~3;
  CLX_REQUIRE(ld_dp2 >= n2p && ld_dp2 % 4 == 0 && ld_y1 >= 64 && ld_y1 % 4 == 0 && ld_x >= 64 && ld_x % 4 == 0 &&
                  (dp0 == nullptr || (ld_dp0 >= 64 && ld_dp0 % 4 == 0)),
              "clx_chain64_bwd: leading dimensions must be multiples of 4 and cover the channels");
  CLX_REQUIRE((((uintptr_t)dp2 | (uintptr_t)y1 | (uintptr_t)x | (uintptr_t)w2t | (uintptr_t)w1t | (uintptr_t)dp0) & 15) == 0,
              "clx_chain64_bwd: pointers must be 16-byte aligned");
  ChainBwdP p{dp2, ld_dp2, N2, y1, ld_y1, x, ld_x, gate_x, (int)M, w2t, w1t, dp0, ld_dp0, dw2, db2, dw1, db1,
              (int)((M + 127) / 128)};
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // two data-gradient and two weight-gradient products
  if (clx_prof_enabled()) clx_prof_events(CLX_PROF_CHAIN64, 2.0 * M * 64 * (64 + N2) * (dp0 ? 2.0 : 1.5), &e0, &e1);
  if (N2 == 64) {
    const int g = chain_grid((const void*)chain64_bwd_kernel<8>, p.ntiles);
    CLX_LAUNCH_TIMED((chain64_bwd_kernel<8>), dim3(g), dim3(256), st, e0, e1, p);
  } else {
    const int g = chain_grid((const void*)chain64_bwd_kernel<1>, p.ntiles);
    CLX_LAUNCH_TIMED((chain64_bwd_kernel<1>), dim3(g), dim3(256), st, e0, e1, p);
  }
  CLX_CHECK_LAUNCH("clx_chain64_bwd");
  return CLX_OK;
}
