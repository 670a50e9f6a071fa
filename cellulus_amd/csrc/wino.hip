// Winograd F(m x m, 3x3), m = 2 or 4, for 2-D 3x3 layers (forward, data gradient, weight
// gradient) on top of the f32-MFMA GEMM kernels; a = m + 2:
//
//   forward / dgrad:  V = B^T d B  (a x a input patches, stride m)    [HBM streaming]
//                     M[xi] = V[xi] . U[xi]^T, xi = 0..a^2-1          [a^2 batched MFMA GEMMs]
//                     Y = A^T M A (+ bias, ReLU, ReLU gate)           [HBM streaming]
//   wgrad:            dU[xi] = (A dY A^T)[xi]^T . V[xi]  over tiles   [a^2 batched MFMA GEMMs]
//                     dW = G^T dU G   (clx_unpack_wgrad_wino)
//
// a^2 multiplications per m^2 outputs instead of 9 m^2: 2.25x (m = 2) / 4x (m = 4) fewer MFMA
// FLOPs, all in f32.  The price is HBM traffic for V and M (a^2/m^2 = 4x / 2.25x the activation
// each) and rounding error (see WT below), so the plan selects it only for layers with enough
// channels on both sides.
//
// Replaces the same reference calls as conv_igemm.hip / conv_wgrad.hip (nn.Conv2d 3x3 and
// its autograd backward, cellulus/models/unet.py:24-51, cellulus/train.py:178).
#include "clx_common.h"
#include "sp_planes.h"
#include "wino_tables.h"

namespace {

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

struct Geom {
  int B, SH, SW, oy, ox;   // images (batch x z planes), stored grid of the source and crop offset
  int ID, SD, oz;          // z planes per batch element (logical / stored) and z crop: image bi is
                           // plane bi % ID of batch element bi / ID (2-D: ID = SD = 1, oz = 0)
  int IH, IW, P;           // logical input extent, zero padding
  int OH, OW, th, tw;      // output extent, tiles per image
  long long T;             // B * th * tw
};

// V[xi][t][c] = (B^T d B)[xi] for the A x A input patch of tile t (stride MT)
// PL: V as the P3 planes of the split-precision products (sp_planes.h; one plane set per xi, `plane_bytes` apart) — the
// work items are dealt out so that a wavefront reads whole lines and writes 128-byte runs of every piece
template <int MT, int R, bool PL = false>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int ld_x, int C4, Geom g,
                                                         float* __restrict__ V, long long total,
                                                         const int* __restrict__ tile_list = nullptr, long long Tc = 0,
                                                         long long plane_bytes = 0) {
  using W = WT<MT, R>;
  constexpr int A = W::A;
  const int C = C4 * 4;
  // (tile_list: only the listed tiles, stored compactly — V[xi][i][c] for the i-th listed tile)
  const long long Trows = tile_list ? Tc : g.T;
  const long long plane = Trows * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int c;
    long long t;
    if constexpr (PL) {
      sp::item_to_row_channel(i, C, t, c);
      if (t >= Trows) continue;
    } else {
      c = (int)(i % C4) * 4;
      t = i / C4;
    }
    const long long tg = tile_list ? (long long)tile_list[t] : t;
    const int tx = (int)(tg % g.tw);
    const long long q = tg / g.tw;
    const int ty = (int)(q % g.th);
    const int b = (int)(q / g.th);
    const int bb = b / g.ID, bz = b - bb * g.ID;
    f32x4 d[A][A];
#pragma unroll
    for (int r = 0; r < A; ++r) {
      const int ly = MT * ty - g.P + r;
#pragma unroll
      for (int s = 0; s < A; ++s) {
        const int lx = MT * tx - g.P + s;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ly < (unsigned)g.IH && (unsigned)lx < (unsigned)g.IW) {
          const long long pix = (((long long)bb * g.SD + bz + g.oz) * g.SH + ly + g.oy) * g.SW + lx + g.ox;
          v = ld4(x + pix * ld_x + c);
        }
        d[r][s] = v;
      }
    }
    // columns: w[:, s] = B^T d[:, s]
    f32x4 w[A][A];
#pragma unroll
    for (int s = 0; s < A; ++s)
#pragma unroll
      for (int r = 0; r < A; ++r) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy(acc, first, W::BT[r][k], d[k][s]);
        w[r][s] = acc;
      }
    // rows: V[r][q] = sum_k w[r][k] * B^T[q][k]
    float* dst = V + t * C + c;
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int qq = 0; qq < A; ++qq) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy(acc, first, W::BT[qq][k], w[r][k]);
        if constexpr (PL) sp::store4(reinterpret_cast<char*>(V) + (r * A + qq) * plane_bytes, t, c, C >> 4, acc);
        else st4(dst + (r * A + qq) * plane, acc);
      }
  }
}

// Y = A^T m A (MT x MT outputs per tile); bias, ReLU, ReLU gate fused
template <int MT, int R>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, int N4, Geom g,
                                                          const float* __restrict__ bias, int relu,
                                                          const float* __restrict__ mask, int ld_mask,
                                                          float* __restrict__ out, int ld_out, int Nreal,
                                                          int accumulate, unsigned int* __restrict__ gate_out,
                                                          int ld_gate, const unsigned int* __restrict__ mask_bits,
                                                          int ld_mask_bits, float* __restrict__ pool_out, int ld_pool,
                                                          long long total, const int* __restrict__ tile_list = nullptr,
                                                          long long Tc = 0) {
  using W = WT<MT, R>;
  constexpr int A = W::A;
  constexpr int MP = MT / 2;          // 2 x 2 pooling windows per tile side (MT = 2 or 4: windows never straddle tiles)
  const int N = N4 * 4;
  const long long plane = (tile_list ? Tc : g.T) * N;
  // gate bits (N4 % 8 == 0, checked by the launcher): the 8 lanes i .. i + 7, i % 8 == 0, are the 32
  // channels of one word of one tile, and they take every branch below together
  const int sub = threadIdx.x & 7;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N4) * 4;
    const long long t = i / N4;
    const long long tg = tile_list ? (long long)tile_list[t] : t;
    const int tx = (int)(tg % g.tw);
    const long long q = tg / g.tw;
    const int ty = (int)(q % g.th);
    const int b = (int)(q / g.th);
    const float* src = M + t * N + n;
    // rows first, one column of m at a time: r[a][s] = sum_k A^T[a][k] m[k][s]
    f32x4 rr[MT][A];
#pragma unroll
    for (int s = 0; s < A; ++s) {
      f32x4 m[A];
#pragma unroll
      for (int k = 0; k < A; ++k) m[k] = ld4(src + (k * A + s) * plane);
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy(acc, first, W::AT[a][k], m[k]);
        rr[a][s] = acc;
      }
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = (n + e < Nreal) ? bias[n + e] : 0.f;
    }
    f32x4 pm[MP][MP];                  // running maxima of the tile's pooling windows (pool_out only)
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int oy = MT * ty + a;
      if (oy >= g.OH) continue;
#pragma unroll
      for (int cc = 0; cc < MT; ++cc) {
        const int ox = MT * tx + cc;
        if (ox >= g.OW) continue;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy(v, first, W::AT[cc][k], rr[a][k]);
        v += bv;
        const long long m = ((long long)b * g.OH + oy) * g.OW + ox;
        if (accumulate) {       // out = act(conv + bias + out); ld_out % 4 == 0, so the 16 bytes exist
          v += ld4(out + m * ld_out + n);
        }
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (mask_bits) {
          const unsigned int wbits = mask_bits[m * ld_mask_bits + (n >> 5)] >> (n & 31);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ((wbits >> e) & 1u) ? v[e] : 0.f;
        }
        if (gate_out) {
          unsigned int nib = 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) nib |= (n + e < Nreal && v[e] > 0.f) ? (1u << e) : 0u;
          unsigned int word = nib << (4 * sub);
          word |= __shfl_xor(word, 1, 64);
          word |= __shfl_xor(word, 2, 64);
          word |= __shfl_xor(word, 4, 64);
          if (sub == 0) gate_out[m * ld_gate + (n >> 5)] = word;
        }
        if (n + 3 < Nreal) {
          if (mask) {
            const f32x4 mk = ld4(mask + m * ld_mask + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (mk[e] > 0.f) ? v[e] : 0.f;
          }
          st4(out + m * ld_out + n, v);
        } else {
          for (int e = 0; e < 4 && n + e < Nreal; ++e) {
            float xv = v[e];
            if (mask) xv = (mask[m * ld_mask + n + e] > 0.f) ? xv : 0.f;
            out[m * ld_out + n + e] = xv;
          }
        }
        if (pool_out) {                // (the launcher admits pool_out only with whole channel groups and no mask)
          if ((a & 1) == 0 && (cc & 1) == 0) pm[a >> 1][cc >> 1] = v;
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) pm[a >> 1][cc >> 1][e] = fmaxf(pm[a >> 1][cc >> 1][e], v[e]);
          }
          if ((a & 1) && (cc & 1)) {   // the window's last pixel (OH, OW even: windows are whole)
            const long long pm_ = ((long long)b * (g.OH >> 1) + (oy >> 1)) * (g.OW >> 1) + (ox >> 1);
            st4(pool_out + pm_ * ld_pool + n, pm[a >> 1][cc >> 1]);
          }
        }
      }
    }
  }
}

// ADJOINT data gradient of F(4x4, 3x3): the forward is Y = A^T [U (B^T d B)] A per tile, so
//   d(input patch) = B [U^T (A dY A^T)] B^T        (A x A = 6 x 6, one per tile)
// and dX is the overlap-add of the patches (stride MT = 4, overlap 2).  P[xi][t][c] = (U^T Mdy)[xi] comes from the
// batched GEMM; this kernel is output-stationary, so nothing is added in memory: a thread owns one column of 4 x 4
// blocks of dX (ADJ_SEG of them, 4 channels) and walks DOWN the tiles — per tile row it transforms tile (ty, bx)
// (patch columns 0-3) and its left neighbour (patch columns 4-5 = block columns 0-1), stores the block with the two
// rows carried over from the tile above, and carries its own rows 4-5 on.  2 (+ 1/ADJ_SEG) tile reads per block
// where a block-by-block gather reads 4 (344 -> ... us on the benchmark's layers).  Epilogue: the ReLU gate of the
// tensor whose gradient this is (float mask or bits).
constexpr int ADJ_SEG = 8;

__global__ __launch_bounds__(256) void wino_adjoint_output_kernel(const float* __restrict__ Pm, int C4, int Bn, int th,
                                                                  int tw, int IH, int IW, const float* __restrict__ mask,
                                                                  int ld_mask, const unsigned int* __restrict__ mask_bits,
                                                                  int ld_mask_bits, float* __restrict__ out, int ld_out,
                                                                  int Creal, long long total) {
  using W = WT<4, 3>;
  constexpr int A = 6;
  const int C = C4 * 4;
  const int bh = (IH + 3) / 4, bw = (IW + 3) / 4;
  const int nseg = (bh + ADJ_SEG - 1) / ADJ_SEG;
  const long long plane = (long long)Bn * th * tw * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long t = i / C4;
    const int bx = (int)(t % bw);
    long long q = t / bw;
    const int sg = (int)(q % nseg);
    const int b = (int)(q / nseg);
    const int by0 = sg * ADJ_SEG, by1 = min(by0 + ADJ_SEG, bh);
    f32x4 carry[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) carry[a][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    // tiles ty = by0 - 1 (only for what it carries into block by0) .. by1 - 1; block by = th (if it exists) has no
    // tile of its own and consists of the carry alone
    for (int ty = by0 - 1; ty < by1; ++ty) {
      f32x4 d[A][4];
#pragma unroll
      for (int a = 0; a < A; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) d[a][e] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ty >= 0 && ty < th) {
#pragma unroll
        for (int ux = 0; ux < 2; ++ux) {
          const int tx = bx - ux;
          if (tx < 0 || tx >= tw) continue;
          const float* src = Pm + (((long long)b * th + ty) * tw + tx) * C + c;
#pragma unroll
          for (int s2 = 0; s2 < A; ++s2) {
            f32x4 pcol[A], wcol[A];
#pragma unroll
            for (int r = 0; r < A; ++r) pcol[r] = ld4(src + (r * A + s2) * plane);
            // wcol[a] = sum_r B[a][r] P[r][s2] = sum_r BT[r][a] P[r][s2]
#pragma unroll
            for (int a = 0; a < A; ++a) {
              f32x4 v = {0.f, 0.f, 0.f, 0.f};
              bool first = true;
#pragma unroll
              for (int r = 0; r < A; ++r) axpy(v, first, W::BT[r][a], pcol[r]);
              wcol[a] = v;
            }
            // d[a][e] += wcol[a] * B[col][s2], col = e (own tile) or 4 + e, e < 2 (left neighbour)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (ux && e >= 2) continue;
              const float coef = ux ? W::BT[s2][4 + (e & 1)] : W::BT[s2][e];
#pragma unroll
              for (int a = 0; a < A; ++a) {
                bool first = false;
                axpy(d[a][e], first, coef, wcol[a]);
              }
            }
          }
        }
      }
      if (ty >= by0) {
        // block by = ty: rows 0-3 of this tile row, plus rows 4-5 of the one above on its rows 0-1
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int y = 4 * ty + a;
          if (y >= IH) continue;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int x = 4 * bx + e;
            if (x >= IW) continue;
            f32x4 v = d[a][e];
            if (a < 2) v += carry[a][e];
            const long long m = ((long long)b * IH + y) * IW + x;
            if (mask_bits) {
              const unsigned int wbits = mask_bits[m * ld_mask_bits + (c >> 5)] >> (c & 31);
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = ((wbits >> k) & 1u) ? v[k] : 0.f;
            }
            if (c + 3 < Creal) {
              if (mask) {
                const f32x4 mk = ld4(mask + m * ld_mask + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (mk[k] > 0.f) ? v[k] : 0.f;
              }
              st4(out + m * ld_out + c, v);
            } else {
              for (int k = 0; k < 4 && c + k < Creal; ++k) {
                float xv = v[k];
                if (mask) xv = (mask[m * ld_mask + c + k] > 0.f) ? xv : 0.f;
                out[m * ld_out + c + k] = xv;
              }
            }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { carry[0][e] = d[4][e]; carry[1][e] = d[5][e]; }
    }
  }
}

// Mdy[xi][t][n] = (A dy A^T)[xi] with A = (A^T)^T (A x MT); dbias[n] += sum of dy
// (block-private LDS accumulator, then one global atomic per channel per block)
template <int MT, int R, bool PL = false>
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, int ld_dy, int N4,
                                                      Geom g, float* __restrict__ Md,
                                                      float* __restrict__ dbias, int Nreal,
                                                      long long total, long long plane_bytes = 0) {
  using W = WT<MT, R>;
  constexpr int A = W::A;
  extern __shared__ float bacc[];
  const int N = N4 * 4;
  const long long plane = g.T * N;
  if (dbias) {
    for (int k = threadIdx.x; k < N; k += blockDim.x) bacc[k] = 0.f;
    __syncthreads();
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int n;
    long long t;
    if constexpr (PL) {
      sp::item_to_row_channel(i, N, t, n);
      if (t >= g.T) continue;
    } else {
      n = (int)(i % N4) * 4;
      t = i / N4;
    }
    const int tx = (int)(t % g.tw);
    const long long q = t / g.tw;
    const int ty = (int)(q % g.th);
    const int b = (int)(q / g.th);
    f32x4 d[MT][MT];
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int c = 0; c < MT; ++c) {
        const int oy = MT * ty + a, ox = MT * tx + c;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (oy < g.OH && ox < g.OW) v = ld4(dy + (((long long)b * g.OH + oy) * g.OW + ox) * ld_dy + n);
        d[a][c] = v;
        sum += v;
      }
    if (dbias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&bacc[n + e], sum[e]);
    }
    // w[r][c] = sum_a A^T[a][r] d[a][c]
    f32x4 w[A][MT];
#pragma unroll
    for (int c = 0; c < MT; ++c)
#pragma unroll
      for (int r = 0; r < A; ++r) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
#pragma unroll
        for (int a = 0; a < MT; ++a) axpy(acc, first, W::AT[a][r], d[a][c]);
        w[r][c] = acc;
      }
    float* dst = Md + t * N + n;
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int qq = 0; qq < A; ++qq) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
#pragma unroll
        for (int c = 0; c < MT; ++c) axpy(acc, first, W::AT[c][qq], w[r][c]);
        if constexpr (PL) sp::store4(reinterpret_cast<char*>(Md) + (r * A + qq) * plane_bytes, t, n, N >> 4, acc);
        else st4(dst + (r * A + qq) * plane, acc);
      }
  }
  if (dbias) {
    __syncthreads();
    for (int k = threadIdx.x; k < Nreal; k += blockDim.x)
      if (bacc[k] != 0.f) atomicAdd(dbias + k, bacc[k]);
  }
}

// dY read ONCE for both of its Winograd transforms: the data gradient's input transform
// V_d = B^T d B of the (K-1)-padded dY (tile t of the dX grid reads the A x A patch at MT t - P) and
// the weight gradient's Mdy = A dy A^T of the MT x MT tile at MT t — which is rows / columns
// P .. P + MT - 1 of that same patch.  gd: geometry of the data gradient's transform (over the dX
// tiles), gw: the weight gradient's (over the dY tiles); dbias as in wino_dy_kernel.
template <int MT, int R, bool PL = false>
__global__ __launch_bounds__(256) void wino_dy_dual_kernel(const float* __restrict__ dy, int ld_dy, int N4,
                                                           Geom gd, Geom gw, float* __restrict__ Vd,
                                                           float* __restrict__ Md, float* __restrict__ dbias,
                                                           int Nreal, long long total, long long pb_d = 0, long long pb_w = 0) {
  using W = WT<MT, R>;
  constexpr int A = W::A;
  constexpr int P = R - 1;
  extern __shared__ float bacc[];
  const int N = N4 * 4;
  const long long plane_d = gd.T * N, plane_w = gw.T * N;
  if (dbias) {
    for (int k = threadIdx.x; k < N; k += blockDim.x) bacc[k] = 0.f;
    __syncthreads();
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int n;
    long long t;
    if constexpr (PL) {
      sp::item_to_row_channel(i, N, t, n);
      if (t >= gd.T) continue;
    } else {
      n = (int)(i % N4) * 4;
      t = i / N4;
    }
    const int tx = (int)(t % gd.tw);
    const long long q = t / gd.tw;
    const int ty = (int)(q % gd.th);
    const int b = (int)(q / gd.th);
    // ---- weight-gradient side first, from its own loads of the inner MT x MT block: holding the whole
    // A x A patch across both transforms costs 256 VGPRs (one wave per SIMD — measured 0.6 ms SLOWER per
    // 3-D step); read twice, the second time the block comes out of the vector cache
    if (ty < gw.th && tx < gw.tw) {
      f32x4 dd[MT][MT];
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < MT; ++c) {
          const int oy = MT * ty + a, ox = MT * tx + c;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (oy < gd.IH && ox < gd.IW) v = ld4(dy + (((long long)b * gd.IH + oy) * gd.IW + ox) * ld_dy + n);
          dd[a][c] = v;
          sum += v;
        }
      if (dbias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&bacc[n + e], sum[e]);
      }
      f32x4 w[A][MT];
#pragma unroll
      for (int c = 0; c < MT; ++c)
#pragma unroll
        for (int r = 0; r < A; ++r) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          bool first = true;
#pragma unroll
          for (int a = 0; a < MT; ++a) axpy(acc, first, W::AT[a][r], dd[a][c]);
          w[r][c] = acc;
        }
      const long long tw_lin = ((long long)b * gw.th + ty) * gw.tw + tx;
      float* dst = Md + tw_lin * N + n;
#pragma unroll
      for (int r = 0; r < A; ++r)
#pragma unroll
        for (int qq = 0; qq < A; ++qq) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          bool first = true;
#pragma unroll
          for (int c = 0; c < MT; ++c) axpy(acc, first, W::AT[c][qq], w[r][c]);
          if constexpr (PL) sp::store4(reinterpret_cast<char*>(Md) + (r * A + qq) * pb_w, tw_lin, n, N >> 4, acc);
          else st4(dst + (r * A + qq) * plane_w, acc);
        }
    }
    f32x4 d[A][A];
#pragma unroll
    for (int r = 0; r < A; ++r) {
      const int ly = MT * ty - P + r;
#pragma unroll
      for (int c = 0; c < A; ++c) {
        const int lx = MT * tx - P + c;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ly < (unsigned)gd.IH && (unsigned)lx < (unsigned)gd.IW)
          v = ld4(dy + (((long long)b * gd.IH + ly) * gd.IW + lx) * ld_dy + n);
        d[r][c] = v;
      }
    }
    // ---- data-gradient side: V_d = B^T d B
    f32x4 w2[A][A];
#pragma unroll
    for (int c = 0; c < A; ++c)
#pragma unroll
      for (int r = 0; r < A; ++r) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy(acc, first, W::BT[r][k], d[k][c]);
        w2[r][c] = acc;
      }
    float* dstv = Vd + t * N + n;
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int qq = 0; qq < A; ++qq) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy(acc, first, W::BT[qq][k], w2[r][k]);
        if constexpr (PL) sp::store4(reinterpret_cast<char*>(Vd) + (r * A + qq) * pb_d, t, n, N >> 4, acc);
        else st4(dstv + (r * A + qq) * plane_d, acc);
      }
  }
  if (dbias) {
    __syncthreads();
    for (int k = threadIdx.x; k < Nreal; k += blockDim.x)
      if (bacc[k] != 0.f) atomicAdd(dbias + k, bacc[k]);
  }
}

// U[xi][n][c] = (G g G^T)[xi], computed in double
// mode FWD:   g = w[n][c][:, :]            rows n < rows_pad (cout_pad), cols c < cin_pad
// mode DGRAD: g = flip(w[n][c]) transposed roles: U[xi][c][n]
// mode 2 (adjoint form): g = w[n][c] as it is, transposed roles: U[xi][c][n] = the forward transform, transposed
// fused (with mode FWD, 2-D): the same values in the fragment layout of the fused kernel (wino_fused_index)
template <int MT, int R>
__device__ __forceinline__ void wino_filter_body(const float* __restrict__ w, float* __restrict__ U, int cout, int cin,
                                                 int rows, int cols, int kdt, int dgrad, long long total,
                                                 long long first, long long step, bool fused = false) {
  using W = WT<MT, R>;
  constexpr int A = W::A;
  // U[xi][row][dz][col]: the batched GEMM sees kdt "taps" along z (1 for 2-D layers)
  const long long plane = (long long)rows * kdt * cols;
  for (long long i = first; i < total; i += step) {
    const int col = (int)(i % cols);
    const long long q = i / cols;
    const int dz = (int)(q % kdt), row = (int)(q / kdt);
    // dgrad 1: the flipped filter (data gradient as a convolution); dgrad 2 (adjoint form): transposed roles and
    // reversed z taps — the z taps stay an ordinary contraction, i.e. a transposed convolution along z — but the
    // (y, x) filter as it is: its transform is the forward's, used transposed
    const bool flip = dgrad == 1;
    const int n = dgrad ? col : row, c = dgrad ? row : col;
    const int wz = dgrad ? kdt - 1 - dz : dz;
    double g[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int s2 = 0; s2 < R; ++s2) {
        double v = 0.0;
        if (n < cout && c < cin) {
          const int rr = flip ? R - 1 - r : r, ss = flip ? R - 1 - s2 : s2;
          v = (double)w[(((long long)n * cin + c) * kdt + wz) * (R * R) + rr * R + ss];
        }
        g[r][s2] = v;
      }
    double tt[A][R];
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int s2 = 0; s2 < R; ++s2) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) acc += W::G[r][k] * g[k][s2];
        tt[r][s2] = acc;
      }
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int qq = 0; qq < A; ++qq) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) acc += tt[r][k] * W::G[qq][k];
        if (fused) U[wino_fused_index(r * A + qq, row, col, cols, A * A)] = (float)acc;
        else U[(r * A + qq) * plane + i] = (float)acc;
      }
  }
}

template <int MT, int R>
__global__ void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int cout,
                                   int cin, int rows, int cols, int kdt, int dgrad, long long total, bool fused) {
  wino_filter_body<MT, R>(w, U, cout, cin, rows, cols, kdt, dgrad, total,
                          (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x, fused);
}

// every packing of a step in one launch: blockIdx.y = job (clx_pack_weights_batch)
__global__ __launch_bounds__(256) void pack_batch_kernel(const clx_pack_job* __restrict__ jobs) {
  const clx_pack_job j = jobs[blockIdx.y];
  const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
  if (j.mode == CLX_PACK_FWD || j.mode == CLX_PACK_DGRAD) {
    const long long total = j.mode == CLX_PACK_FWD ? (long long)j.cout * j.taps * j.cin_pad
                                                   : (long long)j.cin_pad * j.taps * j.cout_pad;
    for (long long i = first; i < total; i += step) {
      float v = 0.f;
      if (j.mode == CLX_PACK_FWD) {
        const int c = (int)(i % j.cin_pad);
        const long long t = i / j.cin_pad;
        const int tap = (int)(t % j.taps), n = (int)(t / j.taps);
        if (c < j.cin) v = j.w[((long long)n * j.cin + c) * j.taps + tap];
      } else {
        const int n = (int)(i % j.cout_pad);
        const long long t = i / j.cout_pad;
        const int tap = (int)(t % j.taps), c = (int)(t / j.taps);
        if (n < j.cout && c < j.cin) v = j.w[((long long)n * j.cin + c) * j.taps + (j.taps - 1 - tap)];
      }
      j.wp[i] = v;
    }
    return;
  }
  const bool fused = j.mode == CLX_PACK_WINO4_FUSED;
  const bool four = j.mode == CLX_PACK_WINO4_FWD || j.mode == CLX_PACK_WINO4_DGRAD || j.mode == CLX_PACK_WINO4_ADJOINT || fused;
  const int dgrad = (j.mode == CLX_PACK_WINO_DGRAD || j.mode == CLX_PACK_WINO4_DGRAD) ? 1
                    : j.mode == CLX_PACK_WINO4_ADJOINT                               ? 2 : 0;
  const int ksize = (j.taps == 9 || j.taps == 27) ? 3 : 2, kd = (j.taps == 27 || j.taps == 8) ? ksize : 1;
  const int rows = dgrad ? j.cin_pad : j.cout_pad, cols = dgrad ? j.cout_pad : j.cin_pad;
  const long long total = (long long)rows * kd * cols;
  if (ksize == 2) wino_filter_body<4, 2>(j.w, j.wp, j.cout, j.cin, rows, cols, kd, dgrad, total, first, step, fused);
  else if (four) wino_filter_body<4, 3>(j.w, j.wp, j.cout, j.cin, rows, cols, kd, dgrad, total, first, step, fused);
  else wino_filter_body<2, 3>(j.w, j.wp, j.cout, j.cin, rows, cols, kd, dgrad, total, first, step);
}

// dw[n][c][R x R] = G^T dU G
template <int MT, int R>
__global__ void wino_unpack_kernel(const float* __restrict__ dU, float* __restrict__ dw, int cout,
                                   int cin, int rows, int cin_pad, int kdt, long long total) {
  using W = WT<MT, R>;
  constexpr int A = W::A;
  // dU[xi][dz][row][cin_pad] (the weight-gradient kernel's [tap][n][c]) -> dw[n][c][dz][R][R]
  const long long plane = (long long)kdt * rows * cin_pad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int dz = (int)(i % kdt);
    const long long q = i / kdt;
    const int c = (int)(q % cin), n = (int)(q / cin);
    const float* src = dU + ((long long)dz * rows + n) * cin_pad + c;
    double t[R][A];
#pragma unroll
    for (int s2 = 0; s2 < A; ++s2) {
      double u[A];
#pragma unroll
      for (int r = 0; r < A; ++r) u[r] = (double)src[(r * A + s2) * plane];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < A; ++r) acc += W::G[r][k] * u[r];
        t[k][s2] = acc;
      }
    }
    float* dst = dw + i * (R * R);
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int l = 0; l < R; ++l) {
        double acc = 0.0;
#pragma unroll
        for (int s2 = 0; s2 < A; ++s2) acc += t[k][s2] * W::G[s2][l];
        dst[k * R + l] = (float)acc;
      }
  }
}

bool applicable(const clx_conv_desc* d) {
  // 2-D layers (KD = 1, one plane) or 3-D layers with a cubic kernel: the Winograd transform is
  // applied in (y, x) per z plane, the z taps stay a plain contraction inside the batched GEMMs
  if (d->nsrc != 1 || d->KH != d->KW) return false;
  if (d->KD == 1) {
    if (d->ID != 1 || d->PD != 0) return false;
  } else {
    if (d->KD != d->KH || d->algo != CLX_ALGO_WINOGRAD4 || d->PD != d->PH) return false;
  }
  if (d->KH != 3 && !(d->KH == 2 && d->algo == CLX_ALGO_WINOGRAD4)) return false;
  if (d->PH != d->PW || (d->PH != 0 && d->PH != d->KH - 1)) return false;
  const clx_src& S = d->src[0];
  if (S.fz != 1 || S.fy != 1 || S.fx != 1) return false;
  if (d->KD == 1 && (S.D != 1 || S.oz != 0)) return false;
  if (S.C % 4 != 0 || d->N <= 0) return false;
  return true;
}

// output tile size selected by the descriptor: CLX_ALGO_WINOGRAD = F(2x2), CLX_ALGO_WINOGRAD4 = F(4x4)
inline int tile_of(const clx_conv_desc* d) { return d->algo == CLX_ALGO_WINOGRAD4 ? 4 : 2; }

// geometry of the (y, x) transforms; `planes` z planes per batch element are separate images
Geom geom(const clx_conv_desc* d, int mt, int planes) {
  Geom g;
  const clx_src& S = d->src[0];
  g.B = d->B * planes; g.SH = S.H; g.SW = S.W; g.oy = S.oy; g.ox = S.ox;
  g.ID = planes; g.SD = S.D; g.oz = S.oz;
  g.IH = d->IH; g.IW = d->IW; g.P = d->PH;
  g.OH = d->IH + 2 * d->PH - (d->KH - 1); g.OW = d->IW + 2 * d->PW - (d->KW - 1);
  g.th = (g.OH + mt - 1) / mt; g.tw = (g.OW + mt - 1) / mt;
  g.T = (long long)g.B * g.th * g.tw;
  return g;
}
inline int out_planes(const clx_conv_desc* d) { return d->ID + 2 * d->PD - (d->KD - 1); }

inline int pad4(int n) { return (n + 3) / 4 * 4; }

// Does this layer's transform-domain arithmetic run in the split precision (clx_conv_desc.precision; gemm_sp.hip)?  ONE
// rule for the forward, data-gradient and weight-gradient calls of a layer — they hand V and A dY A^T to each other as
// P3 planes (vcache, dy_vcache, the workspace) —: 2-D layer, both channel counts multiples of 128.
inline bool wino_sp(const clx_conv_desc* d) {
  if (d->precision != CLX_PREC_F32X3BF16 || d->KD != 1 || d->ID != 1 || d->N % 128 != 0 || d->src[0].C % 128 != 0) return false;
  // the weight-gradient product addresses its operand planes with 32-bit offsets: one transform point's planes stay below
  // 4 GB (the tiles of the layer's LARGER grid — the data-gradient form's — decide for all three calls)
  const int mt = d->algo == CLX_ALGO_WINOGRAD ? 2 : 4;
  // (the layer's forward INPUT extent, whichever form `d` is: >= every grid one of its calls tiles)
  const long long eh = d->IH + (d->PH ? d->KH - 1 : 0), ew = d->IW + (d->PW ? d->KW - 1 : 0);
  const long long tiles = (long long)d->B * ((eh + mt - 1) / mt) * ((ew + mt - 1) / mt);
  const int ch = d->N > d->src[0].C ? d->N : d->src[0].C;
  return sp::planes_bytes(tiles, ch) < (1ll << 32);
}
// grid of a plane-writing transform: 8 ceil(rows / 8) rows x C / 4 items, whole wavefronts
inline long long sp_items(long long rows, int C) { return (rows + 7) / 8 * 8 * (C / 4); }

// the batched GEMM descriptor: A^2 problems over the transformed tensor V [B][ID planes][tiles][C]:
// a (KD, 1, 1) "convolution" along z — a plain [T x C] . [C x N] product for 2-D layers
clx_conv_desc gemm_desc(float* V, int C, const clx_conv_desc* d, const Geom& gin) {
  clx_conv_desc gd = {};
  gd.nsrc = 1;
  gd.src[0].ptr = V; gd.src[0].C = C; gd.src[0].ld = C;
  gd.src[0].D = d->ID; gd.src[0].H = 1; gd.src[0].W = gin.th * gin.tw;
  gd.src[0].fz = gd.src[0].fy = gd.src[0].fx = 1;
  gd.B = d->B; gd.ID = d->ID; gd.IH = 1; gd.IW = gin.th * gin.tw;
  gd.KD = d->KD; gd.KH = gd.KW = 1;
  gd.PD = d->PD;
  return gd;
}

template <int MT, int R>
int wino_fwd_t(const clx_conv_desc* d, hipStream_t st) {
  constexpr int AA = WT<MT, R>::A * WT<MT, R>::A;
  const Geom gin = geom(d, MT, d->ID), gout = geom(d, MT, out_planes(d));
  const clx_src& S = d->src[0];
  const int C = S.C, Np = pad4(d->N);
  const bool spx = wino_sp(d);
  CLX_REQUIRE(!spx || d->wplanes != nullptr, "clx_conv_fwd(winograd): precision f32x3bf16 needs wplanes (the planes of wpack)");
  float* V = d->vcache ? (float*)d->vcache : (float*)d->workspace;
  const int* list = d->tile_list;
  if (list != nullptr) {
    CLX_REQUIRE(d->KD == 1 && d->ID == 1 && d->PH == 0 && d->vcache == nullptr && !d->accumulate && d->tile_count >= 0 &&
                    d->tile_count <= gin.T && gin.T == gout.T,
                "clx_conv_fwd(winograd): tile_list needs a 2-D valid convolution without vcache / accumulate and at most "
                "th * tw * B tiles");
    if (d->tile_count == 0) return CLX_OK;
  }
  // (with a tile list the transforms and the products see tile_count tiles, stored compactly)
  const long long Tin = list ? d->tile_count : gin.T, Tout = list ? d->tile_count : gout.T;
  // split precision: V as P3 planes (6 bytes per element), one plane set per xi
  const long long pbV = spx ? sp::planes_bytes(Tin, C) : 0;
  float* M = spx ? (float*)((char*)d->workspace + AA * sp::planes_bytes(gin.T, C)) : (float*)d->workspace + AA * Tin * C;
  if (!(d->vcache && d->vcache_valid)) {     // (a weight-gradient call may have left V: dy_vcache)
    if (spx) {
      const long long tot_in = sp_items(Tin, C);
      int rc = clx_sp_zero_tail(V, Tin, C, AA, pbV, st);
      if (rc) return rc;
      CLX_LAUNCH_KIND(CLX_PROF_WINO_TRANSFORM, (wino_input_kernel<MT, R, true>), dim3(grid_for(tot_in, 256)), dim3(256), 0, st, S.ptr, S.ld, C / 4,
                      gin, V, tot_in, list, Tin, pbV);
    } else {
      const long long tot_in = Tin * (C / 4);
      CLX_LAUNCH_KIND(CLX_PROF_WINO_TRANSFORM, (wino_input_kernel<MT, R>), dim3(grid_for(tot_in, 256)), dim3(256), 0, st, S.ptr, S.ld, C / 4, gin, V,
                      tot_in, list, Tin, 0ll);
    }
  }
  if (spx) {
    clx_conv_desc ep = {};
    ep.out = M; ep.ld_out = Np;
    const int rc = clx_sp_launch(V, d->wplanes, (int)Tin, d->N, C, Tin, AA, pbV, sp::planes_bytes(Np, C), Tout * Np, &ep, st);
    if (rc) return rc;
  } else {
    clx_conv_desc gd = gemm_desc(V, C, d, gin);
    if (list) {                                // the listed tiles as one row of tiles of one image
      gd.B = 1;
      gd.src[0].W = gd.IW = (int)Tin;
    }
    gd.N = d->N; gd.wpack = d->wpack; gd.out = M; gd.ld_out = Np;
    const int rc = clx_igemm_launch(&gd, AA, Tin * C, (long long)Np * d->KD * C, Tout * Np, st);
    if (rc) return rc;
  }
  const long long tot_out = Tout * (Np / 4);
  // the bit forms need whole words per lane group: channel count a multiple of 32
  CLX_REQUIRE((d->gate_out == nullptr && d->mask_bits == nullptr) || Np % 32 == 0,
              "clx_conv_fwd(winograd): gate_out / mask_bits need a channel count that is a multiple of 32");
  CLX_REQUIRE(d->gate_out == nullptr || d->relu, "clx_conv_fwd(winograd): gate_out needs relu");
  if (d->pool_out != nullptr) {
    CLX_REQUIRE(d->KD == 1 && d->ID == 1 && gout.OH % 2 == 0 && gout.OW % 2 == 0,
                "clx_conv_fwd(winograd): pool_out needs a 2-D layer with even output height and width");
    CLX_REQUIRE(d->N % 4 == 0 && d->ld_pool % 4 == 0 && d->ld_pool >= d->N && ((uintptr_t)d->pool_out & 15) == 0 &&
                    d->mask == nullptr && d->mask_bits == nullptr && !d->accumulate,
                "clx_conv_fwd(winograd): pool_out needs N %% 4 == 0, an aligned ld_pool >= N, and no mask / accumulate");
  }
  CLX_LAUNCH_KIND(CLX_PROF_WINO_TRANSFORM, (wino_output_kernel<MT, R>), dim3(grid_for(tot_out, 256)), dim3(256), 0, st, M, Np / 4, gout, d->bias,
                  d->relu, d->mask, d->ld_mask, d->out, d->ld_out, d->N, d->accumulate, d->gate_out, d->ld_gate, d->mask_bits,
                  d->ld_mask_bits, d->pool_out, d->ld_pool, tot_out, list, Tout);
  CLX_CHECK_LAUNCH("clx_conv_fwd(winograd)");
  return CLX_OK;
}

template <int MT, int R>
int wino_wgrad_t(const clx_conv_desc* d, const float* dy, int ld_dy, float* dwpack, float* dbias, hipStream_t st) {
  constexpr int AA = WT<MT, R>::A * WT<MT, R>::A;
  const Geom gin = geom(d, MT, d->ID), gout = geom(d, MT, out_planes(d));
  const clx_src& S = d->src[0];
  const int C = S.C, N = d->N;     // N is a multiple of 4 (validated by clx_conv_wgrad)
  const bool spx = wino_sp(d);
  float* V = d->vcache ? (float*)d->vcache : (float*)d->workspace;
  const long long pbV = spx ? sp::planes_bytes(gin.T, C) : 0, pbM = spx ? sp::planes_bytes(gout.T, N) : 0;
  float* Md = spx ? (float*)((char*)d->workspace + AA * pbV) : (float*)d->workspace + AA * gin.T * C;
  if (!(d->vcache && d->vcache_valid)) {
    if (spx) {
      const long long tot_in = sp_items(gin.T, C);
      int rc = clx_sp_zero_tail(V, gin.T, C, AA, pbV, st);
      if (rc) return rc;
      CLX_LAUNCH_KIND(CLX_PROF_WINO_TRANSFORM, (wino_input_kernel<MT, R, true>), dim3(grid_for(tot_in, 256)), dim3(256), 0, st, S.ptr, S.ld, C / 4,
                      gin, V, tot_in, (const int*)nullptr, 0ll, pbV);
    } else {
      const long long tot_in = gin.T * (C / 4);
      CLX_LAUNCH_KIND(CLX_PROF_WINO_TRANSFORM, (wino_input_kernel<MT, R>), dim3(grid_for(tot_in, 256)), dim3(256), 0, st, S.ptr, S.ld, C / 4, gin, V,
                      tot_in, (const int*)nullptr, 0ll, 0ll);
    }
  }
  if (spx) {
    const int rc = clx_sp_zero_tail(Md, gout.T, N, AA, pbM, st);
    if (rc) return rc;
  }
  if (d->dy_vcache) {
    // the data gradient of this layer follows: its input transform of dY comes out of the same pass
    Geom gd = gout;                       // images = B x output planes, stored grid = the dY grid
    gd.SH = gout.OH; gd.SW = gout.OW; gd.oy = gd.ox = 0;
    gd.ID = gd.SD = out_planes(d); gd.oz = 0;
    gd.IH = gout.OH; gd.IW = gout.OW; gd.P = R - 1;
    gd.OH = gout.OH + R - 1; gd.OW = gout.OW + R - 1;
    gd.th = (gd.OH + MT - 1) / MT; gd.tw = (gd.OW + MT - 1) / MT;
    gd.T = (long long)gd.B * gd.th * gd.tw;
    const long long tot = spx ? sp_items(gd.T, N) : gd.T * (N / 4);
    int blocks = grid_for(tot, 256);
    if (dbias && blocks > 2048) blocks = 2048;
    if (spx) {
      const long long pbD = sp::planes_bytes(gd.T, N);
      const int rc = clx_sp_zero_tail(d->dy_vcache, gd.T, N, AA, pbD, st);
      if (rc) return rc;
      CLX_LAUNCH_KIND(CLX_PROF_WINO_TRANSFORM, (wino_dy_dual_kernel<MT, R, true>), dim3(blocks), dim3(256), (size_t)N * sizeof(float), st, dy, ld_dy,
                      N / 4, gd, gout, (float*)d->dy_vcache, Md, dbias, N, tot, pbD, pbM);
    } else {
      CLX_LAUNCH_KIND(CLX_PROF_WINO_TRANSFORM, (wino_dy_dual_kernel<MT, R>), dim3(blocks), dim3(256), (size_t)N * sizeof(float), st, dy, ld_dy, N / 4,
                      gd, gout, (float*)d->dy_vcache, Md, dbias, N, tot, 0ll, 0ll);
    }
  } else {
    const long long tot_dy = spx ? sp_items(gout.T, N) : gout.T * (N / 4);
    int blocks = grid_for(tot_dy, 256);
    if (blocks > 2048) blocks = 2048;
    if (spx)
      CLX_LAUNCH_KIND(CLX_PROF_WINO_TRANSFORM, (wino_dy_kernel<MT, R, true>), dim3(blocks), dim3(256), (size_t)N * sizeof(float), st, dy, ld_dy, N / 4,
                      gout, Md, dbias, N, tot_dy, pbM);
    else
      CLX_LAUNCH_KIND(CLX_PROF_WINO_TRANSFORM, (wino_dy_kernel<MT, R>), dim3(blocks), dim3(256), (size_t)N * sizeof(float), st, dy, ld_dy, N / 4, gout,
                      Md, dbias, N, tot_dy, 0ll);
  }
  if (spx) {
    CLX_REQUIRE(gin.T == gout.T, "clx_conv_wgrad(winograd): tile counts of input and output differ");
    CLX_REQUIRE(d->det_turns == nullptr, "clx_conv_wgrad: the reproducible mode exists for the float32 precision only");
    const int rc = clx_sp_wgrad_launch(Md, V, gin.T, N, C, AA, pbM, pbV, (long long)N * C, dwpack, C, st);
    if (rc) return rc;
    CLX_CHECK_LAUNCH("clx_conv_wgrad(winograd, split precision)");
    return CLX_OK;
  }
  clx_conv_desc gd = gemm_desc(V, C, d, gin);
  gd.N = N;
  gd.det_turns = d->det_turns;            // reproducible mode: the xi products add their slices in order
  const int rc = clx_wgrad_launch(&gd, Md, N, dwpack, nullptr, AA, gin.T * C, gout.T * N,
                                  (long long)d->KD * N * C, st);
  if (rc) return rc;
  CLX_CHECK_LAUNCH("clx_conv_wgrad(winograd)");
  return CLX_OK;
}

}  // namespace

extern "C" size_t clx_conv_workspace_bytes(const clx_conv_desc* d, int pass) {
  if (d == nullptr || !applicable(d)) return 0;
  if (pass == CLX_PASS_WGRAD && d->PH != 0) return 0;
  const int mt = tile_of(d), a = mt + d->KH - 1;
  const Geom gin = geom(d, mt, d->ID), gout = geom(d, mt, out_planes(d));
  if (gout.OH <= 0 || gout.OW <= 0 || out_planes(d) <= 0 || gin.T >= (1ll << 31) || gout.T >= (1ll << 31)) return 0;
  const long long C = d->src[0].C, N = pad4(d->N);
  if (wino_sp(d)) {
    // V (and, for the weight gradient, A dY A^T) as P3 planes; the products' results stay float32.  (The adjoint data
    // gradient's products [36][tiles][C] float32 fit the V region: 4 <= 6 bytes per element.)
    const long long v = sp::planes_bytes(gin.T, (int)C);
    const long long m = pass == CLX_PASS_WGRAD ? sp::planes_bytes(gout.T, (int)N) : gout.T * N * (long long)sizeof(float);
    return (size_t)(a * a * (v + m));
  }
  return (size_t)(a * a * (gin.T * C + gout.T * N)) * sizeof(float);
}

extern "C" size_t clx_conv_vcache_bytes(const clx_conv_desc* d, int which) {
  if (d == nullptr || !applicable(d)) return 0;
  const int mt = tile_of(d), a = mt + d->KH - 1, R = d->KH;
  const Geom gin = geom(d, mt, d->ID), gout = geom(d, mt, out_planes(d));
  const long long C = d->src[0].C, N = pad4(d->N);
  if (which == 0) return wino_sp(d) ? (size_t)(a * a * sp::planes_bytes(gin.T, (int)C)) : (size_t)(a * a * gin.T * C) * sizeof(float);
  // dy_vcache: the (K - 1)-padded input transform of dY over the tiles of dX
  const long long th = (gout.OH + R - 1 + mt - 1) / mt, tw = (gout.OW + R - 1 + mt - 1) / mt;
  const long long Td = (long long)gout.B * th * tw;
  return wino_sp(d) ? (size_t)(a * a * sp::planes_bytes(Td, (int)N)) : (size_t)(a * a * Td * N) * sizeof(float);
}

// the adjoint data gradient (clx_conv_desc.adjoint): d is the data-gradient descriptor — source dY (C = the layer's
// padded output channels, extent = the forward output), padding 2, N = the layer's padded input channels
static int wino_adjoint(const clx_conv_desc* d, hipStream_t st) {
  CLX_REQUIRE(d->algo == CLX_ALGO_WINOGRAD4 && d->KH == 3 && d->KW == 3 && d->PH == 2 && d->PW == 2 && d->nsrc == 1 &&
                  ((d->KD == 1 && d->ID == 1 && d->PD == 0) || (d->KD == 3 && d->PD == 2)),
              "clx_conv_fwd(adjoint): a 3x3 (2-D) or 3x3x3 layer in its data-gradient form (padding 2) with "
              "CLX_ALGO_WINOGRAD4 only");
  CLX_REQUIRE(d->bias == nullptr && !d->relu && !d->accumulate && d->gate_out == nullptr,
              "clx_conv_fwd(adjoint): only the ReLU-gate epilogues (mask / mask_bits) exist");
  CLX_REQUIRE(d->N % 4 == 0 && d->src[0].C % 4 == 0, "clx_conv_fwd(adjoint): channel counts must be multiples of 4");
  const int OHf = d->IH, OWf = d->IW;                  // the forward layer's output extent = extent of dY
  const int th = (OHf + 3) / 4, tw = (OWf + 3) / 4;
  const int planes_dy = d->ID, planes_dx = d->ID + 2 * d->PD - (d->KD - 1);      // z planes of dY / of dX
  const long long Tdy = (long long)d->B * planes_dy * th * tw, Tdx = (long long)d->B * planes_dx * th * tw;
  const int Nf = d->src[0].C, Cp = d->N;
  const bool spx = wino_sp(d);           // (N = Cp, C = Nf, extent of dY: the rule gives what it gave the layer's forward form)
  const size_t need = spx ? (size_t)36 * (sp::planes_bytes(Tdx, Cp) + sp::planes_bytes(Tdy, Nf))
                          : (size_t)36 * (Tdx * Cp + Tdy * Nf) * sizeof(float);
  CLX_REQUIRE(d->workspace != nullptr && d->workspace_bytes >= need && ((uintptr_t)d->workspace & 15) == 0,
              "clx_conv_fwd(adjoint): needs the %zu workspace bytes of the layer's weight gradient (%zu given)", need,
              d->workspace_bytes);
  CLX_REQUIRE(d->mask_bits == nullptr || Cp % 32 == 0, "clx_conv_fwd(adjoint): mask_bits needs whole words per pixel");
  float* P = (float*)d->workspace;                     // [36][Tdx][Cp], over what was the weight gradient's V region
  float* Mdy = P + (size_t)36 * Tdx * Cp;              // [36][Tdy][Nf]: left there by clx_conv_wgrad
  if (spx) {
    // A dY A^T lies behind the V region as planes; the products from them, float32 results over the V region
    CLX_REQUIRE(d->wplanes != nullptr, "clx_conv_fwd(adjoint): precision f32x3bf16 needs wplanes");
    const char* Mp = (const char*)d->workspace + 36 * sp::planes_bytes(Tdx, Cp);
    clx_conv_desc ep = {};
    ep.out = P; ep.ld_out = Cp;
    const int rc = clx_sp_launch(Mp, d->wplanes, (int)Tdy, Cp, Nf, Tdy, 36, sp::planes_bytes(Tdy, Nf), sp::planes_bytes(Cp, Nf),
                                 Tdx * Cp, &ep, st);
    if (rc) return rc;
  } else {
  // the batched product as a (KD, 1, 1) transposed convolution along z over the tiles
  clx_conv_desc gd = {};
  gd.nsrc = 1;
  gd.src[0].ptr = Mdy; gd.src[0].C = Nf; gd.src[0].ld = Nf;
  gd.src[0].D = planes_dy; gd.src[0].H = 1; gd.src[0].W = th * tw;
  gd.src[0].fz = gd.src[0].fy = gd.src[0].fx = 1;
  gd.B = d->B; gd.ID = planes_dy; gd.IH = 1; gd.IW = th * tw;
  gd.KD = d->KD; gd.KH = gd.KW = 1; gd.PD = d->PD;
  gd.N = Cp; gd.wpack = d->wpack; gd.out = P; gd.ld_out = Cp;
  const int rc = clx_igemm_launch(&gd, 36, Tdy * Nf, (long long)Cp * d->KD * Nf, Tdx * Cp, st);
  if (rc) return rc;
  }
  const int IH = OHf + 2, IW = OWf + 2;
  const int nseg = ((IH + 3) / 4 + ADJ_SEG - 1) / ADJ_SEG;
  const long long total = (long long)d->B * planes_dx * nseg * ((IW + 3) / 4) * (Cp / 4);
  CLX_LAUNCH_KIND(CLX_PROF_WINO_TRANSFORM, wino_adjoint_output_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, P, Cp / 4,
                  d->B * planes_dx, th, tw, IH, IW, d->mask, d->ld_mask, d->mask_bits, d->ld_mask_bits, d->out, d->ld_out, Cp, total);
  CLX_CHECK_LAUNCH("clx_conv_fwd(winograd adjoint)");
  return CLX_OK;
}

int clx_wino_fwd(const clx_conv_desc* d, hipStream_t st) {
  if (d->adjoint) return wino_adjoint(d, st);
  CLX_REQUIRE(applicable(d), "clx_conv_fwd: Winograd does not apply to this geometry");
  const size_t need = clx_conv_workspace_bytes(d, CLX_PASS_FWD);
  CLX_REQUIRE(d->workspace != nullptr && d->workspace_bytes >= need && need > 0,
              "clx_conv_fwd: Winograd needs %zu workspace bytes (%zu given)", need, d->workspace_bytes);
  CLX_REQUIRE(((uintptr_t)d->workspace & 15) == 0 && ((uintptr_t)d->vcache & 15) == 0,
              "clx_conv_fwd: workspace / vcache must be 16-byte aligned");
  if (d->KH == 2) return wino_fwd_t<4, 2>(d, st);
  return tile_of(d) == 4 ? wino_fwd_t<4, 3>(d, st) : wino_fwd_t<2, 3>(d, st);
}

int clx_wino_wgrad(const clx_conv_desc* d, const float* dy, int ld_dy, float* dwpack, float* dbias,
                   hipStream_t st) {
  CLX_REQUIRE(applicable(d) && d->PH == 0, "clx_conv_wgrad: Winograd does not apply to this geometry");
  const size_t need = clx_conv_workspace_bytes(d, CLX_PASS_WGRAD);
  CLX_REQUIRE(d->workspace != nullptr && d->workspace_bytes >= need && need > 0,
              "clx_conv_wgrad: Winograd needs %zu workspace bytes (%zu given)", need, d->workspace_bytes);
  CLX_REQUIRE(((uintptr_t)d->workspace & 15) == 0 && ((uintptr_t)d->vcache & 15) == 0,
              "clx_conv_wgrad: workspace / vcache must be 16-byte aligned");
  CLX_REQUIRE(d->N <= 8192, "clx_conv_wgrad: too many output channels for the Winograd bias accumulator");
  if (d->KH == 2) return wino_wgrad_t<4, 2>(d, dy, ld_dy, dwpack, dbias, st);
  return tile_of(d) == 4 ? wino_wgrad_t<4, 3>(d, dy, ld_dy, dwpack, dbias, st)
                         : wino_wgrad_t<2, 3>(d, dy, ld_dy, dwpack, dbias, st);
}

// dgrad: 0 forward, 1 flipped filter, 2 adjoint form, 3 forward in the fused kernel's fragment layout
int clx_wino_pack(const float* w, float* wp, int cout, int cin, int cin_pad, int cout_pad, int dgrad,
                  int tile, int ksize, int kd, hipStream_t st) {
  const bool fused = dgrad == 3;
  if (fused) dgrad = 0;
  const int rows = dgrad ? cin_pad : cout_pad, cols = dgrad ? cout_pad : cin_pad;
  const long long total = (long long)rows * kd * cols;
  const int grid = grid_for(total, 256);
  if (ksize == 2) wino_filter_kernel<4, 2><<<grid, 256, 0, st>>>(w, wp, cout, cin, rows, cols, kd, dgrad, total, fused);
  else if (tile == 4) wino_filter_kernel<4, 3><<<grid, 256, 0, st>>>(w, wp, cout, cin, rows, cols, kd, dgrad, total, fused);
  else wino_filter_kernel<2, 3><<<grid, 256, 0, st>>>(w, wp, cout, cin, rows, cols, kd, dgrad, total, fused);
  return CLX_OK;
}

extern "C" int clx_pack_weights_batch(const clx_pack_job* jobs, int njobs, long long max_total, clx_stream stream) {
  CLX_REQUIRE(jobs != nullptr && njobs > 0 && njobs < 65536 && max_total > 0, "clx_pack_weights_batch: bad arguments");
  CLX_REQUIRE(((uintptr_t)jobs & 7) == 0, "clx_pack_weights_batch: the job table must be 8-byte aligned");
  long long gx = (max_total + 255) / 256;
  if (gx > 512) gx = 512;                     // (the large jobs loop: 20 jobs x 512 blocks already cover the chip several times)
  pack_batch_kernel<<<dim3((unsigned)gx, (unsigned)njobs), 256, 0, (hipStream_t)stream>>>(jobs);
  CLX_CHECK_LAUNCH("clx_pack_weights_batch");
  return CLX_OK;
}

extern "C" int clx_unpack_wgrad_wino(const float* du, float* dw, int cout, int cin, int rows,
                                     int cin_pad, int tile, int ksize, int kd, clx_stream stream) {
  CLX_REQUIRE(du && dw, "clx_unpack_wgrad_wino: null pointer");
  CLX_REQUIRE(cout > 0 && cin > 0 && rows >= cout && cin_pad >= cin, "clx_unpack_wgrad_wino: bad extents");
  CLX_REQUIRE((ksize == 3 && (tile == 2 || tile == 4)) || (ksize == 2 && tile == 4),
              "clx_unpack_wgrad_wino: (tile, ksize) must be (2, 3), (4, 3) or (4, 2)");
  CLX_REQUIRE(kd == 1 || (kd == ksize && tile == 4), "clx_unpack_wgrad_wino: kd must be 1 or ksize (F(4x4) only)");
  const long long total = (long long)cout * cin * kd;
  const int grid = grid_for(total, 256);
  hipStream_t st = (hipStream_t)stream;
  if (ksize == 2) wino_unpack_kernel<4, 2><<<grid, 256, 0, st>>>(du, dw, cout, cin, rows, cin_pad, kd, total);
  else if (tile == 4) wino_unpack_kernel<4, 3><<<grid, 256, 0, st>>>(du, dw, cout, cin, rows, cin_pad, kd, total);
  else wino_unpack_kernel<2, 3><<<grid, 256, 0, st>>>(du, dw, cout, cin, rows, cin_pad, kd, total);
  CLX_CHECK_LAUNCH("clx_unpack_wgrad_wino");
  return CLX_OK;
}
