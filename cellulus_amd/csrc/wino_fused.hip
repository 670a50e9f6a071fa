// Winograd F(4x4, 3x3) / F(4x4, 2x2) for 2-D valid layers as ONE kernel per layer (CLX_ALGO_WINOGRAD4_FUSED): the
// transformed tensors V = B^T d B and M = V . U^T never exist in HBM.
//
// wino.hip runs a layer as input transform -> a^2 batched GEMMs -> output transform and moves V and M (a^2 / 16 = 2.25x
// the activation each) through HBM twice: ~5.5x the layer's algorithmic bytes, 20 % of an inference tile and 13 % of a
// training step.  Here a workgroup of 8 waves owns FT = 32 output tiles x FN = 64 output channels and
// ALL a^2 = 36 (25) transform-domain products of them:
//
//   K loop over chunks of FK = 8 input channels:
//     raw patches  : every wave fetches the a x a input patches of ITS 4 tiles (32 bytes per pixel) into its private LDS
//                    region — global -> registers (a chunk ahead) -> LDS
//     B^T d B      : two passes inside that region, wave-local (LDS operations of one wave execute in order: no block
//                    barrier): columns (in place, transposing), then rows -> V[xi][k half][tile][4] in the block's
//                    double-buffered A-operand image
//     products     : wave (g, nh) owns xi = g, g + 4, ... and output channels 32 nh .. 32 nh + 31: per xi one
//                    ds_read_b128 of the A fragment feeds 4 v_mfma_f32_32x32x2_f32 whose B fragments come STRAIGHT
//                    FROM GLOBAL MEMORY into registers — every weight element is used by exactly one wave, so LDS
//                    staging buys nothing; CLX_PACK_WINO4_FUSED stores them as 1-KB pieces in fragment order and a
//                    fragment is re-loaded for the next chunk as soon as its MFMAs are issued (a whole chunk of latency
//                    tolerance on 36 registers).  9 xi x 16 = 144 accumulator registers per lane, two waves per SIMD.
//     one __syncthreads per chunk; waves 0-3 transform first and multiply second, waves 4-7 the other way round, so
//     that the VALU / LDS work of one wave of a SIMD runs under the MFMAs of the other.
//   epilogue (twice, 32 channels each): accumulators -> LDS [xi][tile][33]; a thread per (tile, channel) applies
//     A^T m A, bias / accumulate / ReLU, writes the 4 x 4 outputs (128-byte runs per pixel), the ReLU gate words by
//     ballot (32 lanes = 32 channels = one word) and the 2 x 2 max-pooled outputs.
//
// Per block and channel the MFMAs take 36 x 32 x 64 x 2 / 256 = 576 CU cycles against ~9.2 KB of weights and
// ~1.2 k pixels x 4 B of input from L2: the block order keeps one 64-channel slice of the weights per XCD at a time.
//
// Replaces the same reference calls as wino.hip: nn.Conv2d 3x3 (+ ReLU, + MaxPool2d) of funlib's ConvPass,
// cellulus/models/unet.py:24-51, and the 32 infer-mode forwards of cellulus/models/unet.py:73-100.
#include "clx_common.h"
#include "wino_tables.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

// acc (+)= coef * v with EXPLICIT fused multiply-adds (this file is compiled with -ffp-contract=off): which pass of the
// transform a tile's item lands in depends on the tile's place in its block, and a tile must get the same bits wherever it
// sits (tile lists: tests/test_gpu_wino_fused.py) — left to the compiler, one pass of F(4x4, 2x2) contracted a product
// into its neighbour's add and another did not.
__device__ __forceinline__ void axpy_fma(float& acc, bool& first, float coef, float v) {
  if (coef == 0.f) return;
  if (first) { acc = (coef == 1.f) ? v : (coef == -1.f) ? -v : coef * v; first = false; }
  else if (coef == 1.f) acc += v;
  else if (coef == -1.f) acc -= v;
  else acc = __builtin_fmaf(coef, v, acc);
}

template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

constexpr int FT = 32;   // tiles per block
constexpr int FN = 64;   // output channels per block
constexpr int FK = 8;    // input channels per chunk

struct FusedP {
  const float* x;
  int ld_x, SH, SW, oy, ox, IH, IW;      // source: pixel stride, stored grid, crop offset, logical extent
  int OH, OW, th, tw;                    // output extent, tiles per image
  int T;                                 // tiles to compute (tile_count with a list, else B * th * tw)
  const int* tile_list;
  const float* vf;                       // pre-transformed input in fragment order (wino_pre_kernel), or NULL
  const float* wf;                       // CLX_PACK_WINO4_FUSED weights
  int C, N, nchunks;
  const float* bias;
  int relu, accumulate;
  float* out;
  int ld_out;
  unsigned int* gate_out;
  int ld_gate;
  float* pool_out;
  int ld_pool;
  int ntb, nnb;                          // tile blocks, channel blocks
  int order, gtb, gnb;                   // wino_pre_kernel's block order (launcher) and its padded extents
};

template <int R> struct FusedGeom {
  static constexpr int A = 4 + R - 1;              // patch side: 6 (3x3) or 5 (2x2)
  static constexpr int NXI = A * A;
  static constexpr int XIW = (NXI + 3) / 4;        // xi per wave group: 9 or 7 (the last group of F(4x4, 2x2) holds 6)
  static constexpr int RAW_TILE = A * A * 8;       // floats per tile of the raw / column-transformed patches (dense: a
                                                   // 16-byte piece's place follows from its number alone)
  static constexpr int V_XI = A == 6 ? 268 : 264;  // floats per xi of the A-operand image [2][32][4] + pad: A * V_XI = 8
                                                   // mod 32 spreads the row-transform's stores over the banks
  static constexpr int RAW_ITEMS = 4 * A * A * 2;  // 16-byte pieces a wave fetches per chunk (4 tiles x pixels x 2)
  static constexpr int NRAW = (RAW_ITEMS + 63) / 64;
  static constexpr int ST_ITEMS = 4 * A * 8;       // (tile, column | row, channel) items of a transform pass per wave
  static constexpr int NPASS = (ST_ITEMS + 63) / 64;
  static constexpr int SMEM_LOOP = 2 * FT * RAW_TILE + 2 * NXI * V_XI;
  static constexpr int SMEM_EPI = NXI * FT * 33;
  static constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;
};

__device__ __forceinline__ void decode_tile(const FusedP& p, int tt, int& b, int& ty, int& tx) {
  const int tg = p.tile_list ? p.tile_list[tt] : tt;
  tx = tg % p.tw;
  const int q = tg / p.tw;
  ty = q % p.th;
  b = q / p.th;
}

// Epilogue of both product kernels: the accumulators of xi_j = g + 4 j (wave (g, nh): output channels 32 nh ..) go
// through LDS, 32 channels at a time; a thread per (tile, channel) applies A^T m A, bias / accumulate / ReLU, writes the
// 4 x 4 outputs (128-byte runs per pixel), the ReLU gate words by ballot (32 lanes = 32 channels = one word) and the
// 2 x 2 max-pooled outputs.  smem: NXI * FT * 33 floats (everything the K loop kept in LDS is dead).
template <int R>
__device__ __forceinline__ void fused_epilogue(const FusedP& p, f32x16 (&acc)[FusedGeom<R>::XIW], float* smem, int tid,
                                               int g, int nh, int nb, int t0) {
  using G = FusedGeom<R>;
  using W = WT<4, R>;
  constexpr int A = G::A, NXI = G::NXI, XIW = G::XIW;
  const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
  auto xi_live = [&](int j) { return XIW * 4 == NXI || g + 4 * j < NXI; };
  float* const Ms = smem;                          // [NXI][FT][33]
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
    if (nh == h) {
#pragma unroll
      for (int j = 0; j < XIW; ++j)
        if (xi_live(j)) {
          const int xi = g + 4 * j;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            Ms[(xi * FT + row) * 33 + li] = acc[j][r];
          }
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
      const int item = tid + 512 * it;
      const int n = item & 31, tile = item >> 5;
      const int ng = nb * FN + h * 32 + n;
      const int tt = t0 + tile;
      const bool tile_live = tt < p.T;
      int b, ty, tx;
      decode_tile(p, tile_live ? tt : p.T - 1, b, ty, tx);
      // rows first, one column of m at a time: rr[a][s] = sum_k A^T[a][k] m[k][s]
      float rr[4][A];
      const float* const src = Ms + tile * 33 + n;
#pragma unroll
      for (int s = 0; s < A; ++s) {
        float m[A];
#pragma unroll
        for (int k = 0; k < A; ++k) m[k] = src[(k * A + s) * (FT * 33)];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float acc1 = 0.f;
          bool first = true;
#pragma unroll
          for (int k = 0; k < A; ++k) axpy_fma(acc1, first, W::AT[a][k], m[k]);
          rr[a][s] = acc1;
        }
      }
      const float bv = p.bias ? p.bias[ng] : 0.f;
      float pm[2][2];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int oy = 4 * ty + a;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int ox = 4 * tx + cc;
          const bool live = tile_live && oy < p.OH && ox < p.OW;
          float val = 0.f;
          bool first = true;
#pragma unroll
          for (int k = 0; k < A; ++k) axpy_fma(val, first, W::AT[cc][k], rr[a][k]);
          val += bv;
          const long long m = ((long long)b * p.OH + oy) * p.OW + ox;
          float* const dst = p.out + m * p.ld_out + ng;
          if (p.accumulate && live) val += *dst;
          if (p.relu) val = fmaxf(val, 0.f);
          if (p.gate_out) {               // lanes 0-31 / 32-63: the 32 channels of one word of two tiles
            const unsigned long long bal = __ballot(live && val > 0.f);
            if (n == 0 && live) p.gate_out[m * p.ld_gate + (ng >> 5)] = (unsigned int)(lh ? (bal >> 32) : bal);
          }
          if (live) *dst = val;
          if (p.pool_out) {               // (OH, OW even: a window is whole or absent)
            if ((a & 1) == 0 && (cc & 1) == 0) pm[a >> 1][cc >> 1] = val;
            else pm[a >> 1][cc >> 1] = fmaxf(pm[a >> 1][cc >> 1], val);
            if ((a & 1) && (cc & 1) && live) {
              const long long pmi = ((long long)b * (p.OH >> 1) + (oy >> 1)) * (p.OW >> 1) + (ox >> 1);
              p.pool_out[pmi * p.ld_pool + ng] = pm[a >> 1][cc >> 1];
            }
          }
        }
      }
    }
  }
}

template <int R>
__global__ __launch_bounds__(512, 1) void wino_fused_kernel(const FusedP p) {
  using G = FusedGeom<R>;
  using W = WT<4, R>;
  constexpr int A = G::A, NXI = G::NXI, XIW = G::XIW, RAW_TILE = G::RAW_TILE, V_XI = G::V_XI;
  constexpr int NRAW = G::NRAW, NPASS = G::NPASS;
  __shared__ __attribute__((aligned(16))) float smem[G::SMEM];
  float* const raw = smem;                         // [2][FT][RAW_TILE]
  float* const Vs = smem + 2 * FT * RAW_TILE;      // [2][NXI][V_XI]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);          // (told to be wave-uniform: scalar offsets, no waterfall loops)
  const int g = wid & 3, nh = wid >> 2;            // xi group, channel half (waves w and w + 4 share a SIMD)
  const int li = lane & 31, lh = lane >> 5;
  const int v = xcd_remap(blockIdx.x, p.ntb * p.nnb);
  const int nb = v / p.ntb, tb = v - nb * p.ntb;   // consecutive blocks of an XCD: the same weights, neighbouring tiles
  const int t0 = tb * FT;

  // ---- raw patches: this wave's 4 tiles, 16-byte pieces i = lane + 64 j = (tile, pixel, half), stored at float
  // 4 i of the wave's part of the patch buffer; global byte offsets in 32 bits
  // (buffer loads: a 32-bit lane offset beside a uniform descriptor and a uniform chunk offset — with plain pointers the
  //  compiler kept the offsets as 64-bit pairs, spilled them, and reloaded them at the top of every chunk behind a
  //  vmcnt(0); a piece outside the image (partial last tiles) gets an offset past the descriptor's range and reads zeros)
  typedef int i32x4_ __attribute__((ext_vector_type(4)));
  constexpr uint32_t OOB = 0xffffff00u;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wf), 0, (int)OOB, 0x00020000);
  uint32_t roff[NRAW];
#pragma unroll
  for (int j = 0; j < NRAW; ++j) {
    int i = lane + 64 * j;
    if (i >= G::RAW_ITEMS) i = G::RAW_ITEMS - 1;   // (its LDS store is skipped)
    const int tl = i / (A * A * 2), rem = i - tl * (A * A * 2);
    const int px = rem >> 1;
    const int k = px / A, l = px - k * A;
    int tt = t0 + wid * 4 + tl;
    if (tt >= p.T) tt = p.T - 1;                   // (tiles past the end repeat the last one; never stored)
    int b, ty, tx;
    decode_tile(p, tt, b, ty, tx);
    const int iy = 4 * ty + k, ix = 4 * tx + l;
    roff[j] = (uint32_t)((((long long)b * p.SH + iy + p.oy) * p.SW + ix + p.ox) * p.ld_x + (rem & 1) * 4) * 4u;
    if (iy >= p.IH || ix >= p.IW) roff[j] = OOB;
  }
  float* const raw_w = raw + wid * 4 * RAW_TILE + lane * 4;     // + rb * FT * RAW_TILE + 256 j
  // the pieces travel in two groups through NRG registers: group 0 = pieces [0, NRG), group 1 = the rest
  constexpr int NRG = (NRAW + 1) / 2;
  f32x4 ra[NRG];
  auto load_raw_group = [&](int chunk, int grp) {
#pragma unroll
    for (int j = grp * NRG; j < (grp ? NRAW : NRG); ++j)
      ra[j - grp * NRG] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)roff[j], chunk * (FK * 4), 0));
  };
  auto store_raw_group = [&](int rb, int grp) {
#pragma unroll
    for (int j = grp * NRG; j < (grp ? NRAW : NRG); ++j) {
      if (lane + 64 * j < G::RAW_ITEMS) *reinterpret_cast<f32x4*>(raw_w + rb * (FT * RAW_TILE) + 256 * j) = ra[j - grp * NRG];
    }
  };

  // ---- the two transform passes: items (tile, column l | row r, channel) of this wave's 4 tiles
  int s_col[NPASS], s_row[NPASS], s_v[NPASS];
  bool s_live[NPASS];
#pragma unroll
  for (int q = 0; q < NPASS; ++q) {
    int i = lane + 64 * q;
    s_live[q] = i < G::ST_ITEMS;
    if (!s_live[q]) i = G::ST_ITEMS - 1;
    const int ch = i & 7, combo = i >> 3;
    const int tl = combo / A, lr = combo - tl * A;
    const int tile = wid * 4 + tl;
    s_col[q] = tile * RAW_TILE + lr * 8 + ch;             // pass 1 reads  raw[k][l = lr]   at + k * A * 8
    s_row[q] = tile * RAW_TILE + lr * A * 8 + ch;         // pass 1 writes Wt[l = lr][r]    at + r * 8
                                                          // pass 2 reads  Wt[l][r = lr]    at s_col + l * A * 8
    s_v[q] = lr * A * V_XI + (ch >> 2) * 128 + tile * 4 + (ch & 3);   // pass 2 writes V[xi = lr * A + q] at + q * V_XI
  }
  // columns: w[r][l] = sum_k BT[r][k] d[k][l], written transposed ([l][r]) into the patch's own a x a slots — every
  // read of the wave's passes is issued before the first write (items of one tile straddle passes);
  // rows: V[r][q] = sum_l w[r][l] BT[q][l]
  float dd[NPASS][A];
  auto read_pass = [&](int rb, int q) {            // both passes read the same addresses (raw, then Wt)
#pragma unroll
    for (int k = 0; k < A; ++k) dd[q][k] = raw[rb * (FT * RAW_TILE) + s_col[q] + k * A * 8];
  };
  auto col_out = [&](int rb, int q, int r) {
    float acc1 = 0.f;
    bool first = true;
#pragma unroll
    for (int k = 0; k < A; ++k) axpy_fma(acc1, first, W::BT[r][k], dd[q][k]);
    if (s_live[q]) raw[rb * (FT * RAW_TILE) + s_row[q] + r * 8] = acc1;
  };
  auto row_out = [&](int buf, int q, int qq) {
    float acc1 = 0.f;
    bool first = true;
#pragma unroll
    for (int l = 0; l < A; ++l) axpy_fma(acc1, first, W::BT[qq][l], dd[q][l]);
    if (s_live[q]) Vs[buf * (NXI * V_XI) + s_v[q] + qq * V_XI] = acc1;
  };

  // ---- products: xi_j = g + 4 j; B fragments from global memory, [nb][chunk][xi][nh][lane][4]
  f32x16 acc[XIW];
#pragma unroll
  for (int j = 0; j < XIW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f32x4 bfr[XIW];
  const int wlane = (nh * 64 + lane) * 16;                         // bytes inside a (chunk, xi) piece of 2 KB
  const int wblock = nb * p.nchunks * NXI * 2048;                  // uniform (the whole pack is below 4 GB)
  auto xi_live = [&](int j) { return XIW * 4 == NXI || g + 4 * j < NXI; };
  auto load_b = [&](int chunk, int j) {
    // (a group without a j-th xi re-reads its first fragment: no conditional load)
    const int xi = xi_live(j) ? g + 4 * j : g;
    bfr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, wblock + (chunk * NXI + xi) * 2048, 0));
  };
  const int a_lane = lh * 128 + li * 4;
  f32x4 af[2];
  auto load_af = [&](int buf, int j) {             // A fragment of xi_j
    const int xi = xi_live(j) ? g + 4 * j : g;
    af[j & 1] = *reinterpret_cast<const f32x4*>(Vs + buf * (NXI * V_XI) + a_lane + xi * V_XI);
  };

  // ---- one chunk as ONE instruction stream: the NM MFMAs of chunk c (V buffer PAR) with the transform of chunk
  // c + 1 (patch buffer and V buffer PAR ^ 1), the requests for chunk c + 2's patches (into patch buffer PAR: chunk c's
  // patches were consumed one chunk ago) and for chunk c + 1's weights SLOTTED between them, a few instructions per
  // MFMA: an MFMA holds the SIMD's vector issue for 8 of its 64 cycles, whatever else the wave issues inside the
  // remaining 56 is hidden, and the partner wave of the SIMD fills the pipe while this one waits for LDS.  (The first
  // form of this kernel ran the transform as a block of its own on one wave of a SIMD beside the MFMAs of the
  // other: the older wave's ~160 VALU instructions win the arbitration in bursts and the matrix pipe idles meanwhile:
  // 0.50 of the MFMA peak where the same MFMAs alone reach 0.85.)
  // MFMA m = 4 j + e: the four k pairs of xi_j in a row (dependent-accumulator latency = issue interval = 64 cycles).
  constexpr int NM = 4 * XIW;
  constexpr int OPS = A == 6 ? 2 : 3;                              // transform outputs per slot
  constexpr int NOUT = NPASS * A, NCS = (NOUT + OPS - 1) / OPS;    // outputs per pass family, slots they take
  constexpr int S_RD1 = 1, S_C1 = S_RD1 + NPASS + 1, S_ST0 = S_C1 + NCS, S_LD1 = S_ST0 + 1, S_RD2 = S_LD1 + 1,
                S_C2 = S_RD2 + NPASS + 1, S_ST1 = S_C2 + NCS + 1;
  static_assert(S_ST1 < NM, "the transform's slots must fit the chunk's MFMAs");
  // slot m: transform of the chunk in patch buffer tb -> V buffer tb; patches of chunk c_raw -> patch buffer tb ^ 1
  auto tslot = [&](auto mc, auto tbc, int c_raw) {
    constexpr int m = decltype(mc)::value;
    constexpr int tb = decltype(tbc)::value;
    if constexpr (m == 0) load_raw_group(c_raw, 0);
    if constexpr (m >= S_RD1 && m < S_RD1 + NPASS) read_pass(tb, m - S_RD1);
    if constexpr (m >= S_C1 && m < S_C1 + NCS) {
#pragma unroll
      for (int o = (m - S_C1) * OPS; o < (m - S_C1 + 1) * OPS; ++o)
        if (o < NOUT) col_out(tb, o / A, o % A);
    }
    if constexpr (m == S_ST0) store_raw_group(tb ^ 1, 0);
    if constexpr (m == S_LD1) load_raw_group(c_raw, 1);
    if constexpr (m >= S_RD2 && m < S_RD2 + NPASS) read_pass(tb, m - S_RD2);
    if constexpr (m >= S_C2 && m < S_C2 + NCS) {
#pragma unroll
      for (int o = (m - S_C2) * OPS; o < (m - S_C2 + 1) * OPS; ++o)
        if (o < NOUT) row_out(tb, o / A, o % A);
    }
    if constexpr (m == S_ST1) store_raw_group(tb ^ 1, 1);
  };
  auto chunk = [&](auto parc, int c_next, int c_raw, auto lastc) {
    constexpr int PAR = decltype(parc)::value;
    constexpr bool LAST = decltype(lastc)::value;
    load_af(PAR, 0);
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int j = m / 4, e = m % 4;
      if constexpr (e == 0 && j + 1 < XIW) load_af(PAR, j + 1);
      if (xi_live(j))
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j & 1][e], bfr[j][e], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!LAST) {
        tslot(mc, std::integral_constant<int, PAR ^ 1>{}, c_raw);
        if constexpr (e == 3) {                    // xi_j's last MFMA is issued: its fragment for the next chunk
          load_b(c_next, j);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  // ---- prologue: chunk 0 transformed, chunk 1's patches in their buffer, chunk 0's weights in flight
#pragma unroll
  for (int j = 0; j < XIW; ++j) load_b(0, j);
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) { load_raw_group(0, grp); store_raw_group(0, grp); }
  static_for<NM>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    if constexpr (m != 0 && m != S_ST0 && m != S_LD1 && m != S_ST1) tslot(mc, P0{}, 0);
  });
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) { load_raw_group(p.nchunks > 1 ? 1 : 0, grp); store_raw_group(1, grp); }
  __syncthreads();

  const int last = p.nchunks - 1;
  // Every chunk runs the same stream — the last one transforms and requests a clamped "next" chunk nobody multiplies:
  // the loop carries no conditional load (conv_igemm.hip on what the wait-count pass makes of those) and no second
  // copy of the body.  Two chunks per trip: every LDS buffer index is a constant.
  for (int c = 0; c <= last; c += 2) {
    chunk(P0{}, c + 1 <= last ? c + 1 : last, c + 2 <= last ? c + 2 : last, std::false_type{});
    __syncthreads();
    if (c + 1 <= last) {
      chunk(P1{}, c + 2 <= last ? c + 2 : last, c + 3 <= last ? c + 3 : last, std::false_type{});
      __syncthreads();
    }
  }

  fused_epilogue<R>(p, acc, smem, tid, g, nh, nb, t0);
}

// ---------------------------------------------------------------------------------------------------------------
// The two-launch form for layers with more than one block of output channels (N > 64).  There the fully fused kernel
// repeats the input transform N / 64 times and its patch fetches (32 useful bytes of every 64-byte sector, waited for
// inside the transform) stall the MFMA stream: 89-98 TFLOP/s where the same MFMAs alone reach 128-144.  Instead
//   wino_input_frag_kernel : x -> V ONCE, stored as the A fragments the product kernel's waves load straight into
//                            registers, Vf[tile block][chunk][xi][k half][tile 32][4] (1 KB per wave instruction)
//   wino_pre_kernel        : the products of 32 tiles x 64 channels x all xi with BOTH operands from global memory —
//                            no LDS and no barrier in the K loop: every wave free-runs (xi_j's fragments are re-loaded
//                            for the next chunk as soon as its four MFMAs are issued) — and the same fused output
//                            transform.  M (2.25x the output tensor, written and read by the three-launch form) and the
//                            output-transform launch are gone; V is written once (2.25x the input) and re-read from
//                            L2 / MALL.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void axpy_fma4(f32x4& acc, bool& first, float coef, const f32x4& v) {
  if (coef == 0.f) return;
  if (first) { acc = (coef == 1.f) ? v : (coef == -1.f) ? -v : coef * v; first = false; }
  else if (coef == 1.f) acc += v;
  else if (coef == -1.f) acc -= v;
  else {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(coef, v[e], acc[e]);
  }
}

// items: ((tile block, group of 32 channels), tile in block, channel quad) — a wave = 8 tiles x 8 quads: whole 128-byte
// lines of x in, 128-byte runs of Vf out
template <int R>
__global__ __launch_bounds__(256) void wino_input_frag_kernel(const FusedP p, float* __restrict__ Vf, long long total) {
  using W = WT<4, R>;
  constexpr int A = 4 + R - 1, NXI = A * A;
  const int ncg = (p.C + 31) >> 5;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int quad = (int)(i & 7), tile = (int)((i >> 3) & 31);
    const long long q = i >> 8;
    const int cg = (int)(q % ncg), tb = (int)(q / ncg);
    const int c = cg * 32 + quad * 4;
    const int tt = tb * FT + tile;
    if (c >= p.C || tt >= p.T) continue;             // (rows of tiles past the end are never stored by the product kernel)
    int b, ty, tx;
    decode_tile(p, tt, b, ty, tx);
    f32x4 d[A][A];
#pragma unroll
    for (int k = 0; k < A; ++k) {
      const int iy = 4 * ty + k;
#pragma unroll
      for (int l = 0; l < A; ++l) {
        const int ix = 4 * tx + l;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy < p.IH && ix < p.IW)
          v = *reinterpret_cast<const f32x4*>(p.x + (((long long)b * p.SH + iy + p.oy) * p.SW + ix + p.ox) * p.ld_x + c);
        d[k][l] = v;
      }
    }
    f32x4 w[A][A];
#pragma unroll
    for (int l = 0; l < A; ++l)
#pragma unroll
      for (int r = 0; r < A; ++r) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy_fma4(acc, first, W::BT[r][k], d[k][l]);
        w[r][l] = acc;
      }
    float* const dst = Vf + ((size_t)tb * p.nchunks + (c >> 3)) * (NXI * 256) + ((c >> 2) & 1) * 128 + tile * 4;
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int qq = 0; qq < A; ++qq) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
#pragma unroll
        for (int l = 0; l < A; ++l) axpy_fma4(acc, first, W::BT[qq][l], w[r][l]);
        *reinterpret_cast<f32x4*>(dst + (r * A + qq) * 256) = acc;
      }
  }
}

template <int R>
__global__ __launch_bounds__(512, 1) void wino_pre_kernel(const FusedP p) {
  using G = FusedGeom<R>;
  constexpr int NXI = G::NXI, XIW = G::XIW;
  __shared__ __attribute__((aligned(16))) float smem[G::SMEM_EPI];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wid & 3, nh = wid >> 2;
  // block order.  0: consecutive blocks = the channel blocks of ONE tile block (they share its Vf slab in L2 and run
  // in step; an XCD streams all the weights).  1: = the tile blocks of ONE channel block (its weight slice stays in L2;
  // Vf is re-read from memory by every channel block).  2: groups of 8 tile blocks x 4 channel blocks = the 32 blocks
  // an XCD runs at a time: both operands of a group come out of L2 4 / 8 times.
  const int v = xcd_remap(blockIdx.x, p.gtb * p.gnb);
  int tb, nb;
  if (p.order == 0) { tb = v / p.gnb; nb = v - tb * p.gnb; }
  else if (p.order == 1) { nb = v / p.gtb; tb = v - nb * p.gtb; }
  else {
    const int grp = v >> 5, r = v & 31, ngn = p.gnb >> 2;
    const int gt = grp / ngn, gn = grp - gt * ngn;
    tb = gt * 8 + (r >> 2); nb = gn * 4 + (r & 3);
  }
  if (tb >= p.ntb || nb >= p.nnb) return;          // (padding of the grouped order; uniform per block)
  const int t0 = tb * FT;
  typedef int i32x4_ __attribute__((ext_vector_type(4)));
  constexpr uint32_t OOB = 0xffffff00u;
  // (the slab of one tile block: nchunks * NXI KB, far below 4 GB; the whole Vf may exceed it)
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.vf) + (size_t)tb * p.nchunks * (NXI * 256), 0, (int)OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wf), 0, (int)OOB, 0x00020000);
  const int alane = lane * 16;
  const int wlane = (nh * 64 + lane) * 16;
  const int wblock = nb * p.nchunks * NXI * 2048;
  auto xi_live = [&](int j) { return XIW * 4 == NXI || g + 4 * j < NXI; };
  f32x16 acc[XIW];
#pragma unroll
  for (int j = 0; j < XIW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f32x4 afr[XIW], bfr[XIW];
  auto load_ab = [&](int chunk, int j) {
    const int xi = xi_live(j) ? g + 4 * j : g;     // (a group without a j-th xi re-reads its first: no conditional load)
    afr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, alane, (chunk * NXI + xi) * 1024, 0));
    bfr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, wblock + (chunk * NXI + xi) * 2048, 0));
  };
#pragma unroll
  for (int j = 0; j < XIW; ++j) load_ab(0, j);
  const int last = p.nchunks - 1;
  for (int c = 0; c < last; ++c) {
#pragma unroll
    for (int j = 0; j < XIW; ++j) {
      if (xi_live(j)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[j][e], bfr[j][e], acc[j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_ab(c + 1, j);                           // xi_j's last MFMA is issued: its fragments of the next chunk
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int j = 0; j < XIW; ++j)
    if (xi_live(j)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[j][e], bfr[j][e], acc[j], 0, 0, 0);
    }
  fused_epilogue<R>(p, acc, smem, tid, g, nh, nb, t0);
}

bool fused_applicable(const clx_conv_desc* d) {
  if (d == nullptr || d->nsrc != 1 || d->KD != 1 || d->ID != 1 || d->PD != 0 || d->PH != 0 || d->PW != 0) return false;
  if (d->KH != d->KW || (d->KH != 3 && d->KH != 2)) return false;
  const clx_src& S = d->src[0];
  if (S.fz != 1 || S.fy != 1 || S.fx != 1 || S.D != 1 || S.oz != 0) return false;
  if (S.C < FK || S.C % FK != 0 || d->N < FN || d->N % FN != 0) return false;
  if (d->IH < d->KH || d->IW < d->KW || d->B <= 0) return false;
  // 32-bit byte offsets into the source, 32-bit tile numbers
  if ((long long)d->B * S.H * S.W * S.ld * 4 >= 0xffff0000ll) return false;
  if ((long long)d->N * S.C * (d->KH == 3 ? 36 : 25) * 4 >= 0xffff0000ll) return false;
  const int OH = d->IH - d->KH + 1, OW = d->IW - d->KW + 1;
  if ((long long)d->B * ((OH + 3) / 4) * ((OW + 3) / 4) >= (1ll << 30)) return false;
  return true;
}

}  // namespace

extern "C" int clx_conv_fused_applicable(const clx_conv_desc* d) { return fused_applicable(d) ? 1 : 0; }

extern "C" size_t clx_conv_fused_workspace_bytes(const clx_conv_desc* d) {
  if (!fused_applicable(d) || d->N <= FN) return 0;
  const int OH = d->IH - d->KH + 1, OW = d->IW - d->KW + 1;
  const long long T = (long long)d->B * ((OH + 3) / 4) * ((OW + 3) / 4);
  const long long ntb = (T + FT - 1) / FT;
  return (size_t)(ntb * FT * d->src[0].C * (d->KH == 3 ? 36 : 25)) * sizeof(float);
}

int clx_wino_fused_fwd(const clx_conv_desc* d, hipStream_t st) {
  CLX_REQUIRE(fused_applicable(d),
              "clx_conv_fwd(winograd fused): needs a 2-D valid 3x3 / 2x2 layer, one plain source with C %% 8 == 0 below "
              "4 GB, N %% 64 == 0 (clx_conv_fused_applicable)");
  CLX_REQUIRE(d->mask == nullptr && d->mask_bits == nullptr && d->vcache == nullptr && !d->adjoint,
              "clx_conv_fwd(winograd fused): no mask / mask_bits / vcache / adjoint form");
  const clx_src& S = d->src[0];
  CLX_REQUIRE(((uintptr_t)S.ptr & 15) == 0 && S.ld % 4 == 0 && ((uintptr_t)d->wpack & 15) == 0,
              "clx_conv_fwd(winograd fused): source and weights must be 16-byte aligned");
  FusedP p = {};
  p.x = S.ptr; p.ld_x = S.ld; p.SH = S.H; p.SW = S.W; p.oy = S.oy; p.ox = S.ox; p.IH = d->IH; p.IW = d->IW;
  p.OH = d->IH - d->KH + 1; p.OW = d->IW - d->KW + 1;
  p.th = (p.OH + 3) / 4; p.tw = (p.OW + 3) / 4;
  const long long Tall = (long long)d->B * p.th * p.tw;
  p.tile_list = d->tile_list;
  if (d->tile_list != nullptr) {
    CLX_REQUIRE(!d->accumulate && d->tile_count >= 0 && d->tile_count <= Tall,
                "clx_conv_fwd(winograd fused): tile_list needs no accumulate and at most th * tw * B tiles");
    if (d->tile_count == 0) return CLX_OK;
    p.T = d->tile_count;
  } else {
    p.T = (int)Tall;
  }
  p.wf = d->wpack; p.C = S.C; p.N = d->N; p.nchunks = S.C / FK;
  p.bias = d->bias; p.relu = d->relu; p.accumulate = d->accumulate;
  p.out = d->out; p.ld_out = d->ld_out;
  CLX_REQUIRE(d->gate_out == nullptr || (d->relu && d->ld_gate * 32 >= d->N),
              "clx_conv_fwd(winograd fused): gate_out needs relu and ld_gate >= N / 32");
  p.gate_out = d->gate_out; p.ld_gate = d->ld_gate;
  if (d->pool_out != nullptr) {
    CLX_REQUIRE(p.OH % 2 == 0 && p.OW % 2 == 0 && d->ld_pool >= d->N && !d->accumulate,
                "clx_conv_fwd(winograd fused): pool_out needs even output height and width, ld_pool >= N, no accumulate");
  }
  p.pool_out = d->pool_out; p.ld_pool = d->ld_pool;
  p.ntb = (p.T + FT - 1) / FT; p.nnb = d->N / FN;
  const long long blocks = (long long)p.ntb * p.nnb;
  CLX_REQUIRE(blocks < (1ll << 31), "clx_conv_fwd(winograd fused): too many blocks");
  const int nxi = d->KH == 3 ? 36 : 25;
  // N > 256: the input transform once, as its own launch, into the workspace (fragment order), and the product kernel
  // that loads both operands straight from global memory; N <= 256 (at most four blocks of channels per tile block repeat
  // the transform) or no workspace: everything in one launch.  CLX_WINO_FUSED_MODE = full | pre forces one form.
  static const char* const mode_env = getenv("CLX_WINO_FUSED_MODE");
  const size_t pre_bytes = (size_t)p.ntb * FT * p.C * nxi * sizeof(float);
  // measured (tools/exp/fused_bench.py, 8 x 526^2 / 262^2 inputs): 256 -> 256 one launch 7.3 ms, two 7.6; 256 -> 768 one
  // 5.3, two 4.6 — four repeats of the transform still cost less than writing and re-reading Vf, twelve do not
  bool pre = p.nnb > 4;
  if (mode_env != nullptr && mode_env[0] == 'f') pre = false;
  if (mode_env != nullptr && mode_env[0] == 'p') pre = true;
  if (d->workspace == nullptr || d->workspace_bytes < pre_bytes || ((uintptr_t)d->workspace & 15) != 0) pre = false;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (pre) {
    float* const Vf = (float*)d->workspace;
    const long long total = (long long)p.ntb * ((p.C + 31) / 32) * 256;
    long long gx = (total + 255) / 256;
    if (gx > 32768) gx = 32768;
    // (the input transform of the two-launch form: an HBM-bound launch, timed with the other transforms)
    if (d->KH == 3) CLX_LAUNCH_KIND(CLX_PROF_WINO_TRANSFORM, (wino_input_frag_kernel<3>), dim3((unsigned)gx), dim3(256), 0, st, p, Vf, total);
    else CLX_LAUNCH_KIND(CLX_PROF_WINO_TRANSFORM, (wino_input_frag_kernel<2>), dim3((unsigned)gx), dim3(256), 0, st, p, Vf, total);
    p.vf = Vf;
    static const int order_env = getenv("CLX_FUSED_ORDER") ? atoi(getenv("CLX_FUSED_ORDER")) : 2;
    p.order = order_env;
    p.gtb = p.ntb; p.gnb = p.nnb;
    if (p.order == 2) { p.gtb = (p.ntb + 7) / 8 * 8; p.gnb = (p.nnb + 3) / 4 * 4; }
    const long long blocks = (long long)p.gtb * p.gnb;
    CLX_REQUIRE(blocks < (1ll << 31), "clx_conv_fwd(winograd fused): too many blocks");
    if (clx_prof_enabled()) clx_prof_events(CLX_PROF_WINO_FUSED, 2.0 * nxi * p.T * (double)p.N * p.C, &e0, &e1);
    if (d->KH == 3) CLX_LAUNCH_TIMED((wino_pre_kernel<3>), dim3((unsigned)blocks), dim3(512), st, e0, e1, p);
    else CLX_LAUNCH_TIMED((wino_pre_kernel<2>), dim3((unsigned)blocks), dim3(512), st, e0, e1, p);
    CLX_CHECK_LAUNCH("clx_conv_fwd(winograd fused, two launches)");
    return CLX_OK;
  }
  // (the one-launch form executes the same products — its FLOPs count; the transforms it repeats per channel block do not)
  if (clx_prof_enabled()) clx_prof_events(CLX_PROF_WINO_FUSED, 2.0 * nxi * p.T * (double)p.N * p.C, &e0, &e1);
  if (d->KH == 3) CLX_LAUNCH_TIMED((wino_fused_kernel<3>), dim3((unsigned)blocks), dim3(512), st, e0, e1, p);
  else CLX_LAUNCH_TIMED((wino_fused_kernel<2>), dim3((unsigned)blocks), dim3(512), st, e0, e1, p);
  CLX_CHECK_LAUNCH("clx_conv_fwd(winograd fused)");
  return CLX_OK;
}
