// f32 MFMA GEMM with the WEIGHTS as the matrix core's A operand (gfx950):
//
//   out[m][n] = act( sum_k x[m][k] * w[n][k] + bias[n] )        computed as out^T = w x^T
//
// Every plain product of the network — the 1x1 convolutions and the per-xi products of the Winograd
// layers (cellulus/models/unet.py:24-63 through funlib's ConvPass) — is a pixel-major activation
// matrix times a small weight matrix.  conv_igemm_kernel stages BOTH operands through LDS and
// transposes its result tile through LDS again to store pixel-major rows.  Transposed, neither is
// needed:
//   * B operand = activations, straight from the registers the global loads fill: lane (pixel i,
//     half h) loads the 16-byte runs k = 8q + 4h .. + 3 of ITS pixel's row;
//   * A operand = weights, the only thing in LDS, stored in fragment order (one conflict-free
//     ds_read_b128 per four v_mfma_f32_32x32x2_f32);
//   * the accumulator of the transposed product holds, per lane, 16-byte channel runs of its own
//     pixel (row n = 8g + 4h + j of lane (i, h), register 4g + j): bias, ReLU, gates and the store
//     happen in registers, pixel-major, without a transpose.
// A wave owns 32 pixels x 32 NT channels (NT = 8: 128 accumulator registers), a block 128 pixels:
// twice the MFMAs per byte staged of the 128x128 kernel, no activation stores to LDS, no epilogue
// through LDS, half the fragment reads.  K is walked in chunks of 32 with the weights of the next
// chunk in flight (global -> registers -> other LDS stage) and the activations of the next chunk in
// a second register set; two blocks per CU cover each other's barrier; persistent blocks carry the
// K pipeline across tiles.
//
// MEASURED AND NOT THE DEFAULT (round 3; CLX_GEMMT=1 selects it, tests/test_gpu_chain.py holds it to
// the same bars).  On the benchmark network it is 3-10 % SLOWER than conv_igemm_kernel<128,128>
// (whole 2-D step 39.9 against 38.3 ms; K = 256: 0.76 against 0.69 ms, K = 768: 1.42 against 1.27 ms).
// Ablations (tools/build_variant.sh, -DGT_EXP bits; tools/bench_conv_fwd.py, TFLOP/s at K = 256 / 768 in
// that harness, conv_igemm 89.5 / 103.9): as it stands 86.7 / 101.9; without the output stores 95.0 / 105.8;
// without operand loads and stores 112.9 / 122.8; also without barriers and fragment reads 117.7 / 128.0.
// What costs is exactly what the design hoped to save: FRAGMENT-SHAPED global accesses — a wave
// instruction that touches 32 rows x 32 bytes keeps the texture addresser busy four times as long as one
// that touches 8 full 128-byte lines (cdna_hip_programming.md, "fragment-shaped x loads"), for the
// activation loads, the weight loads and the 32 stores per lane of the register epilogue alike; issuing
// them quad-coalesced (timing-only variants) did not help either.  Full-line staging through LDS — the
// implicit-GEMM kernel's way — stays the better trade on this part.
#include "clx_common.h"

#include <stdlib.h>

#ifndef GT_EXP
#define GT_EXP 0
#endif

namespace {

__device__ __attribute__((aligned(16))) float g_gt_zero16[4] = {0.f, 0.f, 0.f, 0.f};

struct GemmTP {
  const float* x; long long bs_in; int ld_x;
  const float* w; long long bs_w;
  float* out; long long bs_out; int ld_out;
  int M, N, K;
  const float* bias;
  const float* mask; int ld_mask;
  const unsigned int* mask_bits; int ld_mask_bits;
  unsigned int* gate_out; int ld_gate;
  int relu, accumulate;
  int nbm, nbn, batch;
  const float* zeros;     // 16 zero bytes in global memory: the target of out-of-range loads (no branch, no select)
};

// Persistent blocks: block j walks the tiles j', j' + G, j' + 2G ... of the flattened (batch, m tile, n tile) space
// (j' = XCD-aware remap of j: an XCD owns a contiguous run of every window of G tiles) and the K pipeline never
// drains between them: during a tile's last chunk the operands in flight are the NEXT tile's first chunk, so only
// the epilogue — registers to global memory, nothing to wait for — sits between two tiles' MFMAs.  With one tile
// per block the prologue (two dependent global loads, an LDS round trip, a barrier) and the block turnover cost
// a K = 256 tile a quarter of its time (measured with loads, stores and barriers removed: 114 of 157 TFLOP/s).
template <int NT>
__global__ __launch_bounds__(256, 2) void gemm_t_kernel(const GemmTP p) {
  constexpr int HALF = NT >= 4 ? 4 : NT;             // n tiles per MFMA group (16 MFMAs at HALF = 4)
  constexpr int NHALF = NT / HALF;
  constexpr int WPT = (NT + 3) / 4;                  // n tiles staged per wave and chunk
  constexpr int NG = 4 * NHALF;                      // MFMA groups per chunk
  __shared__ __attribute__((aligned(16))) float Ws[2][NT * 4 * 64 * 4];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int tiles_per_batch = p.nbm * p.nbn;
  const long long total = (long long)tiles_per_batch * p.batch;
  const int G = gridDim.x;
  const int nchunks = (p.K + 31) >> 5;

  // everything a tile's loads and its epilogue need
  struct Tile { const float* xrow; const float* wbase; int n0, row, b; bool ok; };
  auto tile_at = [&](long long f) {
    Tile t;
    t.b = (int)(f / tiles_per_batch);
    const int v = (int)(f - (long long)t.b * tiles_per_batch);
    const int tile_n = v % p.nbn, tile_m = v / p.nbn;
    t.n0 = tile_n * (32 * NT);
    t.row = tile_m * 128 + wid * 32 + i;
    t.ok = t.row < p.M;
    t.xrow = p.x + t.b * p.bs_in + (size_t)(t.ok ? t.row : p.M - 1) * p.ld_x + 4 * h;
    t.wbase = p.w + t.b * p.bs_w;
    return t;
  };

  f32x4 xb[4], xn[4], wr[4];          // wr: ONE staged n tile at a time (two per chunk at NT = 8: 16 registers, not 32)
  auto load_x = [&](const Tile& t, int c, f32x4* dst) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 32 * c + 8 * q + 4 * h;
      dst[q] = *reinterpret_cast<const f32x4*>(k < p.K ? t.xrow + 32 * c + 8 * q : p.zeros);
    }
  };
  auto load_w = [&](const Tile& t, int c, int u) {
    const int tt = wid + 4 * u;
    const int n = t.n0 + 32 * tt + i;
    const bool nv = (NT % 4 == 0 || tt < NT) && n < p.N;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 32 * c + 8 * q + 4 * h;
      const bool live = nv && k < p.K;
      wr[q] = *reinterpret_cast<const f32x4*>(live ? t.wbase + (size_t)n * p.K + k : p.zeros);
    }
  };
  auto store_w = [&](int stage, int u) {
    const int tt = wid + 4 * u;
    if (NT % 4 == 0 || tt < NT) {        // (NT a multiple of 4: every wave stages WPT whole tiles, no predicate)
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&Ws[stage][((tt * 4 + q) * 64 + lane) << 2]) = wr[q];
    }
  };

  f32x16 acc[NT];
  f32x4 af[2][HALF];
  auto load_frags = [&](int stage, int g, int slot) {
    const int q = g / NHALF, hf = g % NHALF;
    if ((GT_EXP & 16) && g > 0) return;          // experiment: MFMAs without fragment reads
#pragma unroll
    for (int t = 0; t < HALF; ++t)
      af[slot][t] = *reinterpret_cast<const f32x4*>(&Ws[stage][(((hf * HALF + t) * 4 + q) * 64 + lane) << 2]);
  };
  auto mfma_group = [&](int g, int slot) {
    const int q = g / NHALF, hf = g % NHALF;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < HALF; ++t)
        acc[hf * HALF + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][t][j], xb[q][j], acc[hf * HALF + t], 0, 0, 0);
  };
  // one K chunk of the current tile; the operands loaded meanwhile are chunk `nc` of tile `nt` (no conditionals
  // around loads and stores: see conv_igemm.hip on the wait-count pass)
  int stage = 0;
  auto chunk = [&](const Tile& nt, int nc) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g == 0 && !(GT_EXP & 1)) load_x(nt, nc, xn);
      // weights of the next chunk: one n tile per wave in flight at a time — loaded two groups before its LDS store
      if (GT_EXP & 2) {
      } else if (WPT == 2) {
        if (g == 1) load_w(nt, nc, 0);
        if (g == 3) store_w(stage ^ 1, 0);
        if (g == 4) load_w(nt, nc, 1);
        if (g == NG - 1) store_w(stage ^ 1, 1);
      } else {
        if (g == (NG > 1 ? 1 : 0)) load_w(nt, nc, 0);
        if (g == NG - 1) store_w(stage ^ 1, 0);
      }
      if (g + 1 < NG) load_frags(stage, g + 1, (g + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(g, g & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(GT_EXP & 8)) __syncthreads();
    stage ^= 1;
    if (!(GT_EXP & 1)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) xb[q] = xn[q];
    }
    load_frags(stage, 0, 0);
  };

  long long f = xcd_remap(blockIdx.x, G);
  if (f >= total) return;
  Tile cur = tile_at(f);
  load_x(cur, 0, xb);
#pragma unroll
  for (int u = 0; u < WPT; ++u) { load_w(cur, 0, u); store_w(0, u); }
  __syncthreads();
  load_frags(0, 0, 0);

  for (; f < total; f += G) {
    const Tile nxt = tile_at(f + G < total ? f + G : f);      // (last tile: a harmless reload of its own first chunk)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int c = 0; c + 1 < nchunks; ++c) chunk(cur, c + 1);
    chunk(nxt, 0);

    // ---- epilogue in registers: acc[t][4g + j] = channel n0 + 32 t + 8 g + 4 h + j of pixel `row`
    const int row = cur.row, n0 = cur.n0;
    const bool ok = cur.ok;
    float* orow = p.out + cur.b * p.bs_out + (size_t)row * p.ld_out;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int nt0 = n0 + 32 * t;
      if (nt0 >= p.N) break;
      unsigned int mword = 0xffffffffu;
      if (p.mask_bits != nullptr && ok) mword = p.mask_bits[(size_t)row * p.ld_mask_bits + (nt0 >> 5)];
      unsigned int gbits = 0u;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 8 * g + 4 * h;                     // channel inside the 32-wide tile
        const int n = nt0 + c;
        f32x4 val;
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] = acc[t][4 * g + j];
        const bool full = n + 3 < p.N;
        if (p.bias != nullptr) {
#pragma unroll
          for (int j = 0; j < 4; ++j) val[j] += (n + j < p.N) ? p.bias[n + j] : 0.f;
        }
        if (p.accumulate && ok && n < p.N) {             // out = act(conv + bias + out); n + 3 < ld_out (ld_out % 4 == 0)
          const f32x4 prev = *reinterpret_cast<const f32x4*>(orow + n);
          val += prev;
        }
        if (p.relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) val[j] = fmaxf(val[j], 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          val[j] = ((mword >> (c + j)) & 1u) ? val[j] : 0.f;
          gbits |= (n + j < p.N && val[j] > 0.f ? 1u : 0u) << (c + j);
        }
        if (!ok || n >= p.N || (GT_EXP & 4)) continue;
        if (full) {
          if (p.mask != nullptr) {
            const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + (size_t)row * p.ld_mask + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = (mk[j] > 0.f) ? val[j] : 0.f;
          }
          *reinterpret_cast<f32x4*>(orow + n) = val;
        } else {
          for (int j = 0; j < 4 && n + j < p.N; ++j) {
            float xv = val[j];
            if (p.mask != nullptr) xv = (p.mask[(size_t)row * p.ld_mask + n + j] > 0.f) ? xv : 0.f;
            orow[n + j] = xv;
          }
        }
      }
      if (p.gate_out != nullptr) {                       // the two half-lanes of a pixel hold complementary nibbles
        const unsigned int word = gbits | (unsigned int)__shfl_xor((int)gbits, 32, 64);
        if (ok && h == (t & 1) && nt0 < p.ld_out) p.gate_out[(size_t)row * p.ld_gate + (nt0 >> 5)] = word;
      }
    }
    cur = nxt;
  }
}

}  // namespace

static const float* gt_zero_buffer() {
  static const float* cache[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (cache[dev] == nullptr) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_gt_zero16)) != hipSuccess) return nullptr;
    cache[dev] = (const float*)ptr;
  }
  return cache[dev];
}

// plain products only: one source read as it lies, one tap, no padding
bool clx_gemmt_applicable(const clx_conv_desc* d) {
  // OFF by default: measured 3-10 % slower than conv_igemm_kernel<128,128> on every layer of the benchmark
  // network (header comment); CLX_GEMMT=1 selects it
  static const bool enabled = getenv("CLX_GEMMT") != nullptr && atoi(getenv("CLX_GEMMT")) != 0;
  if (!enabled || d->nsrc != 1) return false;
  if (d->KD != 1 || d->KH != 1 || d->KW != 1 || d->PD || d->PH || d->PW) return false;
  const clx_src& S = d->src[0];
  if (S.fz != 1 || S.fy != 1 || S.fx != 1 || S.oz || S.oy || S.ox) return false;
  if (S.D != d->ID || S.H != d->IH || S.W != d->IW) return false;
  return d->N > 64;           // (narrow layers: the 128x64 kernel, three blocks per CU)
}

int clx_gemmt_launch(const clx_conv_desc* d, int batch, long long bs_in, long long bs_w, long long bs_out,
                     hipStream_t st) {
  const clx_src& S = d->src[0];
  GemmTP p;
  p.x = S.ptr; p.bs_in = bs_in; p.ld_x = S.ld;
  p.w = d->wpack; p.bs_w = bs_w;
  p.out = d->out; p.bs_out = bs_out; p.ld_out = d->ld_out;
  p.M = d->B * d->ID * d->IH * d->IW; p.N = d->N; p.K = S.C;
  p.bias = d->bias; p.mask = d->mask; p.ld_mask = d->ld_mask;
  p.mask_bits = d->mask_bits; p.ld_mask_bits = d->ld_mask_bits;
  p.gate_out = d->gate_out; p.ld_gate = d->ld_gate;
  p.relu = d->relu; p.accumulate = d->accumulate;
  p.nbm = cdiv(p.M, 128);
  p.zeros = gt_zero_buffer();
  CLX_REQUIRE(p.zeros != nullptr, "clx_conv_fwd: cannot resolve the device zero buffer");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (clx_prof_enabled()) clx_prof_events(CLX_PROF_GEMM_T, 2.0 * p.M * p.N * p.K * batch, &e0, &e1);
  p.batch = batch;
  // block columns of 256 channels unless the last one would be mostly padding: 128-wide columns then
  const int rem = p.N % 256;
  const bool wide = rem == 0 || rem > 128;
  p.nbn = cdiv(p.N, wide ? 256 : 128);
  const void* fn = wide ? (const void*)gemm_t_kernel<8> : (const void*)gemm_t_kernel<4>;
  static int slots[2] = {0, 0};
  int& sl = slots[wide ? 0 : 1];
  if (sl == 0) {
    int dev = 0, cus = 0, per_cu = 0;
    CLX_REQUIRE(hipGetDevice(&dev) == hipSuccess &&
                    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) == hipSuccess && per_cu > 0,
                "clx_conv_fwd: occupancy query failed");
    sl = cus * per_cu;
  }
  const long long total = (long long)p.nbm * p.nbn * batch;
  const int grid = (int)(total < sl ? total : sl);
  if (wide)
    CLX_LAUNCH_TIMED((gemm_t_kernel<8>), dim3(grid), dim3(256), st, e0, e1, p);
  else
    CLX_LAUNCH_TIMED((gemm_t_kernel<4>), dim3(grid), dim3(256), st, e0, e1, p);
  return CLX_OK;
}
