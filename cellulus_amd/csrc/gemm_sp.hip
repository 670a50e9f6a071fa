// Split-precision products: float32 GEMMs on the bf16 matrix cores, fed from PRE-SPLIT operand planes.
//
// gfx950 has no TF32 and its f32 MFMA peak (157 TFLOP/s) is 1/16 of the bf16 one.  Every f32 operand element is split
// EXACTLY into three bf16 pieces x = h0 + h1 + h2 (8 + 8 + 8 significand bits, each piece rounded to nearest) and the six products
// a_i b_j with i + j <= 2 are accumulated in the f32 accumulators of v_mfma_f32_32x32x16_bf16: each product is exact, the
// dropped terms (a1 b2, a2 b1, a2 b2) are <= 2^-24 relative — the size of one f32 rounding.  Peak of this form: bf16 MFMA /
// 6 = 419 TFLOP/s f32-equivalent.
//
// Rounds 2-4 split the operands INSIDE the GEMM (gemm_x3.hip, removed in round 5): ~22 VALU instructions per four
// elements on the issue port the MFMAs use, register staging, two barriers per 32-deep chunk — 168 TFLOP/s.  Here the
// split happens ONCE where a tensor is produced (clx_split_planes, the Winograd transforms, the weight packing), into
// the "P3" plane format below, and the K loop is nothing but LDS-DMA (global_load_lds_dwordx4), ds_read_b128 and MFMAs.
//
// P3 format of an [R rows][K] operand (K % 16 == 0; rows padded to a multiple of 64 and at least 128, the padding rows ZERO):
//   1-KB fragments in the MFMA's operand order — fragment (rb, ks, p) = plane p of rows 32 rb .. + 31, k = 16 ks .. + 15,
//   at byte ((rb * K/16 + ks) * 3 + p) * 1024; inside it lane l = 32 h + r of the wavefront owns the 16 bytes
//   x_p[32 rb + r][16 ks + 8 h .. + 7].  One global_load_lds_dwordx4 per fragment moves 1 KB of CONTIGUOUS memory
//   into LDS in exactly the order ds_read_b128 hands it to the matrix core (lane-linear: no bank conflicts, no swizzle).
//   6 bytes per element.
//
// gemm_sp_kernel: out[m][n] = epilogue( sum_k A[m][k] B[n][k] ), 256 x 128 tile, 512 threads = 8 waves (4 x 2) of
// 64 x 64, K walked in 16-deep steps through a ring of FOUR 36-KB stages (three steps in flight, one barrier per step,
// counted vmcnt; operand fragments double-buffered in registers).  Two-level summation as in conv_igemm.hip — fresh accumulators every 64 products, added into a second
// set — with the sign of A alternating between periods: the matrix core adds the 16 products of an instruction to the
// accumulator with a floor-like truncation (a bias of ~1e-7 of the output's rms with the same sign everywhere, which
// is common-mode over the 5e5 pixels a weight gradient sums over; docs/HISTORY.md 6b), and -(A B) carries the same
// expected bias as +(A B), so the difference of the two kinds of period has none.
//
// Three forward / data-gradient kernels share the format, the arithmetic and the epilogue: gemm_sp_kernel<0> (this one: K >
// 1024), gemm_sp2_kernel (128 x 128 tiles, two workgroups per CU: the default up to K = 1024, i.e. every product of the
// benchmark networks) and gemm_sp16_kernel (v_mfma_f32_16x16x32_bf16, opt-in); gemm_sp_kernel<1> is the weight gradient.
//
// Replaces nn.Conv{2,3}d 1x1 (+ReLU) and the transform-domain products of the 3x3 layers, forward and data gradient
// (cellulus/models/unet.py:24-63, cellulus/train.py:178) when clx_conv_desc.precision = CLX_PREC_F32X3BF16.
#include "clx_common.h"
#include "sp_planes.h"
#include <stdlib.h>
#include <type_traits>

// timing-only ablations (tools/build_variant.sh sp<k> gemm_sp.hip -DSP_ABL=<k>; results are wrong; 8 = the products as 16x16x32 MFMAs): 1 = no loads in the K
// loop, 2 = no barriers / waits, 3 = fragments read once, 4 = no epilogue
#ifndef SP_ABL
#define SP_ABL 0
#endif

// shader clock under load (as conv_igemm.hip): the middle block of every launch adds the shader-clock and 100-MHz wall-clock
// ticks of its own life to two counters; clx_profile_clock (conv_igemm.hip) adds them to its own
__device__ unsigned long long g_sp_clk_ticks[2];
int clx_sp_clock_read(double* shader_ticks, double* wall_ticks, int reset) {
  unsigned long long h[2] = {0ull, 0ull};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sp_clk_ticks), sizeof(h)) != hipSuccess) return CLX_ERR_LAUNCH;
  *shader_ticks = (double)h[0];
  *wall_ticks = (double)h[1];
  if (reset) {
    const unsigned long long z[2] = {0ull, 0ull};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_sp_clk_ticks), z, sizeof(z)) != hipSuccess) return CLX_ERR_LAUNCH;
  }
  return CLX_OK;
}

#ifdef SP16_STAMP
// diagnostic build (tools/build_variant.sh stamp gemm_sp.hip -DSP16_STAMP): s_memtime stamps of waves 0 and 4 of block 0 around the
// segments of double steps 16 .. 19 of gemm_sp16_kernel; tools/exp/sp16_stamps.py prints them
__device__ unsigned long long g_sp16_stamps[2][64];
extern "C" int clx_sp16_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sp16_stamps), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : 1;
}
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
using sp::u32x2;
using sp::u32x4;
using sp::FRAG;
using sp::KSTEP;
constexpr int SP_BM = 256, SP_BN = 128;
constexpr int A_FRAGS = SP_BM / 32 * 3;    // 24 fragments of A per stage
constexpr int B_FRAGS = SP_BN / 32 * 3;    // 12 of B
constexpr int STAGE = (A_FRAGS + B_FRAGS) * FRAG;   // 36 KB
constexpr int RING = 4;
constexpr int LDC = SP_BN + 4;

__device__ __forceinline__ void glds16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

using sp::split4;

// f32 [rows][ld] -> P3 planes of its columns [0, K).  A wavefront writes whole fragments (three contiguous 1-KB
// stores); the padding rows are written as zeros.  COLSUM: also colsum[k] += sum over rows of x[row][k]
// (the bias gradient of a layer whose dY is split here): the grid's wavefront count is a multiple of the k steps, so a
// wavefront keeps its k step and sums in registers; one shuffle reduction and 16 atomics per wavefront at the end.
template <bool COLSUM>
__global__ __launch_bounds__(256) void sp_split_kernel(const float* __restrict__ x, long long ld, long long rows, int ksteps,
                                                       char* __restrict__ out, long long nfrag, float* __restrict__ colsum, int nreal) {
  const int lane = threadIdx.x & 63;
  const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
  const int r = lane & 31, h = lane >> 5;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  for (long long f = w0; f < nfrag; f += nw) {
    const int ks = (int)(f % ksteps);
    const long long rb = f / ksteps;
    const long long row = rb * 32 + r;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (row < rows) {
      const float* src = x + row * ld + ks * 16 + h * 8;
      v0 = *reinterpret_cast<const f32x4*>(src);
      v1 = *reinterpret_cast<const f32x4*>(src + 4);
    }
    if constexpr (COLSUM) { s0 += v0; s1 += v1; }
    u32x2 a0, a1, a2, b0, b1, b2;
    split4(v0, a0, a1, a2);
    split4(v1, b0, b1, b2);
    char* dst = out + f * KSTEP + lane * 16;
    *reinterpret_cast<u32x4*>(dst) = u32x4{a0[0], a0[1], b0[0], b0[1]};
    *reinterpret_cast<u32x4*>(dst + FRAG) = u32x4{a1[0], a1[1], b1[0], b1[1]};
    *reinterpret_cast<u32x4*>(dst + 2 * FRAG) = u32x4{a2[0], a2[1], b2[0], b2[1]};
  }
  if constexpr (COLSUM) {
    if (w0 < nfrag) {
      const int ks = (int)(w0 % ksteps);                 // (nw % ksteps == 0: the same k step every trip)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) { s0[e] += __shfl_xor(s0[e], m, 64); s1[e] += __shfl_xor(s1[e], m, 64); }
      }
      if (r == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k0 = ks * 16 + h * 8 + e;
          if (k0 < nreal) atomicAdd(colsum + k0, s0[e]);
          if (k0 + 4 < nreal) atomicAdd(colsum + k0 + 4, s1[e]);
        }
      }
    }
  }
}

// P3 planes -> f32 [rows][ld] (x = h0 + h1 + h2, exact): the inverse of sp_split_kernel, for tests and diagnostics
__global__ __launch_bounds__(256) void sp_join_kernel(const char* __restrict__ in, long long ld, long long rows, int ksteps,
                                                      float* __restrict__ x, long long nfrag) {
  const int lane = threadIdx.x & 63;
  const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
  const int r = lane & 31, h = lane >> 5;
  for (long long f = w0; f < nfrag; f += nw) {
    const int ks = (int)(f % ksteps);
    const long long rb = f / ksteps;
    const long long row = rb * 32 + r;
    if (row >= rows) continue;
    const char* src = in + f * KSTEP + lane * 16;
    const u32x4 q0 = *reinterpret_cast<const u32x4*>(src), q1 = *reinterpret_cast<const u32x4*>(src + FRAG),
                q2 = *reinterpret_cast<const u32x4*>(src + 2 * FRAG);
    float* dst = x + row * ld + ks * 16 + h * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int sh = (e & 1) * 16;
      const float f0 = __uint_as_float(((q0[e >> 1] >> sh) & 0xffffu) << 16);
      const float f1 = __uint_as_float(((q1[e >> 1] >> sh) & 0xffffu) << 16);
      const float f2 = __uint_as_float(((q2[e >> 1] >> sh) & 0xffffu) << 16);
      dst[e] = (f0 + f1) + f2;
    }
  }
}

// the padding rows [rows, padded_rows(rows)) of `batch` plane sets
__global__ __launch_bounds__(256) void sp_zero_tail_kernel(char* __restrict__ planes, long long bs, long long rows, int ksteps) {
  const long long rb0 = rows >> 5, rb1 = sp::padded_rows(rows) / 32;    // row blocks that hold padding
  const int per_rb = ksteps * 6 * 32;                                   // 16-byte pieces of one row block
  const long long total = (rb1 - rb0) * per_rb;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long rb = rb0 + i / per_rb;
    const int k = (int)(i % per_rb), r = k & 31, q = k >> 5;            // q = (k step, piece, half)
    if (rb * 32 + r >= rows)
      *reinterpret_cast<u32x4*>(planes + blockIdx.y * bs + rb * ksteps * (long long)KSTEP + (long long)q * 512 + r * 16) = u32x4{0u, 0u, 0u, 0u};
  }
}

struct SpP {
  const char* A;                      // forward: P3 planes of the [M][K] operand (rows = output pixels).  Weight gradient: of dY [pixels][N]
  const char* B;                      // forward: planes of the [N][K] operand (rows = output channels).  Weight gradient: of x [pixels][C]
  long long bs_a, bs_b, bs_out;       // batch strides (problems of one launch): bytes, bytes, floats
  float* out;
  int M, N, ksteps;                   // forward: ksteps = K / 16, a multiple of 4.  Weight gradient: M = N of dY (rows of dW), N = C
  int rb_a;                           // 32-row blocks A holds (row blocks past it are clamped: their results are never stored)
  const float* bias;
  const float* mask;
  const unsigned int* mask_bits;
  unsigned int* gate_out;
  const float* zeros;
  int relu, accumulate, ld_out, ld_mask, ld_mask_bits, ld_gate;
  int nbm, nbn;
  char* out_planes;                   // forward, one problem per launch: ALSO the P3 planes of out ([M][N]), or NULL
  float* colsum; int colsum_n;        // forward: colsum[n] += sum_m out[m][n], n < colsum_n, or NULL
  // weight gradient: the contraction runs over pixel steps of 16; a block takes `steps_per_slice` of them
  unsigned int stride_a, stride_b;    // bytes of one 32-pixel row block of the dY / x planes
  int total_steps, steps_per_slice, nslices;
};

// The tile's epilogue out of LDS (Cs: [SP_BM][LDC] floats, bias added, written and barrier-synchronised by the caller):
// previous output, ReLU, gates, masks, the float32 store, the output's own planes, its column sums.
template <int BM = SP_BM, int NWAVES = 8>
__device__ __forceinline__ void sp_epilogue_from_lds(const SpP& p, const float* Cs, int m0, int n0, int batch, int tid, int lane) {
  // A wavefront takes 8 rows x 32 channels per pass (the eight lanes of a row hold one gate word; the float32 stores are
  // 128-byte runs, and so are the stores of every piece of the output's own planes: sp_planes.h); the 8 (4) waves of a pass
  // cover 16 (8) rows x 128 channels, 16 passes the tile.
  constexpr int PASS_ROWS = NWAVES / 4 * 8;
  constexpr int ITERS = BM / PASS_ROWS;                 // 16
  constexpr int PH = 8;
  const int wv = tid >> 6;
  const int c4 = (wv & 3) * 32 + (lane & 7) * 4;
  const int row0 = (wv >> 2) * 8 + (lane >> 3);
  const int n = n0 + c4;
  const bool n_live = n < p.N;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  const long long prow = p.out_planes ? sp::padded_rows(p.M) : 0;
#pragma unroll 1
  for (int h0 = 0; h0 < ITERS; h0 += PH) {
    f32x4 val[PH];
    int mrow[PH];
    bool live[PH];
#pragma unroll
    for (int j = 0; j < PH; ++j) {
      const int row = row0 + PASS_ROWS * (h0 + j);
      mrow[j] = m0 + row;
      live[j] = mrow[j] < p.M && n_live;
      val[j] = *reinterpret_cast<const f32x4*>(&Cs[row * LDC + c4]);
    }
    float* const out_base = p.out + batch * p.bs_out + n;
    if (p.accumulate) {
      f32x4 prev[PH];
#pragma unroll
      for (int j = 0; j < PH; ++j)
        prev[j] = *reinterpret_cast<const f32x4*>(live[j] ? out_base + (size_t)mrow[j] * p.ld_out : p.zeros);
#pragma unroll
      for (int j = 0; j < PH; ++j) val[j] += prev[j];
    }
    if (p.relu) {
#pragma unroll
      for (int j = 0; j < PH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) val[j][e] = fmaxf(val[j][e], 0.f);
    }
    if (p.mask_bits) {
      unsigned int wd[PH];
#pragma unroll
      for (int j = 0; j < PH; ++j) wd[j] = live[j] ? p.mask_bits[(size_t)mrow[j] * p.ld_mask_bits + (n >> 5)] : 0u;
#pragma unroll
      for (int j = 0; j < PH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) val[j][e] = ((wd[j] >> ((n & 31) + e)) & 1u) ? val[j][e] : 0.f;
    }
    if (p.gate_out) {
#pragma unroll
      for (int j = 0; j < PH; ++j) {
        unsigned int nib = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) nib |= (live[j] && val[j][e] > 0.f) ? (1u << e) : 0u;
        unsigned int word = nib << (4 * (lane & 7));
        word |= __shfl_xor(word, 1, 64);
        word |= __shfl_xor(word, 2, 64);
        word |= __shfl_xor(word, 4, 64);
        if ((lane & 7) == 0 && mrow[j] < p.M && n < p.ld_out) p.gate_out[(size_t)mrow[j] * p.ld_gate + (n >> 5)] = word;
      }
    }
    if (p.mask) {
      f32x4 mk[PH];
#pragma unroll
      for (int j = 0; j < PH; ++j)
        mk[j] = *reinterpret_cast<const f32x4*>(live[j] ? p.mask + (size_t)mrow[j] * p.ld_mask + n : p.zeros);
#pragma unroll
      for (int j = 0; j < PH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) val[j][e] = (!live[j] || mk[j][e] > 0.f) ? val[j][e] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < PH; ++j)
      if (live[j]) *reinterpret_cast<f32x4*>(out_base + (size_t)mrow[j] * p.ld_out) = val[j];      // N % 128 == 0: whole groups
    if (p.out_planes) {
      // the result as the next product's operand: its three pieces, split here instead of by a pass of its own; the
      // padding rows of the planes (all inside the last tile of rows) as zeros
#pragma unroll
      for (int j = 0; j < PH; ++j)
        if (n_live && mrow[j] < prow)
          sp::store4(p.out_planes, mrow[j], n, p.N >> 4, live[j] ? val[j] : f32x4{0.f, 0.f, 0.f, 0.f});
    }
    if (p.colsum) {
#pragma unroll
      for (int j = 0; j < PH; ++j)
        if (live[j]) csum += val[j];
    }
  }
  if (p.colsum) {
    // column sums of the stored tile (the bias gradient of the layer whose dY this launch produces): over the 8 rows of
    // the wavefront by shuffles, then one float atomic per channel and wavefront
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      csum[e] += __shfl_xor(csum[e], 8, 64);
      csum[e] += __shfl_xor(csum[e], 16, 64);
      csum[e] += __shfl_xor(csum[e], 32, 64);
    }
    if ((lane >> 3) == 0 && n_live) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < p.colsum_n) atomicAdd(p.colsum + n + e, csum[e]);
    }
  }
}

// MODE 0: out[m][n] = epilogue( sum_k A[m][k] B[n][k] )             (forward / data gradient)
// MODE 1: out[n][c] += sum_pixels dY[pixel][n] x[pixel][c]            (weight gradient; float atomics, split over pixel slices)
// Both operands of MODE 1 are pixel-major, and the MFMA wants eight consecutive k (= pixels) of one channel per lane: the
// LDS image of a 16-pixel step is [pixel][32 channels] per 32-channel block and piece (1 KB: one LDS-DMA instruction whose
// lanes gather 16-byte pieces — 256-byte runs of memory), and a fragment is two ds_read_b64_tr_b16, the transposing LDS read
// of gfx950: a 16-lane group reads 4 pixels x 16 channels and every lane receives the 4 pixels of ITS channel.  4 pixels x
// 64 bytes = one 256-byte bank row per 32-lane half: conflict-free.
template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm_sp_kernel(const SpP p) {
  __shared__ __attribute__((aligned(16))) char smem[RING * STAGE];       // 144 KB; the C tile (132 KB) in the epilogue
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  int tile_m, tile_n, batch, step0 = 0, ksteps;
  if constexpr (MODE == 0) {
    const int v = xcd_remap(blockIdx.x, p.nbm * p.nbn);
    tile_n = v % p.nbn; tile_m = v / p.nbn;
    batch = blockIdx.y;
    ksteps = p.ksteps;
  } else {
    const int T = p.nbm * p.nbn;
    const int u = xcd_remap(blockIdx.x, gridDim.x);
    const int per_batch = T * p.nslices;
    batch = u / per_batch;
    const int v = u % per_batch;
    const int slice = v / T, t = v % T;
    tile_n = t % p.nbn; tile_m = t / p.nbn;
    step0 = slice * p.steps_per_slice;
    ksteps = p.total_steps - step0 < p.steps_per_slice ? p.total_steps - step0 : p.steps_per_slice;
  }
  const int m0 = tile_m * SP_BM, n0 = tile_n * SP_BN;
  const bool clk_block = blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && threadIdx.x == 0;   // mid-launch
  unsigned long long clk_c0 = 0, clk_w0 = 0;
  if (clk_block) { clk_c0 = clock64(); clk_w0 = wall_clock64(); }

  // ---- this wave's share of a stage's 36 fragments: f = w + 8 q, q = 0..3, and a fifth one, 32 + (w & 3), for waves 0-3
  // in even steps and waves 4-7 in odd steps: any two consecutive steps are NINE loads for every wave, so the counted
  // wait in front of a step is the same instruction for all of them.  A fragment's address is wave-uniform up to
  // a lane term: scalar bases, one 32-bit vector offset per operand.
  const char* const Ab = p.A + batch * p.bs_a;
  const char* const Bb = p.B + batch * p.bs_b;
  auto frag_base = [&](int f) __attribute__((always_inline)) -> const char* {
    if constexpr (MODE == 0) {
      if (f < A_FRAGS) {
        int rb = tile_m * (SP_BM / 32) + f / 3;
        if (rb > p.rb_a - 1) rb = p.rb_a - 1;
        return Ab + ((long long)rb * ksteps * 3 + f % 3) * FRAG;
      }
      const int g = f - A_FRAGS;
      const int nb = tile_n * (SP_BN / 32) + g / 3;
      return Bb + ((long long)nb * ksteps * 3 + g % 3) * FRAG;
    } else {
      // 32-channel block cb of an operand = its k steps 2 cb, 2 cb + 1 (both halves): the lane term picks the octet
      if (f < A_FRAGS) {
        int cb = tile_m * (SP_BM / 32) + f / 3;
        if (cb > p.M / 32 - 1) cb = p.M / 32 - 1;          // (rows of dW past N: computed, never added)
        return Ab + ((long long)(2 * cb) * 3 + f % 3) * FRAG;
      }
      const int g = f - A_FRAGS;
      const int cb = tile_n * (SP_BN / 32) + g / 3;
      return Bb + ((long long)(2 * cb) * 3 + g % 3) * FRAG;
    }
  };
  const char* gsrc[5];
#pragma unroll
  for (int q = 0; q < 4; ++q) gsrc[q] = frag_base(w + 8 * q);
  gsrc[4] = frag_base(32 + (w & 3));
  // MODE 0: lane l owns the 16 bytes at 16 l of its fragment.  MODE 1: lane l fetches the piece (octet o = l >> 4 of the
  // 32-channel block, pixel (l & 15) ^ 4 o): sixteen consecutive lanes read 256 consecutive bytes (consecutive lanes on
  // different octets — four 16-byte requests per lane quad — cost the kernel a third of its time), and the XOR spreads the four
  // octets' 4-pixel groups over the LDS banks for the transposing reads.  Octet o = k step o >> 1, half o & 1 of the row block.
  const int l_oct = lane >> 4, l_pix = (lane & 15) ^ (4 * (lane >> 4));
  const unsigned int lane16 = MODE == 0 ? (unsigned int)lane * 16u
                                        : (unsigned int)((l_oct >> 1) * KSTEP + (l_oct & 1) * 512 + l_pix * 16);
  const bool low_half = w < 4;
  auto issue = [&](int t, int slot, bool odd) __attribute__((always_inline)) {
    if (SP_ABL == 1 && t > 2) return;
    char* const dst = smem + slot * STAGE;
    if constexpr (MODE == 0) {
      const unsigned int voff = lane16 + (unsigned int)t * KSTEP;        // (K * 192 bytes per row block: far below 4 GB)
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(gsrc[q] + (size_t)voff, dst + (w + 8 * q) * FRAG);
      if (low_half != odd) glds16(gsrc[4] + (size_t)voff, dst + (32 + (w & 3)) * FRAG);
    } else {
      const unsigned int ts = (unsigned int)(step0 + t);
      const unsigned int va = lane16 + (ts >> 1) * p.stride_a + (ts & 1u) * 256u;
      const unsigned int vb = lane16 + (ts >> 1) * p.stride_b + (ts & 1u) * 256u;
      // fragments w, w + 8, w + 16 are dY's (f < 24), w + 24 and the fifth (32 + ...) are x's
#pragma unroll
      for (int q = 0; q < 3; ++q) glds16(gsrc[q] + (size_t)va, dst + (w + 8 * q) * FRAG);
      glds16(gsrc[3] + (size_t)vb, dst + (w + 24) * FRAG);
      if (low_half != odd) glds16(gsrc[4] + (size_t)vb, dst + (32 + (w & 3)) * FRAG);
    }
  };
  // own loads of the step about to be multiplied have landed when at most the two steps behind it are in flight
  auto wait_two = [&]() __attribute__((always_inline)) { if (SP_ABL != 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); };
  auto wait_one = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); };

  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

  const int li = lane & 31, lh = lane >> 5;
  float bias_v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + (wn * 2 + j) * 32 + li;
    bias_v[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
  }

  // LDS addresses of this wave's fragments per ring slot: eight registers, made opaque so that the compiler keeps exactly
  // these and reaches the pieces by instruction offsets (left to itself it materialised — and spilled — one address per
  // read: the slots lie beyond the 64-KB offset field)
  unsigned int la[RING], lb[RING];
#pragma unroll
  for (int k = 0; k < RING; ++k) {
    // MODE 1: the transposing read — lane 4 q + p of a 16-lane group addresses pixel q, channels 4 p .. 4 p + 3 of the group's
    // 16 channels; group g = lane >> 4 holds channels 16 (g & 1) .. and the k half g >> 1 (pixels P = 8 (g >> 1) + q, + 4 for the
    // second read: address ^ 64).  LDS image of a (32-channel block, piece): [octet o][slot = pixel ^ 4 o][8 channels].
    const int r_oct = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1), r_pix = 8 * (lane >> 5) + ((lane >> 2) & 3);
    const int lterm = MODE == 0 ? lane * 16 : r_oct * 256 + (r_pix ^ (4 * r_oct)) * 16 + (lane & 1) * 8;
    la[k] = k * STAGE + (wm * 2 * 3) * FRAG + lterm;
    lb[k] = k * STAGE + (A_FRAGS + wn * 2 * 3) * FRAG + lterm;
    asm volatile("" : "+v"(la[k]), "+v"(lb[k]));
  }

  // ---- the K loop.  Phase t of a wave:
  //     wait until its own loads of step t have landed (at most the 9 loads of steps t + 1, t + 2 outstanding)
  //     barrier      -> step t is complete in LDS, and every wave is past its reads of step t - 1: that slot is free
  //     request step t + 3 into it, read the fragments of step t, 24 MFMAs
  // The two waves of a SIMD run the same program between the same barriers: left alone both issue their loads (~100
  // cycles per LDS-DMA instruction) and wait for their fragment reads at the same time, with the matrix pipe idle — a
  // third of the kernel (205 -> 310 TFLOP/s at K = 2304 with the reads taken out).  Waves 4-7 therefore run HALF A STEP
  // LATE: they keep the second half of a step's products for after the next barrier, so one wave of every SIMD
  // multiplies while the other talks to memory (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  (Fragments
  // double-buffered in registers, the textbook answer, do not fit: 128 accumulators + 96 > 256.)
  u32x4 fa[2][3], fb[2][3];               // [tile][piece]
#if SP_ABL == 6
  u32x4 sa[2][3], sb[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 3; ++q) { sa[i][q] = u32x4{(unsigned)lane, 1u, 2u, 3u}; sb[i][q] = u32x4{4u, (unsigned)lane, 6u, 7u}; }
#endif
  auto read_frags = [&](int slot) __attribute__((always_inline)) {
#if SP_ABL != 3
    const char* const as = smem + la[slot];
    const char* const bs = smem + lb[slot];
    auto rd = [&](const char* ptr) __attribute__((always_inline)) -> u32x4 {
      if constexpr (MODE == 0) {
        return *reinterpret_cast<const u32x4*>(ptr);
      } else {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ptr));             // pixels 8 h + 0 .. 3
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)((uintptr_t)ptr ^ 64));   // pixels 8 h + 4 .. 7 (slot ^ 4)
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        return u32x4{l2[0], l2[1], h2[0], h2[1]};
      }
    };
#if SP_ABL == 5 || SP_ABL == 7          // timing only: a third / two thirds of the fragment reads
    fa[0][0] = rd(as); fb[0][0] = rd(bs); fb[1][0] = rd(bs + 3 * FRAG); fa[1][0] = rd(as + 3 * FRAG);
#if SP_ABL == 7
    fa[0][1] = rd(as + FRAG); fb[0][1] = rd(bs + FRAG); fb[1][1] = rd(bs + 4 * FRAG); fa[1][1] = rd(as + 4 * FRAG);
#else
    fa[0][1] = fa[0][0] + 1u; fb[0][1] = fb[0][0] + 1u; fb[1][1] = fb[1][0] + 1u; fa[1][1] = fa[1][0] + 1u;
#endif
    fa[0][2] = fa[0][1] + 3u; fb[0][2] = fb[0][1] + 3u; fb[1][2] = fb[1][1] + 3u; fa[1][2] = fa[1][1] + 3u;
#else
    // (in the order the products consume them: the first MFMA needs a[0][2] and b[0][0] only)
    fa[0][2] = rd(as + 2 * FRAG);
    fb[0][0] = rd(bs);
    fa[0][0] = rd(as);
    fb[0][2] = rd(bs + 2 * FRAG);
    fa[0][1] = rd(as + FRAG);
    fb[0][1] = rd(bs + FRAG);
#pragma unroll
    for (int q = 0; q < 3; ++q) fb[1][q] = rd(bs + (3 + q) * FRAG);
#pragma unroll
    for (int q = 0; q < 3; ++q) fa[1][q] = rd(as + (3 + q) * FRAG);
#endif
#endif
  };
#if SP_ABL == 3
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 3; ++q) { fa[i][q] = u32x4{(unsigned)lane, 1u, 2u, 3u}; fb[i][q] = u32x4{4u, (unsigned)lane, 6u, 7u}; }
#endif
  // the 12 MFMAs of tile row i; NEG: the A fragments negated (in place: they are dead afterwards)
  auto mfma_row = [&](auto i_tag, auto neg_tag) __attribute__((always_inline)) {
    constexpr int i = decltype(i_tag)::value;
    constexpr bool NEG = decltype(neg_tag)::value;
    if constexpr (NEG) {
#pragma unroll
      for (int q = 0; q < 3; ++q) fa[i][q] ^= 0x80008000u;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x16 c = acc[i][j];
#if SP_ABL == 6
      if (j == 0) asm volatile("" :: "v"(fa[i][0]), "v"(fa[i][1]), "v"(fa[i][2]), "v"(fb[i][0]), "v"(fb[i][1]), "v"(fb[i][2]));
      const bf16x8 a0 = __builtin_bit_cast(bf16x8, sa[i][0]), a1 = __builtin_bit_cast(bf16x8, sa[i][1]),
                   a2 = __builtin_bit_cast(bf16x8, sa[i][2]);
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, sb[j][0]), b1 = __builtin_bit_cast(bf16x8, sb[j][1]),
                   b2 = __builtin_bit_cast(bf16x8, sb[j][2]);
#else
      const bf16x8 a0 = __builtin_bit_cast(bf16x8, fa[i][0]), a1 = __builtin_bit_cast(bf16x8, fa[i][1]),
                   a2 = __builtin_bit_cast(bf16x8, fa[i][2]);
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, fb[j][0]), b1 = __builtin_bit_cast(bf16x8, fb[j][1]),
                   b2 = __builtin_bit_cast(bf16x8, fb[j][2]);
#endif
      // smallest terms first
#if SP_ABL == 8                            // timing only: the same FLOPs as pairs of v_mfma_f32_16x16x32_bf16 (the shape the chip
      {                                    // clocks higher on, profiles/r06_mfma_ceiling_bf16.txt) on the same fragment registers
        f32x4 lo = {c[0], c[1], c[2], c[3]}, hi = {c[4], c[5], c[6], c[7]};
#define SP_PAIR(a, b)                                                  \
        lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, lo, 0, 0, 0); \
        hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, hi, 0, 0, 0);
        SP_PAIR(a2, b0) SP_PAIR(a0, b2) SP_PAIR(a1, b1) SP_PAIR(a1, b0) SP_PAIR(a0, b1) SP_PAIR(a0, b0)
#undef SP_PAIR
        c[0] = lo[0]; c[1] = lo[1]; c[2] = lo[2]; c[3] = lo[3]; c[4] = hi[0]; c[5] = hi[1]; c[6] = hi[2]; c[7] = hi[3];
      }
#else
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
#endif
      acc[i][j] = c;
    }
  };
  // end of a period of four steps: the period's sums go into `tot` with the period's sign, `acc` restarts from zero.
  // In place, spelled as instructions (conv_igemm.hip found the same: written as `tot += acc; acc = 0` the compiler starts the
  // next period's products in a third register set).  The s_nops cover the 11 wait states an 8-pass MFMA result needs
  // before a VALU read, which nobody inserts for an asm block.
  auto flush = [&](auto neg_tag) __attribute__((always_inline)) {
    constexpr bool NEG = decltype(neg_tag)::value;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (NEG) tot[i][j] -= acc[i][j];
        else tot[i][j] += acc[i][j];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // phase S (0..3) of the period that starts at step t0.  KIND 0: somewhere in the middle (requests step t + 3); 1 .. 4:
  // phases 0 .. 3 of the LAST period (only its first phase has a step left to request; the waits count down).
  // FIRST: the kernel's first period (a late wave has no half step pending in its phase 0).
  auto phase = [&](int t0, auto s_tag, auto kind_tag, auto late_tag, auto neg_tag, auto first_tag) __attribute__((always_inline)) {
    constexpr int S = decltype(s_tag)::value;
    constexpr int KIND = decltype(kind_tag)::value;
    constexpr bool LATE = decltype(late_tag)::value;       // waves 4-7
    constexpr bool NEG = decltype(neg_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    if constexpr (KIND <= 2) wait_two();
    else if constexpr (KIND == 3) wait_one();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SP_ABL != 2) __builtin_amdgcn_s_barrier();
    if constexpr (LATE && !(FIRST && S == 0)) {
      // the second half of the step before this one: the last step of the previous period — opposite sign — in phase 0
      if constexpr (S == 0) { mfma_row(I1{}, std::integral_constant<bool, !NEG>{}); flush(std::integral_constant<bool, !NEG>{}); }
      else mfma_row(I1{}, neg_tag);
      __builtin_amdgcn_sched_barrier(0);
    }
    // Which goes first behind the barrier, this step's fragment reads or the LDS-DMA requests of step t + 3?  A wave's
    // ds_reads BEHIND its own global_load_lds return 1000-1800 cycles later than in front of them (gemm_sp16_kernel's stamps).
    // The weight gradient (24 transposing reads per step) gains 25-27 % from reading first (132 -> 168 and 176 -> 220 TFLOP/s
    // on the benchmark's 1x1 layers); the forward product, whose requests' lead is worth more than its 12 plain reads,
    // loses 2-4 % (-DSP_READS_FIRST / -DSP_DMA_FIRST force one order for both)
#if defined(SP_READS_FIRST)
    constexpr bool READS_FIRST = true;
#elif defined(SP_DMA_FIRST)
    constexpr bool READS_FIRST = false;
#else
    constexpr bool READS_FIRST = MODE == 1;
#endif
    // (the requests BEHIND the first row of products instead: -1 ... -4 % in both modes)
    if constexpr (READS_FIRST) {
      read_frags(S);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (KIND == 0 || KIND == 1) issue(t0 + S + 3, (S + 3) & 3, ((S + 3) & 1) != 0);
    } else {
      if constexpr (KIND == 0 || KIND == 1) issue(t0 + S + 3, (S + 3) & 3, ((S + 3) & 1) != 0);
      read_frags(S);
    }
#ifdef SP_FENCE_READS
    __builtin_amdgcn_sched_barrier(0);
#endif
    mfma_row(I0{}, neg_tag);
    if constexpr (!LATE) {
      mfma_row(I1{}, neg_tag);
      if constexpr (S == 3) flush(neg_tag);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using K0 = std::integral_constant<int, 0>;
  auto period_mid = [&](int t0, auto late_tag, auto neg_tag, auto first_tag) __attribute__((always_inline)) {
    phase(t0, std::integral_constant<int, 0>{}, K0{}, late_tag, neg_tag, first_tag);
    phase(t0, std::integral_constant<int, 1>{}, K0{}, late_tag, neg_tag, first_tag);
    phase(t0, std::integral_constant<int, 2>{}, K0{}, late_tag, neg_tag, first_tag);
    phase(t0, std::integral_constant<int, 3>{}, K0{}, late_tag, neg_tag, first_tag);
  };
  auto period_last = [&](int t0, auto late_tag, auto neg_tag, auto first_tag) __attribute__((always_inline)) {
    phase(t0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, late_tag, neg_tag, first_tag);
    phase(t0, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, late_tag, neg_tag, first_tag);
    phase(t0, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, late_tag, neg_tag, first_tag);
    phase(t0, std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{}, late_tag, neg_tag, first_tag);
    if constexpr (decltype(late_tag)::value) {       // the half step a late wave still owes
      mfma_row(I1{}, neg_tag);
      flush(neg_tag);
    }
  };
  // periods alternate in sign, starting with +; the first and the last one are peeled (K >= 128: at least two periods)
  auto k_loop = [&](auto late_tag) __attribute__((always_inline)) {
    const int nper = ksteps >> 2;
    period_mid(0, late_tag, std::false_type{}, std::true_type{});
    int per = 1;
    for (; per + 2 < nper; per += 2) {
      period_mid(4 * per, late_tag, std::true_type{}, std::false_type{});
      period_mid(4 * per + 4, late_tag, std::false_type{}, std::false_type{});
    }
    if (per + 2 == nper) {
      period_mid(4 * per, late_tag, std::true_type{}, std::false_type{});
      period_last(4 * per + 4, late_tag, std::false_type{}, std::false_type{});
    } else {
      period_last(4 * per, late_tag, std::true_type{}, std::false_type{});
    }
  };

  issue(0, 0, false);
  issue(1, 1, true);
  issue(2, 2, false);
  // (a static s_setprio 1 for the late half — MI355X_MICROARCH.md "Two waves per SIMD", item 4 — changes nothing here:
  //  125.0 / 199.7 / 228.0 against 125.4 / 199.4 / 228.1 TFLOP/s on the three benchmark shapes)
  if (w < 4) k_loop(std::false_type{});
  else k_loop(std::true_type{});

  if (clk_block) { atomicAdd(&g_sp_clk_ticks[0], clock64() - clk_c0); atomicAdd(&g_sp_clk_ticks[1], wall_clock64() - clk_w0); }
  if (SP_ABL == 4) { if (tot[0][0][0] == 123.f) p.out[0] = tot[1][1][3] + tot[0][1][2] + tot[1][0][1]; return; }
  if constexpr (MODE == 1) {
    // combine: float atomics into out[n][c] — per accumulator register two 128-byte row segments, the full-rate shape
    float* const ob = p.out + batch * p.bs_out;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + (wn * 2 + j) * 32 + li;
        if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < p.M) atomicAdd(ob + (size_t)row * p.ld_out + col, tot[i][j][r]);
          }
        }
      }
    return;
  }
  // ---- epilogue (conv_igemm.hip's): plain products on whole tiles store straight from the accumulators, everything else
  // goes through an LDS transpose so that every lane stores — and reads the optional operands as — 16-byte channel runs
  if (!p.bias && !p.relu && !p.accumulate && !p.mask && !p.mask_bits && !p.gate_out && !p.out_planes && !p.colsum && m0 + SP_BM <= p.M) {
    float* const ob = p.out + batch * p.bs_out;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + (wn * 2 + j) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          ob[(size_t)row * p.ld_out + col] = tot[i][j][r];
        }
      }
    return;
  }
  __syncthreads();
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (wn * 2 + j) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        Cs[row * LDC + col] = tot[i][j][r] + bias_v[j];
      }
    }
  __syncthreads();
  sp_epilogue_from_lds(p, Cs, m0, n0, batch, tid, lane);
}

// ---- the forward / data-gradient product with TWO workgroups per CU (short contractions) ------------------------------------
// gemm_sp_kernel<0> keeps one 512-thread workgroup on a CU: nothing runs under a tile's pipeline fill (the first stages come
// from HBM) or under its epilogue, and on the contractions of the benchmark networks — 256 and 768 deep: 16 and 48 steps —
// those are a third of a tile's time (125 / 170-190 TFLOP/s where K = 2304 reaches 205-225).  Here a workgroup is 256
// threads = 4 waves (2 x 2) of 64 x 64 on a 128 x 128 tile with a ring of THREE 24-KB stages (72 KB): two of them share a
// CU — one wave of each on every SIMD, 256 registers each — and one's fill and epilogue fall under the other's products.
// No late half: the two workgroups of a CU are out of phase by themselves.  Costs: 24 KB of operands per 0.52 MFLOP
// instead of 36 KB per 1.05 (a third more traffic out of L2), six LDS-DMA requests per wave and step instead of 4.5.
constexpr int S2_BM = 128, S2_BN = 128;
constexpr int S2_A_FRAGS = S2_BM / 32 * 3, S2_B_FRAGS = S2_BN / 32 * 3;      // 12 + 12
constexpr int S2_STAGE = (S2_A_FRAGS + S2_B_FRAGS) * FRAG;                   // 24 KB
constexpr int S2_RING = 3;
__global__ __launch_bounds__(256, 2) void gemm_sp2_kernel(const SpP p) {
  __shared__ __attribute__((aligned(16))) char smem[S2_RING * S2_STAGE];       // 72 KB; the C tile (66 KB) in the epilogue
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int nbm = (p.M + S2_BM - 1) / S2_BM, nbn = p.N / S2_BN;
  const int v = xcd_remap(blockIdx.x, nbm * nbn);
  const int tile_n = v % nbn, tile_m = v / nbn;
  const int batch = blockIdx.y;
  const int ksteps = p.ksteps;
  const int m0 = tile_m * S2_BM, n0 = tile_n * S2_BN;
  const bool clk_block = blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && threadIdx.x == 0;
  unsigned long long clk_c0 = 0, clk_w0 = 0;
  if (clk_block) { clk_c0 = clock64(); clk_w0 = wall_clock64(); }

  // loads: the 24 fragments of a stage, six per wave (f = w + 4 q): A's twelve, then B's
  const char* const Ab = p.A + batch * p.bs_a;
  const char* const Bb = p.B + batch * p.bs_b;
  const char* gsrc[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int f = w + 4 * q;
    if (f < S2_A_FRAGS) {
      int rb = tile_m * (S2_BM / 32) + f / 3;
      if (rb > p.rb_a - 1) rb = p.rb_a - 1;
      gsrc[q] = Ab + ((long long)rb * ksteps * 3 + f % 3) * FRAG;
    } else {
      const int g = f - S2_A_FRAGS;
      const int nb = tile_n * (S2_BN / 32) + g / 3;
      gsrc[q] = Bb + ((long long)nb * ksteps * 3 + g % 3) * FRAG;
    }
  }
  const unsigned int lane16 = (unsigned int)lane * 16u;
  auto issue = [&](int t, int slot) __attribute__((always_inline)) {
    char* const dst = smem + slot * S2_STAGE;
    const unsigned int voff = lane16 + (unsigned int)t * KSTEP;
#pragma unroll
    for (int q = 0; q < 6; ++q) glds16(gsrc[q] + (size_t)voff, dst + (w + 4 * q) * FRAG);
  };

  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
  const int li = lane & 31, lh = lane >> 5;
  unsigned int la[S2_RING], lb[S2_RING];
#pragma unroll
  for (int k = 0; k < S2_RING; ++k) {
    la[k] = k * S2_STAGE + (wm * 2 * 3) * FRAG + lane * 16;
    lb[k] = k * S2_STAGE + (S2_A_FRAGS + wn * 2 * 3) * FRAG + lane * 16;
    asm volatile("" : "+v"(la[k]), "+v"(lb[k]));
  }
  u32x4 fa[2][3], fb[2][3];
  auto read_frags = [&](int slot) __attribute__((always_inline)) {
    const char* const as = smem + la[slot];
    const char* const bs = smem + lb[slot];
    fa[0][2] = *reinterpret_cast<const u32x4*>(as + 2 * FRAG);
    fb[0][0] = *reinterpret_cast<const u32x4*>(bs);
    fa[0][0] = *reinterpret_cast<const u32x4*>(as);
    fb[0][2] = *reinterpret_cast<const u32x4*>(bs + 2 * FRAG);
    fa[0][1] = *reinterpret_cast<const u32x4*>(as + FRAG);
    fb[0][1] = *reinterpret_cast<const u32x4*>(bs + FRAG);
#pragma unroll
    for (int q = 0; q < 3; ++q) fb[1][q] = *reinterpret_cast<const u32x4*>(bs + (3 + q) * FRAG);
#pragma unroll
    for (int q = 0; q < 3; ++q) fa[1][q] = *reinterpret_cast<const u32x4*>(as + (3 + q) * FRAG);
  };
  auto mfma_step = [&](auto neg_tag, auto row_tag) __attribute__((always_inline)) {
    constexpr bool NEG = decltype(neg_tag)::value;
    constexpr int ROW = decltype(row_tag)::value;             // -1: both rows of tiles
    if constexpr (NEG) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (ROW < 0 || ROW == i)
#pragma unroll
          for (int q = 0; q < 3; ++q) fa[i][q] ^= 0x80008000u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (ROW < 0 || ROW == i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, fa[i][0]), a1 = __builtin_bit_cast(bf16x8, fa[i][1]),
                     a2 = __builtin_bit_cast(bf16x8, fa[i][2]);
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, fb[j][0]), b1 = __builtin_bit_cast(bf16x8, fb[j][1]),
                     b2 = __builtin_bit_cast(bf16x8, fb[j][2]);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);       // smallest terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
        acc[i][j] = c;
      }
  };
  auto flush = [&](auto neg_tag) __attribute__((always_inline)) {
    constexpr bool NEG = decltype(neg_tag)::value;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (NEG) tot[i][j] -= acc[i][j];
        else tot[i][j] += acc[i][j];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  // step t: this wave's six loads of step t have landed when at most the six of step t + 1 are in flight; behind the barrier
  // the whole stage is in LDS and every wave is past its reads of step t - 1, whose slot takes step t + 2.
  // TAIL: 0 = a step t + 2 exists; 1 = the last but one step; 2 = the last
  auto step = [&](int t, int slot, auto tail_tag, auto neg_tag) __attribute__((always_inline)) {
    constexpr int TAIL = decltype(tail_tag)::value;
    if constexpr (TAIL == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the fragment reads in front of the LDS-DMA requests: a wave's ds_reads behind its own global_load_lds return late
    // (gemm_sp16_kernel's stamps), and with a second workgroup on the CU the requests' lead matters less than in
    // gemm_sp_kernel<0> (-DSP2_DMA_FIRST, the other order: -1 ... -7 % on the benchmark shapes)
#ifdef SP2_DMA_FIRST
    if constexpr (TAIL == 0) issue(t + 2, slot == 0 ? 2 : slot - 1);
    read_frags(slot);
#else
    read_frags(slot);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAIL == 0) issue(t + 2, slot == 0 ? 2 : slot - 1);
#endif
    mfma_step(neg_tag, std::integral_constant<int, -1>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  // a period of four steps (64 k) that starts at step t0, whose slot is s0 (t0 % 3); LAST: the kernel's last period
  auto period = [&](int t0, int s0, auto neg_tag, auto last_tag) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value;
    const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
    step(t0, s0, T0{}, neg_tag);
    step(t0 + 1, s1, T0{}, neg_tag);
    if constexpr (LAST) { step(t0 + 2, s2, T1{}, neg_tag); step(t0 + 3, s0, T2{}, neg_tag); }
    else { step(t0 + 2, s2, T0{}, neg_tag); step(t0 + 3, s0, T0{}, neg_tag); }
    flush(neg_tag);
  };
  issue(0, 0);
  issue(1, 1);
  {
    const int nper = ksteps >> 2;           // K >= 128: at least two periods; signs alternate, starting with +
    int s0 = 0;                             // slot of the period's first step: (4 per) % 3 = per % 3
    for (int per = 0; per + 1 < nper; ++per) {
      if (per & 1) period(4 * per, s0, std::true_type{}, std::false_type{});
      else period(4 * per, s0, std::false_type{}, std::false_type{});
      s0 = s0 == 2 ? 0 : s0 + 1;
    }
    if ((nper - 1) & 1) period(4 * (nper - 1), s0, std::true_type{}, std::true_type{});
    else period(4 * (nper - 1), s0, std::false_type{}, std::true_type{});
  }
  if (clk_block) { atomicAdd(&g_sp_clk_ticks[0], clock64() - clk_c0); atomicAdd(&g_sp_clk_ticks[1], wall_clock64() - clk_w0); }
  // ---- epilogue, as gemm_sp_kernel<0>'s
  if (!p.bias && !p.relu && !p.accumulate && !p.mask && !p.mask_bits && !p.gate_out && !p.out_planes && !p.colsum && m0 + S2_BM <= p.M) {
    float* const ob = p.out + batch * p.bs_out;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + (wn * 2 + j) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          ob[(size_t)row * p.ld_out + col] = tot[i][j][r];
        }
      }
    return;
  }
  __syncthreads();
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = (wn * 2 + j) * 32 + li;
    const int n = n0 + col;
    const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        Cs[row * LDC + col] = tot[i][j][r] + bv;
      }
  }
  __syncthreads();
  sp_epilogue_from_lds<S2_BM, 4>(p, Cs, m0, n0, batch, tid, lane);
}

// ---- the forward / data-gradient product on v_mfma_f32_16x16x32_bf16 (opt-in: CLX_SP_MFMA=16) -------------------------------
// The same tile, ring, planes, two-level summation and epilogue as gemm_sp_kernel<0>; the six products of a fragment pair as
// 16 x 16 x 32 instructions.  Why: the chip holds a higher clock on this shape (an accumulator register is read and written
// once per 32 products instead of once per 16: profiles/r06_mfma_ceiling_bf16.txt, 2.03 against 1.80 GHz on random operands;
// in this kernel 1.96 against 1.64 GHz on the K = 2304 launch).
// A 16 x 16 x 32 operand is 16 rows x 32 k: lane l holds row l % 16, k = 8 (l / 16) .. + 7 — the 16 bytes the P3 fragment of the
// k step (l / 32 of a PAIR of ring stages), half (l / 16) & 1, keeps for that row: the LDS image and the LDS-DMA stay as they
// are, a DOUBLE step (two ring stages, 32 k) is multiplied between barriers, and the ring is two double stages.
// Registers: 64 + 64 accumulators, the A fragments of the double step (4 row groups x 3 pieces: 48) and two of the four
// column groups of B at a time (2 x 12).  Waves 4-7 run half a double step late, as in gemm_sp_kernel.
// What the stamps of the diagnostic build (-DSP16_STAMP, tools/exp/sp16_stamps.py) taught:
//   * a wave's ds_reads BEHIND its own global_load_lds return 1000-1800 cycles later than in front of them (18 reads: 1050
//     behind three requests, 570 in front): every segment reads first and requests afterwards;
//   * back-to-back 16 x 16 x 32 instructions hold the SIMD's vector issue half of the time: the partner wave's 18 ds_read_b128
//     take 700-1000 cycles beside them (400 beside nothing), and the older wave wins every issue slot it wants — without
//     priorities the early half's products ran INSIDE the late half's instead of behind them and the late half's reads were
//     covered by nothing: s_setprio 1 for the late half, 3 around the early half's reads, 0 for its products.
// Measured (one MI355X, TFLOP/s f32-equivalent, this kernel / gemm_sp_kernel<0>): K = 2304: 212-216 / 204-208; K = 768:
// 162 / 170; K = 256: 101 / 126 (the longer fill of the two-double-stage ring and the LDS epilogue of every tile).  The
// contraction lengths of the benchmark networks are 256 and 768: NOT the default.  Matrix pipe 0.71 busy at 1.77 GHz here,
// 0.73 at 1.67 GHz there (profiles/r06_pmc_sp16_vs_sp32.txt): occupancy won back by the scheduling came off the clock.
typedef float f32x4_ __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 1) void gemm_sp16_kernel(const SpP p) {
  __shared__ __attribute__((aligned(16))) char smem[RING * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SP16_STAMP
  __shared__ unsigned long long stamp_lds[2][64];
  int stamp_n = 0;
  const bool stamp_wave = blockIdx.x == 0 && blockIdx.y == 0 && (w == 0 || w == 4);
  auto stamp = [&](int d) __attribute__((always_inline)) {
    if (stamp_wave && d >= 16 && d < 20) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (lane == 0) stamp_lds[w >> 2][stamp_n & 63] = t;
      ++stamp_n;
    }
  };
#define SP_STAMP(d) stamp(d)
#else
#define SP_STAMP(d)
#endif
  const int wm = w >> 1, wn = w & 1;
  const int v = xcd_remap(blockIdx.x, p.nbm * p.nbn);
  const int tile_n = v % p.nbn, tile_m = v / p.nbn;
  const int batch = blockIdx.y;
  const int ksteps = p.ksteps;
  const int m0 = tile_m * SP_BM, n0 = tile_n * SP_BN;
  const bool clk_block = blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && threadIdx.x == 0;
  unsigned long long clk_c0 = 0, clk_w0 = 0;
  if (clk_block) { clk_c0 = clock64(); clk_w0 = wall_clock64(); }

  // ---- loads of a double step: 48 fragments of A (six per wave: f = w + 8 q; stage f / 24 of the pair, fragment f % 24 of it)
  // and 24 of B (three per wave: stage f / 12, fragment f % 12), requested at different times (below)
  const char* const Ab = p.A + batch * p.bs_a;
  const char* const Bb = p.B + batch * p.bs_b;
  const char* gsrc_a[6];
  const char* gsrc_b[3];
  int ldst_a[6], ldst_b[3];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int f = w + 8 * q, sub = f / A_FRAGS, idx = f % A_FRAGS;
    int rb = tile_m * (SP_BM / 32) + idx / 3;
    if (rb > p.rb_a - 1) rb = p.rb_a - 1;
    gsrc_a[q] = Ab + ((long long)rb * ksteps * 3 + idx % 3) * FRAG + sub * KSTEP;
    ldst_a[q] = sub * STAGE + idx * FRAG;
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int f = w + 8 * q, sub = f / B_FRAGS, idx = f % B_FRAGS;
    const int nb = tile_n * (SP_BN / 32) + idx / 3;
    gsrc_b[q] = Bb + ((long long)nb * ksteps * 3 + idx % 3) * FRAG + sub * KSTEP;
    ldst_b[q] = sub * STAGE + (A_FRAGS + idx) * FRAG;
  }
  const unsigned int lane16 = (unsigned int)lane * 16u;
#ifndef SP16_ABL
#define SP16_ABL 0
#endif
  auto issue_a = [&](int d, int pair) __attribute__((always_inline)) {
    if (SP16_ABL == 1 && d > 1) return;
    char* const dst = smem + pair * 2 * STAGE;
    const unsigned int voff = lane16 + (unsigned int)d * (2u * KSTEP);
#pragma unroll
    for (int q = 0; q < 6; ++q) glds16(gsrc_a[q] + (size_t)voff, dst + ldst_a[q]);
  };
  auto issue_b = [&](int d, int pair) __attribute__((always_inline)) {
    if (SP16_ABL == 1 && d > 1) return;
    char* const dst = smem + pair * 2 * STAGE;
    const unsigned int voff = lane16 + (unsigned int)d * (2u * KSTEP);
#pragma unroll
    for (int q = 0; q < 3; ++q) glds16(gsrc_b[q] + (size_t)voff, dst + ldst_b[q]);
  };

  f32x4_ acc[4][4], tot[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

  // LDS addresses of this wave's operands per double stage (opaque: two registers per operand, the rest instruction offsets)
  unsigned int la[2], lb[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int g = lane >> 4;
    const int lterm = (g >> 1) * STAGE + (g & 1) * 512 + (lane & 15) * 16;
    la[u] = u * 2 * STAGE + (wm * 2 * 3) * FRAG + lterm;
    lb[u] = u * 2 * STAGE + (A_FRAGS + wn * 2 * 3) * FRAG + lterm;
    asm volatile("" : "+v"(la[u]), "+v"(lb[u]));
  }
  u32x4 fa[4][3], fb[2][2][3];            // A: [row group][piece]; B: [buffer][column group of the pair][piece]
  // group G of 16 rows (columns) = row block G >> 1 of the wave's two, rows 16 (G & 1) .. of it
  auto read_a = [&](int pair) __attribute__((always_inline)) {
    const char* const as = smem + la[pair];
#pragma unroll
    for (int G = 0; G < 4; ++G)
#pragma unroll
      for (int q = 0; q < 3; ++q) fa[G][q] = *reinterpret_cast<const u32x4*>(as + ((G >> 1) * 3 + q) * FRAG + (G & 1) * 256);
  };
  auto read_b = [&](int pair, auto half_tag, auto buf_tag) __attribute__((always_inline)) {
    constexpr int H = decltype(half_tag)::value, BUF = decltype(buf_tag)::value;
    const char* const bs = smem + lb[pair];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int q = 0; q < 3; ++q) fb[BUF][jj][q] = *reinterpret_cast<const u32x4*>(bs + (H * 3 + q) * FRAG + jj * 256);
  };
  auto negate_a = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int G = 0; G < 4; ++G)
#pragma unroll
      for (int q = 0; q < 3; ++q) fa[G][q] ^= 0x80008000u;
  };
  // the 48 MFMAs of two column groups (half H of the wave's 64 columns): product-major, so that the four row groups'
  // instructions lie between two that accumulate into the same registers
  auto mfma_half = [&](auto half_tag, auto buf_tag) __attribute__((always_inline)) {
    constexpr int H = decltype(half_tag)::value, BUF = decltype(buf_tag)::value;
#if SP16_ABL == 4                           // timing only: no products (the fragments still have to arrive)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int q = 0; q < 3; ++q) asm volatile("" :: "v"(fb[BUF][jj][q]), "v"(fa[jj][q]), "v"(fa[2 + jj][q]));
    return;
#endif
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // smallest terms first
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int G = 0; G < 4; ++G)
          acc[G][2 * H + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[G][PA[t]]),
                                                                        __builtin_bit_cast(bf16x8, fb[BUF][jj][PB[t]]),
                                                                        acc[G][2 * H + jj], 0, 0, 0);
    }
  };
  auto flush = [&](auto neg_tag) __attribute__((always_inline)) {
    constexpr bool NEG = decltype(neg_tag)::value;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (NEG) tot[i][j] -= acc[i][j];
        else tot[i][j] += acc[i][j];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  // Double step d of a period (S = 0: its first, 1: its second and last; it lives in double stage S) is two half steps
  // between three barriers:
  //   OPEN  — A(d + 1) excepted, this wave's loads have landed; every wave is past its reads of double step d - 1:
  //           request B(d + 1) into the other double stage.  An early wave reads A, B0, B1 and multiplies the first half of
  //           the columns; a late wave (4-7) multiplies the second half of double step d - 1 from registers, then reads.
  //   MID   — every wave holds A(d) in registers: request A(d + 2) into THIS double stage (a lead of one and a half double
  //           steps for two thirds of the bytes; the B fragments — weights, shared by every block — have one).  An early
  //           wave reads B2, B3 and multiplies the second half; a late wave multiplies the first half, then reads B2, B3
  //           and keeps them.
  // The two waves of a SIMD thus alternate between reading and multiplying, as in gemm_sp_kernel.
  // A1: double step d + 1 exists (its A fragments are this wave's six loads still in flight at OPEN); A2: d + 2 exists.
  auto dstep = [&](int d, auto s_tag, auto late_tag, auto neg_tag, auto first_tag, auto a1_tag, auto a2_tag) __attribute__((always_inline)) {
    constexpr int S = decltype(s_tag)::value;
    constexpr bool LATE = decltype(late_tag)::value, NEG = decltype(neg_tag)::value, FIRST = decltype(first_tag)::value;
    constexpr bool A1 = decltype(a1_tag)::value, A2 = decltype(a2_tag)::value;
    if constexpr (A1) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    SP_STAMP(d);                                                  // 0: at OPEN (after the wait for the loads)
    if (SP16_ABL != 2) __builtin_amdgcn_s_barrier();
    SP_STAMP(d);                                                  // 1: through OPEN
    // (an LDS-DMA instruction costs its wave 60-180 cycles of issue: the half that multiplies first requests afterwards,
    //  so that one wave of every SIMD feeds the matrix pipe while the other talks to memory)
    if constexpr (!LATE) {
#ifndef SP16_NO_PRIO
      __builtin_amdgcn_s_setprio(3);         // an early wave's READS go in front of the late half's products, its products behind
#endif
      read_a(S);
      read_b(S, H0{}, H0{});
#ifndef SP16_NO_PRIO
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
#endif
      __builtin_amdgcn_sched_barrier(0); SP_STAMP(d); __builtin_amdgcn_sched_barrier(0);      // 2: reads issued
      if constexpr (A1) issue_b(d + 1, S ^ 1);
#ifdef SP16_STAMP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      __builtin_amdgcn_sched_barrier(0); SP_STAMP(d); __builtin_amdgcn_sched_barrier(0);      // 3: B requested, fragments in registers
      if constexpr (NEG) negate_a();
      mfma_half(H0{}, H0{});
    } else {
      if constexpr (!(FIRST && S == 0)) {
        mfma_half(H1{}, H1{});
        if constexpr (S == 0) flush(std::integral_constant<bool, !NEG>{});       // (the previous period's last half)
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0); SP_STAMP(d); __builtin_amdgcn_sched_barrier(0);      // 2: products issued
      read_a(S);
      read_b(S, H0{}, H0{});
      __builtin_amdgcn_sched_barrier(0); SP_STAMP(d); __builtin_amdgcn_sched_barrier(0);      // 3: reads issued
      if constexpr (A1) issue_b(d + 1, S ^ 1);
      if constexpr (NEG) negate_a();
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SP_STAMP(d);                                                  // 4: at MID
    if (SP16_ABL != 2) __builtin_amdgcn_s_barrier();
    SP_STAMP(d);                                                  // 5: through MID
    if constexpr (!LATE) {
#ifndef SP16_NO_PRIO
      __builtin_amdgcn_s_setprio(3);
#endif
      read_b(S, H1{}, H1{});
#ifndef SP16_NO_PRIO
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
#endif
      __builtin_amdgcn_sched_barrier(0); SP_STAMP(d); __builtin_amdgcn_sched_barrier(0);      // 6: reads issued
      if constexpr (A2) issue_a(d + 2, S);
#ifdef SP16_STAMP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      __builtin_amdgcn_sched_barrier(0); SP_STAMP(d); __builtin_amdgcn_sched_barrier(0);      // 7: A requested, fragments in registers
      mfma_half(H1{}, H1{});
      if constexpr (S == 1) flush(neg_tag);
    } else {
      mfma_half(H0{}, H0{});
      __builtin_amdgcn_sched_barrier(0); SP_STAMP(d); __builtin_amdgcn_sched_barrier(0);      // 6: products issued
      read_b(S, H1{}, H1{});
      __builtin_amdgcn_sched_barrier(0); SP_STAMP(d); __builtin_amdgcn_sched_barrier(0);      // 7: reads issued
      if constexpr (A2) issue_a(d + 2, S);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // a period of 64 k = two double steps.  TAIL: 0 = two more periods follow at least, 1 = the last but one, 2 = the last
  auto period = [&](int per, auto late_tag, auto neg_tag, auto first_tag, auto tail_tag) __attribute__((always_inline)) {
    constexpr int TAIL = decltype(tail_tag)::value;
    using T = std::true_type;
    dstep(2 * per, H0{}, late_tag, neg_tag, first_tag, T{}, std::integral_constant<bool, TAIL != 2>{});
    dstep(2 * per + 1, H1{}, late_tag, neg_tag, first_tag, std::integral_constant<bool, TAIL != 2>{},
          std::integral_constant<bool, TAIL != 2>{});
    if constexpr (TAIL == 2 && decltype(late_tag)::value) {     // the half double step a late wave still owes
      mfma_half(H1{}, H1{});
      flush(neg_tag);
    }
  };
  // periods alternate in sign, starting with +; the first and the last one are peeled (K >= 128: at least two periods)
  using P0 = std::integral_constant<int, 0>;
  using P2 = std::integral_constant<int, 2>;
  auto k_loop = [&](auto late_tag) __attribute__((always_inline)) {
    const int nper = ksteps >> 2;
    using T = std::true_type;
    using F = std::false_type;
    period(0, late_tag, F{}, T{}, P0{});
    int per = 1;
    for (; per + 2 < nper; per += 2) {
      period(per, late_tag, T{}, F{}, P0{});
      period(per + 1, late_tag, F{}, F{}, P0{});
    }
    if (per + 2 == nper) {
      period(per, late_tag, T{}, F{}, P0{});
      period(per + 1, late_tag, F{}, F{}, P2{});
    } else {
      period(per, late_tag, T{}, F{}, P2{});
    }
  };
  issue_a(0, 0);
  issue_b(0, 0);
  issue_a(1, 1);
  // The late half multiplies FIRST behind every barrier, from registers, while the early half reads; it has to be through
  // its products when the early half's fragments arrive, or the two blocks of products run side by side and the late
  // half's reads behind them are covered by nothing: priority for its instructions (measured with the stamps of the
  // diagnostic build: the late half's 48 products took 1650 cycles beside the early half's, its reads 600 more)
#ifndef SP16_NO_PRIO
  if (w >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  if (w < 4) k_loop(std::false_type{});
  else k_loop(std::true_type{});
  __builtin_amdgcn_s_setprio(0);

#ifdef SP16_STAMP
  if (stamp_wave && lane < 64) g_sp16_stamps[w >> 2][lane] = lane < stamp_n ? stamp_lds[w >> 2][lane] : 0ull;
#endif
  if (clk_block) { atomicAdd(&g_sp_clk_ticks[0], clock64() - clk_c0); atomicAdd(&g_sp_clk_ticks[1], wall_clock64() - clk_w0); }
  // ---- epilogue: through the LDS transpose (accumulator register r of lane l: row 4 (l / 16) + r, column l % 16 of its 16 x 16)
  __syncthreads();
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = wn * 64 + 16 * j + (lane & 15);
    const int n = n0 + col;
    const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[(wm * 64 + 16 * i + 4 * (lane >> 4) + r) * LDC + col] = tot[i][j][r] + bv;
  }
  __syncthreads();
  sp_epilogue_from_lds(p, Cs, m0, n0, batch, tid, lane);
}

__device__ __attribute__((aligned(16))) float g_sp_zero16[4] = {0.f, 0.f, 0.f, 0.f};

}  // namespace

static const float* sp_zero_buffer() {
  static const float* cache[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (cache[dev] == nullptr) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_sp_zero16)) != hipSuccess) return nullptr;
    cache[dev] = (const float*)ptr;
  }
  return cache[dev];
}

int clx_sp_zero_tail(void* planes, long long rows, int K, int batch, long long bs, hipStream_t st) {
  if (sp::padded_rows(rows) == rows || batch <= 0) return CLX_OK;
  const int ksteps = K / 16;
  const long long total = (sp::padded_rows(rows) / 32 - rows / 32) * ksteps * 6 * 32;
  sp_zero_tail_kernel<<<dim3((unsigned)cdiv(total, 256), (unsigned)batch), 256, 0, st>>>((char*)planes, bs, rows, ksteps);
  return CLX_OK;
}

extern "C" size_t clx_planes_bytes(long long rows, int K) {
  if (rows <= 0 || K <= 0 || K % 16 != 0) return 0;
  return (size_t)sp::planes_bytes(rows, K);
}

// colsum != NULL: colsum[k] += sum_rows x[row][k] for k < nreal as well (the bias gradient, when x is a layer's dY)
int clx_sp_split(const float* x, long long ld, long long rows, int K, void* planes, float* colsum, int nreal, hipStream_t st) {
  CLX_REQUIRE(x != nullptr && planes != nullptr, "clx_split_planes: null pointer");
  CLX_REQUIRE(rows > 0 && K > 0 && K % 16 == 0 && ld >= K && ld % 4 == 0, "clx_split_planes: K must be a multiple of 16, ld >= K a multiple of 4");
  CLX_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 15) == 0, "clx_split_planes: pointers must be 16-byte aligned");
  const int ksteps = K / 16;
  const long long nfrag = sp::padded_rows(rows) / 32 * ksteps;
  long long blocks = (nfrag + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  if (colsum != nullptr) {
    // wavefronts (4 per block) a multiple of the k steps: blocks a multiple of ksteps / gcd(ksteps, 4).  A resident grid
    // (8 blocks per CU) and no more: every wavefront ends in 16 atomics on the K column sums, and 8192 blocks' worth of
    // them on 256 addresses took as long as the split itself
    if (blocks > 2048) blocks = 2048;
    int g = ksteps % 4 == 0 ? 4 : ksteps % 2 == 0 ? 2 : 1;
    const long long unit = ksteps / g;
    blocks = (blocks + unit - 1) / unit * unit;
    CLX_LAUNCH_KIND(CLX_PROF_SPLIT_PLANES, sp_split_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, x, ld, rows, ksteps, (char*)planes,
                    nfrag, colsum, nreal);
  } else {
    CLX_LAUNCH_KIND(CLX_PROF_SPLIT_PLANES, sp_split_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, x, ld, rows, ksteps, (char*)planes,
                    nfrag, (float*)nullptr, 0);
  }
  return CLX_OK;
}

extern "C" int clx_split_planes(const float* x, long long ld, long long rows, int K, void* planes, clx_stream stream) {
  const int rc = clx_sp_split(x, ld, rows, K, planes, nullptr, 0, (hipStream_t)stream);
  if (rc) return rc;
  CLX_CHECK_LAUNCH("clx_split_planes");
  return CLX_OK;
}

extern "C" int clx_join_planes(const void* planes, long long rows, int K, float* x, long long ld, clx_stream stream) {
  CLX_REQUIRE(x != nullptr && planes != nullptr, "clx_join_planes: null pointer");
  CLX_REQUIRE(rows > 0 && K > 0 && K % 16 == 0 && ld >= K, "clx_join_planes: K must be a multiple of 16, ld >= K");
  const long long nfrag = sp::padded_rows(rows) / 32 * (K / 16);
  long long blocks = (nfrag + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  sp_join_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>((const char*)planes, ld, rows, K / 16, x, nfrag);
  CLX_CHECK_LAUNCH("clx_join_planes");
  return CLX_OK;
}

bool clx_sp_applicable(const clx_conv_desc* d) {
  if (d->precision != CLX_PREC_F32X3BF16 || d->wplanes == nullptr || d->nsrc != 1) return false;
  if (d->KD != 1 || d->KH != 1 || d->KW != 1 || d->PD || d->PH || d->PW) return false;
  const clx_src& S = d->src[0];
  if (S.fz != 1 || S.fy != 1 || S.fx != 1 || S.oz || S.oy || S.ox) return false;
  if (S.D != d->ID || S.H != d->IH || S.W != d->IW) return false;
  return d->N % SP_BN == 0 && S.C % 64 == 0 && S.C >= 128 && d->ld_out % 4 == 0;
}

extern "C" int clx_conv_sp_covers(const clx_conv_desc* d) {
  return d != nullptr && d->algo == CLX_ALGO_DIRECT && d->aplanes != nullptr && clx_sp_applicable(d) ? 1 : 0;
}

// the batched product behind clx_gemm_planes and the precision switch of clx_conv_fwd: `batch` problems, operand b at
// A + b * bs_a / B + b * bs_b (bytes), result at out + b * bs_out (floats)
int clx_sp_launch(const void* A, const void* B, int M, int N, int K, long long rows_a, int batch, long long bs_a, long long bs_b,
                  long long bs_out, const clx_conv_desc* ep, hipStream_t st) {
  CLX_REQUIRE(M > 0 && N > 0 && N % SP_BN == 0 && K >= 128 && K % 64 == 0, "clx_gemm_planes: needs N %% 128 == 0, K %% 64 == 0 and K >= 128");
  CLX_REQUIRE(rows_a >= M, "clx_gemm_planes: the A planes hold fewer rows than M");
  SpP p = {};
  p.A = (const char*)A; p.B = (const char*)B;
  p.bs_a = bs_a; p.bs_b = bs_b; p.bs_out = bs_out;
  p.out = ep->out; p.ld_out = ep->ld_out;
  p.M = M; p.N = N; p.ksteps = K / 16;
  p.rb_a = (int)(sp::padded_rows(rows_a) / 32);
  p.bias = ep->bias; p.mask = ep->mask; p.mask_bits = ep->mask_bits; p.gate_out = ep->gate_out;
  p.relu = ep->relu; p.accumulate = ep->accumulate; p.ld_mask = ep->ld_mask; p.ld_mask_bits = ep->ld_mask_bits; p.ld_gate = ep->ld_gate;
  p.zeros = sp_zero_buffer();
  CLX_REQUIRE(p.zeros != nullptr, "clx_gemm_planes: cannot resolve the device zero buffer");
  p.out_planes = batch == 1 ? (char*)ep->out_planes : nullptr;
  p.colsum = batch == 1 ? ep->out_colsum : nullptr; p.colsum_n = N;
  CLX_REQUIRE(p.out_planes == nullptr || ep->ld_out == N, "clx_gemm_planes: out_planes needs a dense output (ld_out == N)");
  p.nbm = cdiv(M, SP_BM); p.nbn = N / SP_BN;
  // CLX_SP_TILE=128: gemm_sp2_kernel (128 x 128 tiles, two workgroups per CU) always; =256: never; unset: up to K = 1024
  // (measured against gemm_sp_kernel<0>: K = 256 +8 ... +22 %, K = 768 0 ... +6 %, K = 2304 -4 %)
  const char* const tile_env = getenv("CLX_SP_TILE");
  const int tile_choice = tile_env != nullptr ? atoi(tile_env) : 0;
  const char* const maxk_env = getenv("CLX_SP_TILE_MAXK");           // (the rule's threshold, for measurements)
  const int maxk = maxk_env != nullptr ? atoi(maxk_env) : 1024;
  const bool small_tiles = tile_choice == 128 || (tile_choice != 256 && K <= maxk);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (clx_prof_enabled()) clx_prof_events(small_tiles ? CLX_PROF_GEMM_SP2 : CLX_PROF_GEMM_SP, 2.0 * M * N * K * batch, &e0, &e1);
  // CLX_SP_MFMA=16: the 16 x 16 x 32 form of the kernel (gemm_sp16_kernel: faster from K ~ 2000 on, slower on the contraction
  // lengths of the benchmark networks, 256 and 768; read per launch so that a test can switch it)
  const char* const shape_env = getenv("CLX_SP_MFMA");
  if (shape_env != nullptr && atoi(shape_env) == 16) CLX_LAUNCH_TIMED(gemm_sp16_kernel, dim3(p.nbm * p.nbn, batch), dim3(512), st, e0, e1, p);
  else if (small_tiles) CLX_LAUNCH_TIMED(gemm_sp2_kernel, dim3(cdiv(M, S2_BM) * (N / S2_BN), batch), dim3(256), st, e0, e1, p);
  else CLX_LAUNCH_TIMED(gemm_sp_kernel<0>, dim3(p.nbm * p.nbn, batch), dim3(512), st, e0, e1, p);
  return CLX_OK;
}

// dW[b][n][c] += sum_pixels dY[b][pixel][n] x[b][pixel][c] from the planes of dY ([rows][N]) and x ([rows][C]); `batch` problems
int clx_sp_wgrad_launch(const void* dy_planes, const void* x_planes, long long rows, int N, int C, int batch, long long bs_dy,
                        long long bs_x, long long bs_out, float* dw, int ld_dw, hipStream_t st) {
  CLX_REQUIRE(rows > 0 && N > 0 && C > 0 && N % 128 == 0 && C % 128 == 0, "clx_wgrad_planes: needs N %% 128 == 0 and C %% 128 == 0");
  SpP p = {};
  p.A = (const char*)dy_planes; p.B = (const char*)x_planes;
  p.bs_a = bs_dy; p.bs_b = bs_x; p.bs_out = bs_out;
  p.out = dw; p.ld_out = ld_dw;
  p.M = N; p.N = C;
  p.stride_a = (unsigned int)(N / 16) * KSTEP; p.stride_b = (unsigned int)(C / 16) * KSTEP;
  CLX_REQUIRE(sp::planes_bytes(rows, N) < (1ll << 32) && sp::planes_bytes(rows, C) < (1ll << 32), "clx_wgrad_planes: operand planes beyond 4 GB");
  p.total_steps = (int)(sp::padded_rows(rows) / 16);
  p.nbm = cdiv(N, SP_BM); p.nbn = C / SP_BN;
  const int tiles = p.nbm * p.nbn * batch;
  // pixel slices: the grid that costs the fewest rounds of co-resident blocks (one per CU), a block's prologue + atomics
  // priced as 12 steps
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int best_ns = 1;
  double best = 1e30;
  const int max_ns = p.total_steps / 8 > 0 ? p.total_steps / 8 : 1;
  for (int ns = 1; ns <= max_ns && ns <= 4096; ++ns) {
    const int sps = (cdiv(p.total_steps, ns) + 3) / 4 * 4;
    const int real_ns = cdiv(p.total_steps, sps);
    if (p.total_steps - (real_ns - 1) * sps < 8) continue;              // the last slice keeps two periods
    const double cost = (double)cdiv((long long)tiles * real_ns, cus) * (sps + 12);
    if (cost < best) { best = cost; best_ns = ns; }
    if ((long long)tiles * ns > 16ll * cus) break;
  }
  p.steps_per_slice = (cdiv(p.total_steps, best_ns) + 3) / 4 * 4;
  p.nslices = cdiv(p.total_steps, p.steps_per_slice);
  CLX_REQUIRE(p.total_steps >= 8 && p.total_steps - (p.nslices - 1) * p.steps_per_slice >= 8, "clx_wgrad_planes: internal: a slice of fewer than 8 steps");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (clx_prof_enabled()) clx_prof_events(CLX_PROF_WGRAD_SP, 2.0 * rows * N * C * batch, &e0, &e1);
  CLX_LAUNCH_TIMED(gemm_sp_kernel<1>, dim3(p.nbm * p.nbn * p.nslices * batch), dim3(512), st, e0, e1, p);
  return CLX_OK;
}

extern "C" int clx_wgrad_planes(const void* dy_planes, const void* x_planes, long long rows, int N, int C, float* dw, int ld_dw,
                                clx_stream stream) {
  CLX_REQUIRE(dy_planes && x_planes && dw && ld_dw >= C, "clx_wgrad_planes: null pointer / ld_dw < C");
  const int rc = clx_sp_wgrad_launch(dy_planes, x_planes, rows, N, C, 1, 0, 0, 0, dw, ld_dw, (hipStream_t)stream);
  if (rc) return rc;
  CLX_CHECK_LAUNCH("clx_wgrad_planes");
  return CLX_OK;
}

extern "C" int clx_gemm_planes(const void* a_planes, const void* b_planes, int M, int N, int K, const float* bias, int relu,
                               float* out, int ld_out, clx_stream stream) {
  CLX_REQUIRE(a_planes && b_planes && out, "clx_gemm_planes: null pointer");
  CLX_REQUIRE(ld_out >= N && ld_out % 4 == 0 && ((uintptr_t)out & 15) == 0, "clx_gemm_planes: out must be 16-byte aligned, ld_out %% 4 == 0");
  clx_conv_desc ep = {};
  ep.out = out; ep.ld_out = ld_out; ep.bias = bias; ep.relu = relu;
  const int rc = clx_sp_launch(a_planes, b_planes, M, N, K, M, 1, 0, 0, 0, &ep, (hipStream_t)stream);
  if (rc) return rc;
  CLX_CHECK_LAUNCH("clx_gemm_planes");
  return CLX_OK;
}
