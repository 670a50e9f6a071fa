// Split-precision products: float32 GEMMs on the bf16 matrix cores, fed from PRE-SPLIT operand planes.
//
// gfx950 has no TF32 and its f32 MFMA peak (157 TFLOP/s) is 1/16 of the bf16 one.  Every f32 operand element is split
// EXACTLY into three bf16 pieces x = h0 + h1 + h2 (8 + 8 + 8 significand bits, by truncation) and the six products
// a_i b_j with i + j <= 2 are accumulated in the f32 accumulators of v_mfma_f32_32x32x16_bf16: each product is exact, the
// dropped terms (a1 b2, a2 b1, a2 b2) are <= 2^-24 relative — the size of one f32 rounding.  Peak of this form: bf16 MFMA /
// 6 = 419 TFLOP/s f32-equivalent.
//
// Rounds 2-4 split the operands INSIDE the GEMM (gemm_x3.hip, removed in round 5): ~22 VALU instructions per four
// elements on the issue port the MFMAs use, register staging, two barriers per 32-deep chunk — 168 TFLOP/s.  Here the
// split happens ONCE where a tensor is produced (clx_split_planes, the Winograd transforms, the weight packing), into
// the "P3" plane format below, and the K loop is nothing but LDS-DMA (global_load_lds_dwordx4), ds_read_b128 and MFMAs.
//
// P3 format of an [R rows][K] operand (K % 16 == 0; rows padded to a multiple of 32, the padding rows ZERO):
//   1-KB fragments in the MFMA's operand order — fragment (rb, ks, p) = plane p of rows 32 rb .. + 31, k = 16 ks .. + 15,
//   at byte ((rb * K/16 + ks) * 3 + p) * 1024; inside it lane l = 32 h + r of the wavefront owns the 16 bytes
//   x_p[32 rb + r][16 ks + 8 h .. + 7].  One global_load_lds_dwordx4 per fragment moves 1 KB of CONTIGUOUS memory
//   into LDS in exactly the order ds_read_b128 hands it to the matrix core (lane-linear: no bank conflicts, no swizzle).
//   6 bytes per element.
//
// gemm_sp_kernel: out[m][n] = epilogue( sum_k A[m][k] B[n][k] ), 256 x 128 tile, 512 threads = 8 waves (4 x 2) of
// 64 x 64, K walked in 16-deep steps through a ring of FOUR 36-KB stages (three steps in flight, one barrier per step,
// counted vmcnt).  Two-level summation as in conv_igemm.hip — fresh accumulators every 64 products, added into a second
// set — with the sign of A alternating between periods: the matrix core adds the 16 products of an instruction to the
// accumulator with a floor-like truncation (a bias of ~1e-7 of the output's rms with the same sign everywhere, which
// is common-mode over the 5e5 pixels a weight gradient sums over; docs/HISTORY.md 6b), and -(A B) carries the same
// expected bias as +(A B), so the difference of the two kinds of period has none.
//
// Replaces nn.Conv{2,3}d 1x1 (+ReLU) and the transform-domain products of the 3x3 layers, forward and data gradient
// (cellulus/models/unet.py:24-63, cellulus/train.py:178) when clx_conv_desc.precision = CLX_PREC_F32X3BF16.
#include "clx_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int FRAG = 1024;                 // bytes of one fragment
constexpr int KSTEP = 3 * FRAG;            // the three planes of one (row block, k step)
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

// x = h0 + h1 + h2 exactly, each h_i with <= 8 significant bits (the top half of an f32 word); four elements at a time,
// packed two per word (element 0 in the low half)
__device__ __forceinline__ void split4(const f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
  unsigned int u[4], a1[4], a2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    u[e] = __float_as_uint(v[e]);
    const float r1 = v[e] - __uint_as_float(u[e] & 0xffff0000u);
    a1[e] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(a1[e] & 0xffff0000u);
    a2[e] = __float_as_uint(r2);
  }
  p0[0] = __builtin_amdgcn_perm(u[1], u[0], 0x07060302u);
  p0[1] = __builtin_amdgcn_perm(u[3], u[2], 0x07060302u);
  p1[0] = __builtin_amdgcn_perm(a1[1], a1[0], 0x07060302u);
  p1[1] = __builtin_amdgcn_perm(a1[3], a1[2], 0x07060302u);
  p2[0] = __builtin_amdgcn_perm(a2[1], a2[0], 0x07060302u);
  p2[1] = __builtin_amdgcn_perm(a2[3], a2[2], 0x07060302u);
}

// f32 [rows][ld] -> P3 planes of its columns [0, K).  A wavefront writes whole fragments (three contiguous 1-KB
// stores); rows in [rows, 32 ceil(rows / 32)) are written as zeros.
__global__ __launch_bounds__(256) void sp_split_kernel(const float* __restrict__ x, long long ld, long long rows, int ksteps,
                                                       char* __restrict__ out, long long nfrag) {
  const int lane = threadIdx.x & 63;
  const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
  const int r = lane & 31, h = lane >> 5;
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
    u32x2 a0, a1, a2, b0, b1, b2;
    split4(v0, a0, a1, a2);
    split4(v1, b0, b1, b2);
    char* dst = out + f * KSTEP + lane * 16;
    *reinterpret_cast<u32x4*>(dst) = u32x4{a0[0], a0[1], b0[0], b0[1]};
    *reinterpret_cast<u32x4*>(dst + FRAG) = u32x4{a1[0], a1[1], b1[0], b1[1]};
    *reinterpret_cast<u32x4*>(dst + 2 * FRAG) = u32x4{a2[0], a2[1], b2[0], b2[1]};
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

struct SpP {
  const char* A;                      // P3 planes of the [M][K] operand (rows = output pixels)
  const char* B;                      // P3 planes of the [N][K] operand (rows = output channels)
  long long bs_a, bs_b, bs_out;       // batch strides (gridDim.y problems): bytes, bytes, floats
  float* out;
  int M, N, ksteps;                   // ksteps = K / 16, a multiple of 4
  int rb_a;                           // 32-row blocks A holds (row blocks past it are clamped: their results are never stored)
  const float* bias;
  const float* mask;
  const unsigned int* mask_bits;
  unsigned int* gate_out;
  const float* zeros;
  int relu, accumulate, ld_out, ld_mask, ld_mask_bits, ld_gate;
  int nbm, nbn;
};

__global__ __launch_bounds__(512, 1) void gemm_sp_kernel(const SpP p) {
  __shared__ __attribute__((aligned(16))) char smem[RING * STAGE];       // 144 KB; the C tile (132 KB) in the epilogue
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int v = xcd_remap(blockIdx.x, p.nbm * p.nbn);
  const int tile_n = v % p.nbn, tile_m = v / p.nbn;
  const int m0 = tile_m * SP_BM, n0 = tile_n * SP_BN;
  const int ksteps = p.ksteps;

  // ---- this wave's share of a stage's 36 fragments: f = w + 8 q, q = 0..3, and a fifth one, 32 + (w & 3), for waves 0-3
  // in even steps and waves 4-7 in odd steps: any two consecutive steps are NINE loads for every wave, so the counted
  // wait in front of a step is the same instruction for all of them.  A fragment's address is wave-uniform up to
  // 16 * lane: scalar bases, one 32-bit vector offset.
  const char* const Ab = p.A + blockIdx.y * p.bs_a;
  const char* const Bb = p.B + blockIdx.y * p.bs_b;
  auto frag_base = [&](int f) -> const char* {
    if (f < A_FRAGS) {
      int rb = tile_m * (SP_BM / 32) + f / 3;
      if (rb > p.rb_a - 1) rb = p.rb_a - 1;
      return Ab + ((long long)rb * ksteps * 3 + f % 3) * FRAG;
    }
    const int g = f - A_FRAGS;
    const int nb = tile_n * (SP_BN / 32) + g / 3;
    return Bb + ((long long)nb * ksteps * 3 + g % 3) * FRAG;
  };
  const char* gsrc[5];
#pragma unroll
  for (int q = 0; q < 4; ++q) gsrc[q] = frag_base(w + 8 * q);
  gsrc[4] = frag_base(32 + (w & 3));
  const unsigned int lane16 = (unsigned int)lane * 16u;
  const bool low_half = w < 4;
  auto issue = [&](int t, int slot, bool odd) {
    const unsigned int voff = lane16 + (unsigned int)t * KSTEP;        // (K * 192 bytes per row block: far below 4 GB)
    char* const dst = smem + slot * STAGE;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(gsrc[q] + (size_t)voff, dst + (w + 8 * q) * FRAG);
    if (low_half != odd) glds16(gsrc[4] + (size_t)voff, dst + (32 + (w & 3)) * FRAG);
  };
  // own loads of the step about to be multiplied have landed when at most the two steps behind it are in flight
  auto wait_two = [&]() { asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); };
  auto wait_one = [&]() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); };

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

  const char* const afr = smem + (wm * 2 * 3) * FRAG + lane * 16;
  const char* const bfr = smem + (A_FRAGS + wn * 2 * 3) * FRAG + lane * 16;

  // one 16-deep step out of ring slot `slot`; NEG: the A fragments negated
  auto step = [&](int slot, auto neg_tag) {
    constexpr bool NEG = decltype(neg_tag)::value;
    u32x4 a[2][3], b[2][3];
    const char* const as = afr + slot * STAGE;
    const char* const bs = bfr + slot * STAGE;
#pragma unroll
    for (int q = 0; q < 3; ++q) a[0][q] = *reinterpret_cast<const u32x4*>(as + q * FRAG);
#pragma unroll
    for (int q = 0; q < 3; ++q) b[0][q] = *reinterpret_cast<const u32x4*>(bs + q * FRAG);
#pragma unroll
    for (int q = 0; q < 3; ++q) b[1][q] = *reinterpret_cast<const u32x4*>(bs + (3 + q) * FRAG);
#pragma unroll
    for (int q = 0; q < 3; ++q) a[1][q] = *reinterpret_cast<const u32x4*>(as + (3 + q) * FRAG);
    if constexpr (NEG) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[i][q] ^= 0x80008000u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        // smallest terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][2]), __builtin_bit_cast(bf16x8, b[j][0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][0]), __builtin_bit_cast(bf16x8, b[j][2]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][1]), __builtin_bit_cast(bf16x8, b[j][1]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][1]), __builtin_bit_cast(bf16x8, b[j][0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][0]), __builtin_bit_cast(bf16x8, b[j][1]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][0]), __builtin_bit_cast(bf16x8, b[j][0]), c, 0, 0, 0);
        acc[i][j] = c;
      }
  };
  // end of a period of four steps: the period's sums go into `tot` with the period's sign, `acc` restarts from zero
  // In place, spelled as instructions (conv_igemm.hip found the same: written as `tot += acc; acc = 0` the compiler starts the
  // next period's products in a third register set).  The MFMAs that wrote `acc` were issued at least a barrier ago; the
  // s_nops cover the 11 wait states an 8-pass MFMA result needs before a VALU read, which nobody inserts for an asm block.
  auto flush = [&](auto neg_tag) {
    constexpr bool NEG = decltype(neg_tag)::value;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7\n\ts_nop 4" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          f32x2 t = {tot[i][j][r], tot[i][j][r + 1]}, x = {acc[i][j][r], acc[i][j][r + 1]};
          if constexpr (NEG) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\tv_mov_b64 %1, 0" : "+v"(t), "+v"(x));
          else asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_mov_b64 %1, 0" : "+v"(t), "+v"(x));
          tot[i][j][r] = t[0]; tot[i][j][r + 1] = t[1];
          acc[i][j][r] = x[0]; acc[i][j][r + 1] = x[1];
        }
    __builtin_amdgcn_sched_barrier(0);
  };
  // a period in the middle of the K loop: every step requests the step three ahead of it into the slot the previous
  // step has just left (all waves are past that step's reads once they have met at this step's barrier)
  auto period_mid = [&](int t0, auto neg_tag) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      wait_two();
      __builtin_amdgcn_s_barrier();
      issue(t0 + s + 3, (s + 3) & 3, ((s + 3) & 1) != 0);
      __builtin_amdgcn_sched_barrier(0);
      step(s, neg_tag);
      __builtin_amdgcn_sched_barrier(0);     // the step's LDS reads stay in front of the next step's barrier
    }
    flush(neg_tag);
  };
  // the last period: only its first step has anything left to request
  auto period_last = [&](int t0, auto neg_tag) {
    wait_two();
    __builtin_amdgcn_s_barrier();
    issue(t0 + 3, 3, true);
    __builtin_amdgcn_sched_barrier(0);
    step(0, neg_tag);
    __builtin_amdgcn_sched_barrier(0);
    wait_two();
    __builtin_amdgcn_s_barrier();
    step(1, neg_tag);
    __builtin_amdgcn_sched_barrier(0);
    wait_one();
    __builtin_amdgcn_s_barrier();
    step(2, neg_tag);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    step(3, neg_tag);
    flush(neg_tag);
  };

  issue(0, 0, false);
  issue(1, 1, true);
  issue(2, 2, false);
  const int nper = ksteps >> 2;
  int per = 0;
  for (; per + 2 < nper; per += 2) {
    period_mid(4 * per, std::false_type{});
    period_mid(4 * per + 4, std::true_type{});
  }
  if (per + 2 == nper) {
    period_mid(4 * per, std::false_type{});
    period_last(4 * per + 4, std::true_type{});
  } else {
    period_last(4 * per, std::false_type{});
  }

  // ---- epilogue (conv_igemm.hip's): plain products on whole tiles store straight from the accumulators, everything else
  // goes through an LDS transpose so that every lane stores — and reads the optional operands as — 16-byte channel runs
  if (!p.bias && !p.relu && !p.accumulate && !p.mask && !p.mask_bits && !p.gate_out && m0 + SP_BM <= p.M) {
    float* const ob = p.out + blockIdx.y * p.bs_out;
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
  constexpr int F4_PER_ROW = SP_BN / 4;                 // 32
  constexpr int ITERS = SP_BM * F4_PER_ROW / 512;       // 16
  constexpr int PH = 8;
  const int c4 = (tid % F4_PER_ROW) * 4;
  const int n = n0 + c4;
  const bool n_live = n < p.N;
#pragma unroll 1
  for (int h0 = 0; h0 < ITERS; h0 += PH) {
    f32x4 val[PH];
    int mrow[PH];
    bool live[PH];
#pragma unroll
    for (int j = 0; j < PH; ++j) {
      const int row = (tid + 512 * (h0 + j)) / F4_PER_ROW;
      mrow[j] = m0 + row;
      live[j] = mrow[j] < p.M && n_live;
      val[j] = *reinterpret_cast<const f32x4*>(&Cs[row * LDC + c4]);
    }
    float* const out_base = p.out + blockIdx.y * p.bs_out + n;
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
  }
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

extern "C" size_t clx_planes_bytes(long long rows, int K) {
  if (rows <= 0 || K <= 0 || K % 16 != 0) return 0;
  return (size_t)((rows + 31) / 32) * (size_t)(K / 16) * KSTEP;
}

extern "C" int clx_split_planes(const float* x, long long ld, long long rows, int K, void* planes, clx_stream stream) {
  CLX_REQUIRE(x != nullptr && planes != nullptr, "clx_split_planes: null pointer");
  CLX_REQUIRE(rows > 0 && K > 0 && K % 16 == 0 && ld >= K && ld % 4 == 0, "clx_split_planes: K must be a multiple of 16, ld >= K a multiple of 4");
  CLX_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 15) == 0, "clx_split_planes: pointers must be 16-byte aligned");
  const long long nfrag = (rows + 31) / 32 * (K / 16);
  long long blocks = (nfrag + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  CLX_LAUNCH_KIND(CLX_PROF_SPLIT_PLANES, sp_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ld, rows, K / 16,
                  (char*)planes, nfrag);
  CLX_CHECK_LAUNCH("clx_split_planes");
  return CLX_OK;
}

extern "C" int clx_join_planes(const void* planes, long long rows, int K, float* x, long long ld, clx_stream stream) {
  CLX_REQUIRE(x != nullptr && planes != nullptr, "clx_join_planes: null pointer");
  CLX_REQUIRE(rows > 0 && K > 0 && K % 16 == 0 && ld >= K, "clx_join_planes: K must be a multiple of 16, ld >= K");
  const long long nfrag = (rows + 31) / 32 * (K / 16);
  long long blocks = (nfrag + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  sp_join_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>((const char*)planes, ld, rows, K / 16, x, nfrag);
  CLX_CHECK_LAUNCH("clx_join_planes");
  return CLX_OK;
}

// the batched product behind clx_gemm_planes and the precision switch of clx_conv_fwd: `batch` problems, operand b at
// A + b * bs_a / B + b * bs_b (bytes), result at out + b * bs_out (floats)
int clx_sp_launch(const void* A, const void* B, int M, int N, int K, long long rows_a, int batch, long long bs_a, long long bs_b,
                  long long bs_out, const clx_conv_desc* ep, hipStream_t st) {
  CLX_REQUIRE(M > 0 && N > 0 && N % SP_BN == 0 && K >= 64 && K % 64 == 0, "clx_gemm_planes: needs N %% 128 == 0 and K %% 64 == 0");
  CLX_REQUIRE(rows_a >= M, "clx_gemm_planes: the A planes hold fewer rows than M");
  SpP p;
  p.A = (const char*)A; p.B = (const char*)B;
  p.bs_a = bs_a; p.bs_b = bs_b; p.bs_out = bs_out;
  p.out = ep->out; p.ld_out = ep->ld_out;
  p.M = M; p.N = N; p.ksteps = K / 16;
  p.rb_a = (int)((rows_a + 31) / 32);
  p.bias = ep->bias; p.mask = ep->mask; p.mask_bits = ep->mask_bits; p.gate_out = ep->gate_out;
  p.relu = ep->relu; p.accumulate = ep->accumulate; p.ld_mask = ep->ld_mask; p.ld_mask_bits = ep->ld_mask_bits; p.ld_gate = ep->ld_gate;
  p.zeros = sp_zero_buffer();
  CLX_REQUIRE(p.zeros != nullptr, "clx_gemm_planes: cannot resolve the device zero buffer");
  p.nbm = cdiv(M, SP_BM); p.nbn = N / SP_BN;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (clx_prof_enabled()) clx_prof_events(CLX_PROF_GEMM_SP, 2.0 * M * N * K * batch, &e0, &e1);
  CLX_LAUNCH_TIMED(gemm_sp_kernel, dim3(p.nbm * p.nbn, batch), dim3(512), st, e0, e1, p);
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
