// OPT-IN precision "f32x3bf16" (clx_conv_desc.precision = CLX_PREC_F32X3BF16): the plain GEMMs of the
// network — 1x1 layers and the batched GEMMs of the Winograd layers, out[m][n] = sum_k A[m][k] B[n][k] —
// on the bf16 matrix cores without giving up float32 results.  gfx950 has no TF32 and its f32 MFMA
// peak (157 TFLOP/s) is 1/16 of the bf16 one; every f32 operand is split EXACTLY into three bf16
// pieces x = h0 + h1 + h2 (8 + 8 + 8 significand bits, by truncation) and the six products
// a_i b_j with i + j <= 2 are accumulated in the f32 accumulators of v_mfma_f32_32x32x16_bf16: each
// product is exact, the dropped terms are <= 2^-24 relative — the size of one f32 rounding.
// The matrix core adds the 16 products of one instruction to the accumulator with a floor-like
// truncation (measured: a mean error of about -1e-7 rms at K = 2304, independent of the sign of the
// result, where the f32 MFMA has none).  That common-mode bias is harmless in one GEMM and adds up
// coherently over the pixels of a weight gradient, so it is cancelled: the k-steps alternate between
// two accumulators, the second one fed with -A, and the result is their difference — both carry the
// same expected bias.
// Peak for this form: bf16 MFMA / 6 = 419 TFLOP/s f32-equivalent.  NOT the default: the headline
// path is the true-f32 kernel of conv_igemm.hip; this one has its own bench object and the same
// parity tests.
//
// 128x128 tile, 256 threads (2x2 waves of 64x64), K chunk 32: global f32 loads -> split in registers ->
// three bf16 planes per operand in LDS (row stride 40 halfwords: conflict-free 16-byte fragment reads)
// -> 2 k-steps x 4 tiles x 6 MFMAs.  Epilogue as in conv_igemm.hip (bias, ReLU, accumulate, gate
// bits / masks, 16-byte stores through an LDS transpose).
#include "clx_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int XBM = 128, XBN = 128, XBK = 32;
constexpr int XRS = 40;                       // LDS row stride in bf16 (80 B)
constexpr int XPIECE = XBM * XRS;             // bf16 elements per piece plane
constexpr int XLDC = XBN + 4;

struct X3P {
  const float* A; long long lda;
  const float* B;                              // [N][K]
  float* out; int ld_out;
  int M, N, K;
  const float* bias; const float* mask; const unsigned int* mask_bits; unsigned int* gate_out;
  int relu, accumulate, ld_mask, ld_mask_bits, ld_gate;
  long long bs_a, bs_b, bs_out;
  int nbm, nbn;
};

// x = h0 + h1 + h2 exactly, each h_i with <= 8 significant bits (the top half of an f32 word)
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

__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(const X3P p) {
  __shared__ __attribute__((aligned(16))) float smem[XBM * XLDC];      // 67.6 KB: C tile; A/B planes need 61.4 KB
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Bs = As + 3 * XPIECE;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int v = xcd_remap(blockIdx.x, p.nbm * p.nbn);
  const int m0 = (v / p.nbn) * XBM, n0 = (v % p.nbn) * XBN;
  // staging thread -> (row, 4 columns): the 16 contiguous lanes of a ds_write_b64 group take rows r and r + 4,
  // 4 x 80 B = 16 banks apart, so their two 64-byte runs cover the 32 banks once (rows r and r + 1 overlap on 4
  // banks: measured as SQ_LDS_BANK_CONFLICT = 1/3 of the kernel's LDS cycles)
  const int lr = (t >> 6) * 8 + ((t >> 4) & 3) + 4 * ((t >> 3) & 1), lk = (t & 7) * 4;
  const float* Ab = p.A + blockIdx.y * p.bs_a;
  const float* ap[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = min(m0 + lr + 32 * i, p.M - 1);          // tail rows re-read the last row (never stored)
    ap[i] = Ab + (size_t)m * p.lda + lk;
  }
  const float* bp = p.B + blockIdx.y * p.bs_b + (size_t)(n0 + lr) * p.K + lk;

  f32x16 acc[2][2][2];                                       // [k-step parity][i][j]; parity 1 holds -(A B)
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;
  const unsigned int aflip = (lk >= 16) ? 0x80008000u : 0u;   // columns of the second k-step: A pieces negated

  f32x4 ga[4], gb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ga[i] = *reinterpret_cast<const f32x4*>(ap[i]);
    gb[i] = *reinterpret_cast<const f32x4*>(bp + (size_t)(32 * i) * p.K);
  }
  const int nk = p.K / XBK;
  const int fr = lane & 31, fh = lane >> 5;
  for (int kc = 0; kc < nk; ++kc) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x2 p0, p1, p2;
      split4(ga[i], p0, p1, p2);
      const int off = (lr + 32 * i) * XRS + lk;
      p0[0] ^= aflip; p0[1] ^= aflip; p1[0] ^= aflip; p1[1] ^= aflip; p2[0] ^= aflip; p2[1] ^= aflip;
      *reinterpret_cast<u32x2*>(As + off) = p0;
      *reinterpret_cast<u32x2*>(As + XPIECE + off) = p1;
      *reinterpret_cast<u32x2*>(As + 2 * XPIECE + off) = p2;
      split4(gb[i], p0, p1, p2);
      *reinterpret_cast<u32x2*>(Bs + off) = p0;
      *reinterpret_cast<u32x2*>(Bs + XPIECE + off) = p1;
      *reinterpret_cast<u32x2*>(Bs + 2 * XPIECE + off) = p2;
    }
    __syncthreads();
    if (kc + 1 < nk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ga[i] = *reinterpret_cast<const f32x4*>(ap[i] + (kc + 1) * XBK);
        gb[i] = *reinterpret_cast<const f32x4*>(bp + (size_t)(32 * i) * p.K + (kc + 1) * XBK);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          a[i][q] = *reinterpret_cast<const bf16x8*>(As + q * XPIECE + (wm * 64 + i * 32 + fr) * XRS + ks * 16 + fh * 8);
          b[i][q] = *reinterpret_cast<const bf16x8*>(Bs + q * XPIECE + (wn * 64 + i * 32 + fr) * XRS + ks * 16 + fh * 8);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[ks][i][j];
          // smallest terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[ks][i][j] = c;
        }
    }
  }

  // ---- epilogue (the one of conv_igemm_kernel<128,128>): C/D map col = lane & 31,
  // row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  __syncthreads();
  float* Cs = smem;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (wn * 2 + j) * 32 + fr;
      const int n = n0 + col;
      const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        Cs[row * XLDC + col] = (acc[0][i][j][r] - acc[1][i][j][r]) + bv;
      }
    }
  __syncthreads();
  constexpr int F4_PER_ROW = XBN / 4;
  constexpr int ITERS = XBM * F4_PER_ROW / 256;
#pragma unroll 4
  for (int it = 0; it < ITERS; ++it) {
    const int idx = t + 256 * it;
    const int row = idx / F4_PER_ROW, c4 = (idx % F4_PER_ROW) * 4;
    const int m = m0 + row, n = n0 + c4;
    const bool live = m < p.M && n < p.N;
    f32x4 val = {0.f, 0.f, 0.f, 0.f};
    float* dst = p.out + blockIdx.y * p.bs_out + (size_t)m * p.ld_out + n;
    if (live) {
      val = *reinterpret_cast<const f32x4*>(&Cs[row * XLDC + c4]);
      if (p.accumulate) val += *reinterpret_cast<const f32x4*>(dst);
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
      }
      if (p.mask_bits) {
        const unsigned int w = p.mask_bits[(size_t)m * p.ld_mask_bits + (n >> 5)] >> (n & 31);
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = ((w >> e) & 1u) ? val[e] : 0.f;
      }
      if (p.mask) {
        const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + (size_t)m * p.ld_mask + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = (mk[e] > 0.f) ? val[e] : 0.f;
      }
    }
    if (p.gate_out) {
      unsigned int nib = 0u;
#pragma unroll
      for (int e = 0; e < 4; ++e) nib |= (live && val[e] > 0.f) ? (1u << e) : 0u;
      unsigned int word = nib << (4 * (lane & 7));
      word |= __shfl_xor(word, 1, 64);
      word |= __shfl_xor(word, 2, 64);
      word |= __shfl_xor(word, 4, 64);
      if ((lane & 7) == 0 && m < p.M && n < p.ld_out) p.gate_out[(size_t)m * p.ld_gate + (n >> 5)] = word;
    }
    if (live) *reinterpret_cast<f32x4*>(dst) = val;        // N % 128 == 0: whole 16-byte groups
  }
}

}  // namespace

// true if descriptor `d` (with `batch` problems of stride bs_*) is a plain [M x K] . [N x K]^T product this
// kernel covers: one source read pixel by pixel, 1x1(x1) kernel, N % 128 == 0, K % 32 == 0
bool clx_x3_applicable(const clx_conv_desc* d) {
  if (d->precision != CLX_PREC_F32X3BF16 || d->nsrc != 1) return false;
  if (d->KD != 1 || d->KH != 1 || d->KW != 1 || d->PD || d->PH || d->PW) return false;
  const clx_src& S = d->src[0];
  if (S.fz != 1 || S.fy != 1 || S.fx != 1 || S.oz || S.oy || S.ox) return false;
  if (S.D != d->ID || S.H != d->IH || S.W != d->IW) return false;
  return d->N % 128 == 0 && S.C % 32 == 0 && d->ld_out % 4 == 0;
}

int clx_x3_launch(const clx_conv_desc* d, int batch, long long bs_in, long long bs_w, long long bs_out,
                  hipStream_t st) {
  X3P p;
  const clx_src& S = d->src[0];
  p.A = S.ptr; p.lda = S.ld; p.B = d->wpack; p.out = d->out; p.ld_out = d->ld_out;
  p.M = d->B * d->ID * d->IH * d->IW; p.N = d->N; p.K = S.C;
  p.bias = d->bias; p.mask = d->mask; p.mask_bits = d->mask_bits; p.gate_out = d->gate_out;
  p.relu = d->relu; p.accumulate = d->accumulate; p.ld_mask = d->ld_mask; p.ld_mask_bits = d->ld_mask_bits;
  p.ld_gate = d->ld_gate;
  p.bs_a = bs_in; p.bs_b = bs_w; p.bs_out = bs_out;
  p.nbm = cdiv(p.M, XBM); p.nbn = p.N / XBN;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (clx_prof_enabled()) clx_prof_events(CLX_PROF_GEMM_X3, 2.0 * p.M * p.N * p.K * batch, &e0, &e1);
  CLX_LAUNCH_TIMED(gemm_x3_kernel, dim3(p.nbm * p.nbn, batch), dim3(256), st, e0, e1, p);
  return CLX_OK;
}
