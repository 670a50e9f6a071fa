// "P3" operand planes of the split-precision products (gemm_sp.hip; format: include/clx.h): helpers shared by the
// kernels that WRITE planes (the split pass, the Winograd transforms, the product epilogue).
#pragma once
#include "clx_common.h"

namespace sp {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int FRAG = 1024;          // bytes of one fragment: 32 rows x 16 k x bf16
constexpr int KSTEP = 3 * FRAG;     // the three pieces of one (32-row block, 16-k step)

// rows the planes of an [rows][K] operand hold: padded to a multiple of 64, at least 128 (the weight gradient walks the
// pixels in periods of four 16-pixel steps, at least two of them); the padding rows are zero
__host__ __device__ inline long long padded_rows(long long rows) { return rows <= 128 ? 128 : (rows + 63) / 64 * 64; }
// bytes of those planes
__host__ __device__ inline long long planes_bytes(long long rows, int K) { return padded_rows(rows) / 32 * (long long)(K / 16) * KSTEP; }

// byte offset of the 16 bytes x_0[row][8 * octet .. + 7] (piece 0; pieces 1, 2 follow at + FRAG, + 2 FRAG)
__host__ __device__ inline long long piece_offset(long long row, int octet, int ksteps) {
  return ((row >> 5) * ksteps + (octet >> 1)) * (long long)KSTEP + (octet & 1) * 512 + (row & 31) * 16;
}

// x = h0 + h1 + h2 exactly, each h_i a bfloat16: h0 = rn(x), h1 = rn(x - h0), h2 = x - h0 - h1 (<= 8 significant bits left:
// exact).  Round-to-nearest pieces (v_cvt_pk_bf16_f32) keep |h1| <= 2^-9 |x| and |h2| <= 2^-17 |x| with either sign, so the three
// products the kernels drop (h1 g2, h2 g1, h2 g2: <= 2^-25 of the product) are smaller than with truncated pieces and have
// no preferred sign (truncation left a relative bias of -8e-9 on the products).  Four elements at a time, packed two per
// word (element 0 in the low half).
__device__ __forceinline__ void split4(const f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  auto pack = [](float a, float b) -> unsigned int {
    const bf16x2_ h = __builtin_convertvector(f32x2_{a, b}, bf16x2_);
    return __builtin_bit_cast(unsigned int, h);
  };
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float x0 = v[2 * e], x1 = v[2 * e + 1];
    const unsigned int w0 = pack(x0, x1);
    const float r0 = x0 - __uint_as_float(w0 << 16), r1 = x1 - __uint_as_float(w0 & 0xffff0000u);
    const unsigned int w1 = pack(r0, r1);
    const float s0 = r0 - __uint_as_float(w1 << 16), s1 = r1 - __uint_as_float(w1 & 0xffff0000u);
    p0[e] = w0;
    p1[e] = w1;
    p2[e] = pack(s0, s1);
  }
}

// the three pieces of x[row][c .. c + 3] (c % 4 == 0) into the planes at `base`: three 8-byte stores
__device__ __forceinline__ void store4(char* base, long long row, int c, int ksteps, const f32x4 v) {
  u32x2 p0, p1, p2;
  split4(v, p0, p1, p2);
  char* dst = base + piece_offset(row, c >> 3, ksteps) + ((c >> 2) & 1) * 8;
  *reinterpret_cast<u32x2*>(dst) = p0;
  *reinterpret_cast<u32x2*>(dst + FRAG) = p1;
  *reinterpret_cast<u32x2*>(dst + 2 * FRAG) = p2;
}

// Work items of a kernel that makes planes out of (row, 4-channel group) items, C % 32 == 0: a wavefront takes 8
// consecutive rows x 32 consecutive channels, so that it READS whole 128-byte lines of a channels-last tensor and
// WRITES, per piece, four 128-byte runs (8 rows x 16 bytes of four octets).  item -> (row, c); rows are dealt out in
// groups of 8: the caller sizes its grid for 8 * ceil(rows / 8) * C / 4 items and skips row >= rows.
__device__ __forceinline__ void item_to_row_channel(long long i, int C, long long& row, int& c) {
  const int lane = (int)(i & 63);
  const long long wv = i >> 6;
  const int nblk = C >> 5;
  const int cblk = (int)(wv % nblk);
  row = (wv / nblk) * 8 + (lane >> 3);
  c = cblk * 32 + (lane & 7) * 4;
}

}  // namespace sp

// zero the padding rows [rows, padded_rows(rows)) of `batch` plane sets (stride bs bytes) of an [rows][K] operand
int clx_sp_zero_tail(void* planes, long long rows, int K, int batch, long long bs, hipStream_t st);
