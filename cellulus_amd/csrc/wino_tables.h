// Winograd transform matrices shared by wino.hip (transform kernels + batched GEMMs) and wino_fused.hip
// (one kernel per layer: transforms in LDS, all a^2 products in registers).
#pragma once

namespace {

// Transform matrices of F(MT x MT, 3 x 3) (Cook-Toom, y = A^T [(G g) o (B^T d)]), A = MT + 2.
//   MT = 2: points {0, 1, -1, inf}
//   MT = 4: points {0, 1, -1, 1/2, -2, inf} — measured on a 768-channel layer in f32 (max abs
//           error / max|y|): 4.7e-6, against 1.1e-5 for the textbook {0, +-1, +-2}, 7e-7 for
//           MT = 2 and 3.5e-7 for the direct convolution (tools/wino_numerics.py).
// All entries of A^T and B^T are dyadic, i.e. exact in f32; G is applied in double.
template <int MT, int R> struct WT;
template <> struct WT<2, 3> {
  static constexpr int A = 4;
  static constexpr float BT[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
  static constexpr float AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
  static constexpr double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
};
template <> struct WT<4, 3> {
  static constexpr int A = 6;
  static constexpr float BT[6][6] = {{1, -1.5f, -2, 1.5f, 1, 0},  {0, -1, 0.5f, 2.5f, 1, 0}, {0, 1, -2.5f, 0.5f, 1, 0},
                                     {0, -2, -1, 2, 1, 0},        {0, 0.5f, -1, -0.5f, 1, 0}, {0, 1, -1.5f, -2, 1.5f, 1}};
  static constexpr float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 0.5f, -2, 0}, {0, 1, 1, 0.25f, 4, 0}, {0, 1, -1, 0.125f, -8, 1}};
  static constexpr double G[6][3] = {{1, 0, 0},
                                     {1.0 / 3, 1.0 / 3, 1.0 / 3},
                                     {-1.0 / 3, 1.0 / 3, -1.0 / 3},
                                     {-16.0 / 15, -8.0 / 15, -4.0 / 15},
                                     {1.0 / 15, -2.0 / 15, 4.0 / 15},
                                     {0, 0, 1}};
};

// F(4x4, 2x2) — the 2x2 convolution over the low-resolution tensor in the sub-pixel form of the
// upsample convolution (plan.py): points {0, 1, -1, 1/2, inf}, 25 multiplications per 4x4 outputs
// instead of 64; every |A^T| entry <= 1.
template <> struct WT<4, 2> {
  static constexpr int A = 5;
  static constexpr float BT[5][5] = {{0.5f, -1, -0.5f, 1, 0}, {0, -0.5f, 0.5f, 1, 0}, {0, 0.5f, -1.5f, 1, 0},
                                     {0, -1, 0, 1, 0},        {0, 0.5f, -1, -0.5f, 1}};
  static constexpr float AT[4][5] = {{1, 1, 1, 1, 0}, {0, 1, -1, 0.5f, 0}, {0, 1, 1, 0.25f, 0}, {0, 1, -1, 0.125f, 1}};
  static constexpr double G[5][2] = {{2, 0}, {1, 1}, {-1.0 / 3, 1.0 / 3}, {-8.0 / 3, -4.0 / 3}, {0, 1}};
};

// acc (+)= coef * v with the coefficient known at compile time: zeros vanish, +-1 become add/sub
template <typename V>
__device__ __forceinline__ void axpy(V& acc, bool& first, float coef, const V& v) {
  if (coef == 0.f) return;
  if (first) { acc = (coef == 1.f) ? v : (coef == -1.f) ? -v : coef * v; first = false; }
  else if (coef == 1.f) acc += v;
  else if (coef == -1.f) acc -= v;
  else acc += coef * v;
}

// Weight layout of the fused kernel (wino_fused.hip; CLX_PACK_WINO4_FUSED): U[xi][n][c] = (G g G^T)[xi] stored as the
// MFMA B fragments the kernel's waves load straight into registers, one contiguous 1-KB piece per wave instruction:
//   [n / 64][c / 8][xi][(n % 64) / 32][lane = ((c % 8) / 4) * 32 + n % 32][c % 4]
__host__ __device__ inline long long wino_fused_index(int xi, int n, int c, int C, int nxi) {
  const int nb = n >> 6, nh = (n >> 5) & 1, i = n & 31, ch = c >> 3, h = (c >> 2) & 1, e = c & 3;
  return ((((long long)nb * (C >> 3) + ch) * nxi + xi) * 2 + nh) * 256 + (h * 32 + i) * 4 + e;
}

}  // namespace
