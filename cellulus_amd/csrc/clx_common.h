// Shared helpers for the libclx HIP sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/clx.h"

void clx_set_error(const char* fmt, ...);

#define CLX_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      clx_set_error(__VA_ARGS__);             \
      return CLX_ERR_ARG;                     \
    }                                         \
  } while (0)

#define CLX_CHECK_LAUNCH(name)                                           \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      clx_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return CLX_ERR_LAUNCH;                                             \
    }                                                                    \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// n / d for 0 <= n < 2^31 with a host-prepared multiplier (Granlund–Montgomery
// round-up form): q = (umulhi(m, n) + n) >> l.
struct FastDiv {
  uint32_t m, l, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.l = l;
  f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return (__umulhi(f.m, n) + n) >> f.l;
}

// Bijective XCD-aware block remap: blocks b and b+8 share an XCD (round-robin
// dispatch), so give every XCD a contiguous range of virtual ids.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// small-channel (first layer) convolution path, conv_smallc.hip
bool clx_smallc_applicable(const clx_conv_desc* d);
int clx_smallc_fwd(const clx_conv_desc* d, hipStream_t st);
int clx_smallc_wgrad(const clx_conv_desc* d, const float* dy, int ld_dy, float* dwpack,
                     float* dbias, hipStream_t st);

// batched launches of the MFMA kernels (used by the Winograd path, wino.hip)
int clx_igemm_launch(const clx_conv_desc* d, int batch, long long bs_in, long long bs_w,
                     long long bs_out, hipStream_t st);
int clx_wgrad_launch(const clx_conv_desc* d, const float* dy, int ld_dy, float* dwpack, float* dbias,
                     int batch, long long bs_x, long long bs_dy, long long bs_out, hipStream_t st);
// split-precision products (gemm_sp.hip): true if the plain product `d` describes — one source read pixel by pixel, 1x1x1
// kernel — is one gemm_sp_kernel covers (precision switch set, N % 128 == 0, K % 64 == 0, K >= 128) and has its weight planes
bool clx_sp_applicable(const clx_conv_desc* d);
// `batch` products out[b] = epilogue(A[b] B[b]^T) from P3 planes (strides: bytes, bytes, floats); the epilogue fields of
// `ep` (out, ld_out, bias, relu, accumulate, mask, mask_bits, gate_out and their strides) are honoured
int clx_sp_launch(const void* A, const void* B, int M, int N, int K, long long rows_a, int batch, long long bs_a, long long bs_b,
                  long long bs_out, const clx_conv_desc* ep, hipStream_t st);
// shader-clock / wall-clock ticks recorded by the split-precision product kernels (added by clx_profile_clock)
int clx_sp_clock_read(double* shader_ticks, double* wall_ticks, int reset);
// planes <- split(x[rows][ld], columns [0, K)); colsum != NULL: colsum[k] += the column sums of x, k < nreal
int clx_sp_split(const float* x, long long ld, long long rows, int K, void* planes, float* colsum, int nreal, hipStream_t st);
// dW[b][n][c] += sum_rows dY[b][row][n] x[b][row][c] from planes (strides: bytes, bytes, floats); N, C multiples of 128
int clx_sp_wgrad_launch(const void* dy_planes, const void* x_planes, long long rows, int N, int C, int batch, long long bs_dy,
                        long long bs_x, long long bs_out, float* dw, int ld_dw, hipStream_t st);
// Winograd F(2x2, 3x3) / F(4x4, 3x3) path (wino.hip)
int clx_wino_fwd(const clx_conv_desc* d, hipStream_t st);
int clx_wino_wgrad(const clx_conv_desc* d, const float* dy, int ld_dy, float* dwpack, float* dbias,
                   hipStream_t st);
int clx_wino_pack(const float* w, float* wp, int cout, int cin, int cin_pad, int cout_pad, int dgrad, int tile,
                  int ksize, int kd, hipStream_t st);

// Winograd F(4x4) with the transforms inside the product kernel (wino_fused.hip)
int clx_wino_fused_fwd(const clx_conv_desc* d, hipStream_t st);

// in-library kernel timing (clx_core.hip); kinds match enum clx_profile_kind in clx.h
bool clx_prof_enabled();
void clx_prof_events(int kind, double flops, hipEvent_t* e0, hipEvent_t* e1);
// kernel launch with the optional event pair of clx_prof_events (null events: a plain launch)
#define CLX_LAUNCH_TIMED(kernel, grid, block, st, e0, e1, ...) \
  hipExtLaunchKernelGGL(kernel, grid, block, 0, st, e0, e1, 0, __VA_ARGS__)
// a launch of profile kind `kind` (the HBM-bound kernels: no FLOPs) with `lds` bytes of dynamic LDS
#define CLX_LAUNCH_KIND(kind, kernel, grid, block, lds, st, ...)                              \
  do {                                                                                        \
    hipEvent_t e0__ = nullptr, e1__ = nullptr;                                                \
    if (clx_prof_enabled()) clx_prof_events(kind, 0.0, &e0__, &e1__);                         \
    hipExtLaunchKernelGGL(kernel, grid, block, lds, st, e0__, e1__, 0, __VA_ARGS__);          \
  } while (0)
