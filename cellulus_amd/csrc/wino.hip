// Winograd F(2x2, 3x3) convolution for 2-D 3x3 layers (forward, data gradient, weight
// gradient) on top of the f32-MFMA GEMM kernels:
//
//   forward / dgrad:  V = B^T d B  (4x4 input patches, stride 2)      [HBM streaming]
//                     M[xi] = V[xi] . U[xi]^T, xi = 0..15             [16 batched MFMA GEMMs]
//                     Y = A^T M A (+ bias, ReLU, ReLU gate)           [HBM streaming]
//   wgrad:            dU[xi] = (A dY A^T)[xi]^T . V[xi]  over tiles   [16 batched MFMA GEMMs]
//                     dW = G^T dU G   (clx_unpack_wgrad_wino)
//
// 16 multiplications per 2x2 outputs instead of 36: 2.25x fewer MFMA FLOPs, all in f32
// (measured error 3e-6 vs 1e-6 for the direct form on a 768-channel layer).  The price is
// HBM traffic for V and M (4x the activation each), so the plan selects it only for layers
// with enough channels on both sides.
//
// Replaces the same reference calls as conv_igemm.hip / conv_wgrad.hip (nn.Conv2d 3x3 and
// its autograd backward, cellulus/models/unet.py:24-51, cellulus/train.py:178).
#include "clx_common.h"

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
  int B, SH, SW, oy, ox;   // stored grid of the source and crop offset
  int IH, IW, P;           // logical input extent, zero padding
  int OH, OW, th, tw;      // output extent, tiles per image
  long long T;             // B * th * tw
};

// V[xi][t][c] = (B^T d B)[xi],  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
__global__ void wino_input_kernel(const float* __restrict__ x, int ld_x, int C4, Geom g,
                                  float* __restrict__ V, long long total) {
  const int C = C4 * 4;
  const long long plane = g.T * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long t = i / C4;
    const int tx = (int)(t % g.tw);
    const long long q = t / g.tw;
    const int ty = (int)(q % g.th);
    const int b = (int)(q / g.th);
    f32x4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ly = 2 * ty - g.P + r;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int lx = 2 * tx - g.P + s;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ly < (unsigned)g.IH && (unsigned)lx < (unsigned)g.IW) {
          const long long pix = ((long long)b * g.SH + ly + g.oy) * g.SW + lx + g.ox;
          v = ld4(x + pix * ld_x + c);
        }
        d[r][s] = v;
      }
    }
    f32x4 w[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {      // rows: B^T d
      w[0][s] = d[0][s] - d[2][s];
      w[1][s] = d[1][s] + d[2][s];
      w[2][s] = d[2][s] - d[1][s];
      w[3][s] = d[1][s] - d[3][s];
    }
    float* dst = V + t * C + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {      // columns: (.) B
      st4(dst + (r * 4 + 0) * plane, w[r][0] - w[r][2]);
      st4(dst + (r * 4 + 1) * plane, w[r][1] + w[r][2]);
      st4(dst + (r * 4 + 2) * plane, w[r][2] - w[r][1]);
      st4(dst + (r * 4 + 3) * plane, w[r][1] - w[r][3]);
    }
  }
}

// Y = A^T m A,  A^T = [1 1 1 0; 0 1 -1 -1]; bias, ReLU, ReLU gate fused
__global__ void wino_output_kernel(const float* __restrict__ M, int N4, Geom g,
                                   const float* __restrict__ bias, int relu,
                                   const float* __restrict__ mask, int ld_mask,
                                   float* __restrict__ out, int ld_out, int Nreal, long long total) {
  const int N = N4 * 4;
  const long long plane = g.T * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N4) * 4;
    long long t = i / N4;
    const int tx = (int)(t % g.tw);
    const long long q = t / g.tw;
    const int ty = (int)(q % g.th);
    const int b = (int)(q / g.th);
    const float* src = M + t * N + n;
    f32x4 r0[4], r1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {      // rows: A^T m
      const f32x4 m0 = ld4(src + (0 * 4 + s) * plane), m1 = ld4(src + (1 * 4 + s) * plane);
      const f32x4 m2 = ld4(src + (2 * 4 + s) * plane), m3 = ld4(src + (3 * 4 + s) * plane);
      r0[s] = m0 + m1 + m2;
      r1[s] = m1 - m2 - m3;
    }
    f32x4 y[2][2];
    y[0][0] = r0[0] + r0[1] + r0[2];
    y[0][1] = r0[1] - r0[2] - r0[3];
    y[1][0] = r1[0] + r1[1] + r1[2];
    y[1][1] = r1[1] - r1[2] - r1[3];
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = (n + e < Nreal) ? bias[n + e] : 0.f;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int oy = 2 * ty + a;
      if (oy >= g.OH) continue;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int ox = 2 * tx + c;
        if (ox >= g.OW) continue;
        const long long m = ((long long)b * g.OH + oy) * g.OW + ox;
        f32x4 v = y[a][c] + bv;
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
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
      }
    }
  }
}

// Mdy[xi][t][n] = (A dy A^T)[xi],  A = [1 0; 1 1; 1 -1; 0 -1]; dbias[n] += sum of dy
// (block-private LDS accumulator, then one global atomic per channel per block)
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, int ld_dy, int N4,
                                                      Geom g, float* __restrict__ Md,
                                                      float* __restrict__ dbias, int Nreal,
                                                      long long total) {
  extern __shared__ float bacc[];
  const int N = N4 * 4;
  const long long plane = g.T * N;
  if (dbias) {
    for (int k = threadIdx.x; k < N; k += blockDim.x) bacc[k] = 0.f;
    __syncthreads();
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N4) * 4;
    long long t = i / N4;
    const int tx = (int)(t % g.tw);
    const long long q = t / g.tw;
    const int ty = (int)(q % g.th);
    const int b = (int)(q / g.th);
    f32x4 d[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int oy = 2 * ty + a, ox = 2 * tx + c;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (oy < g.OH && ox < g.OW) v = ld4(dy + (((long long)b * g.OH + oy) * g.OW + ox) * ld_dy + n);
        d[a][c] = v;
      }
    if (dbias) {
      const f32x4 sum = d[0][0] + d[0][1] + d[1][0] + d[1][1];
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&bacc[n + e], sum[e]);
    }
    f32x4 w[4][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {      // rows: A dy
      w[0][c] = d[0][c];
      w[1][c] = d[0][c] + d[1][c];
      w[2][c] = d[0][c] - d[1][c];
      w[3][c] = -d[1][c];
    }
    float* dst = Md + t * N + n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {      // columns: (.) A^T
      st4(dst + (r * 4 + 0) * plane, w[r][0]);
      st4(dst + (r * 4 + 1) * plane, w[r][0] + w[r][1]);
      st4(dst + (r * 4 + 2) * plane, w[r][0] - w[r][1]);
      st4(dst + (r * 4 + 3) * plane, -w[r][1]);
    }
  }
  if (dbias) {
    __syncthreads();
    for (int k = threadIdx.x; k < Nreal; k += blockDim.x)
      if (bacc[k] != 0.f) atomicAdd(dbias + k, bacc[k]);
  }
}

// U[xi][n][c] = (G g G^T)[xi],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
// mode FWD:   g = w[n][c][:, :]            rows n < rows_pad (cout_pad), cols c < cin_pad
// mode DGRAD: g = flip(w[n][c]) transposed roles: U[xi][c][n]
__global__ void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int cout,
                                   int cin, int rows, int cols, int dgrad, long long total) {
  const long long plane = (long long)rows * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % cols), row = (int)(i / cols);
    const int n = dgrad ? col : row, c = dgrad ? row : col;
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        float v = 0.f;
        if (n < cout && c < cin) {
          const int rr = dgrad ? 2 - r : r, ss = dgrad ? 2 - s : s;
          v = w[((long long)n * cin + c) * 9 + rr * 3 + ss];
        }
        g[r][s] = v;
      }
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      t[0][s] = g[0][s];
      t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
      t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
      t[3][s] = g[2][s];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      U[(r * 4 + 0) * plane + i] = t[r][0];
      U[(r * 4 + 1) * plane + i] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
      U[(r * 4 + 2) * plane + i] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
      U[(r * 4 + 3) * plane + i] = t[r][2];
    }
  }
}

// dw[n][c][3x3] = G^T dU G,  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
__global__ void wino_unpack_kernel(const float* __restrict__ dU, float* __restrict__ dw, int cout,
                                   int cin, int rows, int cin_pad, long long total) {
  const long long plane = (long long)rows * cin_pad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cin), n = (int)(i / cin);
    const float* src = dU + (long long)n * cin_pad + c;
    float u[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int s = 0; s < 4; ++s) u[r][s] = src[(r * 4 + s) * plane];
    float t[3][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      t[0][s] = u[0][s] + 0.5f * (u[1][s] + u[2][s]);
      t[1][s] = 0.5f * (u[1][s] - u[2][s]);
      t[2][s] = 0.5f * (u[1][s] + u[2][s]) + u[3][s];
    }
    float* dst = dw + i * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      dst[r * 3 + 0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
      dst[r * 3 + 1] = 0.5f * (t[r][1] - t[r][2]);
      dst[r * 3 + 2] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
    }
  }
}

bool applicable(const clx_conv_desc* d) {
  if (d->nsrc != 1 || d->KD != 1 || d->KH != 3 || d->KW != 3 || d->ID != 1 || d->PD != 0) return false;
  if (d->PH != d->PW || (d->PH != 0 && d->PH != 2)) return false;
  const clx_src& S = d->src[0];
  if (S.fz != 1 || S.fy != 1 || S.fx != 1 || S.D != 1 || S.oz != 0) return false;
  if (S.C % 4 != 0 || d->N <= 0) return false;
  return true;
}

Geom geom(const clx_conv_desc* d) {
  Geom g;
  const clx_src& S = d->src[0];
  g.B = d->B; g.SH = S.H; g.SW = S.W; g.oy = S.oy; g.ox = S.ox;
  g.IH = d->IH; g.IW = d->IW; g.P = d->PH;
  g.OH = d->IH + 2 * d->PH - 2; g.OW = d->IW + 2 * d->PW - 2;
  g.th = (g.OH + 1) / 2; g.tw = (g.OW + 1) / 2;
  g.T = (long long)g.B * g.th * g.tw;
  return g;
}

inline int pad4(int n) { return (n + 3) / 4 * 4; }

}  // namespace

extern "C" size_t clx_conv_workspace_bytes(const clx_conv_desc* d, int pass) {
  if (d == nullptr || !applicable(d)) return 0;
  if (pass == CLX_PASS_WGRAD && d->PH != 0) return 0;
  const Geom g = geom(d);
  if (g.OH <= 0 || g.OW <= 0 || g.T >= (1ll << 31)) return 0;
  const long long C = d->src[0].C, N = pad4(d->N);
  return (size_t)(16 * g.T * (C + N)) * sizeof(float);
}

int clx_wino_fwd(const clx_conv_desc* d, hipStream_t st) {
  CLX_REQUIRE(applicable(d), "clx_conv_fwd: CLX_ALGO_WINOGRAD does not apply to this geometry");
  const size_t need = clx_conv_workspace_bytes(d, CLX_PASS_FWD);
  CLX_REQUIRE(d->workspace != nullptr && d->workspace_bytes >= need && need > 0,
              "clx_conv_fwd: Winograd needs %zu workspace bytes (%zu given)", need, d->workspace_bytes);
  CLX_REQUIRE(((uintptr_t)d->workspace & 15) == 0, "clx_conv_fwd: workspace must be 16-byte aligned");
  const Geom g = geom(d);
  const clx_src& S = d->src[0];
  const int C = S.C, Np = pad4(d->N);
  float* V = (float*)d->workspace;
  float* M = V + 16 * g.T * C;
  const long long tot_in = g.T * (C / 4);
  wino_input_kernel<<<grid_for(tot_in, 256), 256, 0, st>>>(S.ptr, S.ld, C / 4, g, V, tot_in);
  // 16 GEMMs [T x C] . [C x N] as a batched 1x1 "convolution" over T pixels
  clx_conv_desc gd = {};
  gd.nsrc = 1;
  gd.src[0].ptr = V; gd.src[0].C = C; gd.src[0].ld = C;
  gd.src[0].D = 1; gd.src[0].H = 1; gd.src[0].W = (int)g.T;
  gd.src[0].fz = gd.src[0].fy = gd.src[0].fx = 1;
  gd.B = 1; gd.ID = 1; gd.IH = 1; gd.IW = (int)g.T;
  gd.KD = gd.KH = gd.KW = 1;
  gd.N = d->N; gd.wpack = d->wpack; gd.out = M; gd.ld_out = Np;
  const int rc = clx_igemm_launch(&gd, 16, g.T * C, (long long)Np * C, g.T * Np, st);
  if (rc) return rc;
  const long long tot_out = g.T * (Np / 4);
  wino_output_kernel<<<grid_for(tot_out, 256), 256, 0, st>>>(M, Np / 4, g, d->bias, d->relu, d->mask,
                                                               d->ld_mask, d->out, d->ld_out, d->N, tot_out);
  CLX_CHECK_LAUNCH("clx_conv_fwd(winograd)");
  return CLX_OK;
}

int clx_wino_wgrad(const clx_conv_desc* d, const float* dy, int ld_dy, float* dwpack, float* dbias,
                   hipStream_t st) {
  CLX_REQUIRE(applicable(d) && d->PH == 0, "clx_conv_wgrad: CLX_ALGO_WINOGRAD does not apply to this geometry");
  const size_t need = clx_conv_workspace_bytes(d, CLX_PASS_WGRAD);
  CLX_REQUIRE(d->workspace != nullptr && d->workspace_bytes >= need && need > 0,
              "clx_conv_wgrad: Winograd needs %zu workspace bytes (%zu given)", need, d->workspace_bytes);
  CLX_REQUIRE(((uintptr_t)d->workspace & 15) == 0, "clx_conv_wgrad: workspace must be 16-byte aligned");
  const Geom g = geom(d);
  const clx_src& S = d->src[0];
  const int C = S.C, N = d->N;     // N is a multiple of 4 (validated by clx_conv_wgrad)
  float* V = (float*)d->workspace;
  float* Md = V + 16 * g.T * C;
  const long long tot_in = g.T * (C / 4);
  wino_input_kernel<<<grid_for(tot_in, 256), 256, 0, st>>>(S.ptr, S.ld, C / 4, g, V, tot_in);
  const long long tot_dy = g.T * (N / 4);
  CLX_REQUIRE(N <= 8192, "clx_conv_wgrad: too many output channels for the Winograd bias accumulator");
  int blocks = grid_for(tot_dy, 256);
  if (blocks > 2048) blocks = 2048;
  wino_dy_kernel<<<blocks, 256, (size_t)N * sizeof(float), st>>>(dy, ld_dy, N / 4, g, Md, dbias, N, tot_dy);
  clx_conv_desc gd = {};
  gd.nsrc = 1;
  gd.src[0].ptr = V; gd.src[0].C = C; gd.src[0].ld = C;
  gd.src[0].D = 1; gd.src[0].H = 1; gd.src[0].W = (int)g.T;
  gd.src[0].fz = gd.src[0].fy = gd.src[0].fx = 1;
  gd.B = 1; gd.ID = 1; gd.IH = 1; gd.IW = (int)g.T;
  gd.KD = gd.KH = gd.KW = 1;
  gd.N = N;
  const int rc = clx_wgrad_launch(&gd, Md, N, dwpack, nullptr, 16, g.T * C, g.T * N, (long long)N * C, st);
  if (rc) return rc;
  CLX_CHECK_LAUNCH("clx_conv_wgrad(winograd)");
  return CLX_OK;
}

int clx_wino_pack(const float* w, float* wp, int cout, int cin, int cin_pad, int cout_pad, int dgrad,
                  hipStream_t st) {
  const int rows = dgrad ? cin_pad : cout_pad, cols = dgrad ? cout_pad : cin_pad;
  const long long total = (long long)rows * cols;
  wino_filter_kernel<<<grid_for(total, 256), 256, 0, st>>>(w, wp, cout, cin, rows, cols, dgrad, total);
  return CLX_OK;
}

extern "C" int clx_unpack_wgrad_wino(const float* du, float* dw, int cout, int cin, int rows,
                                     int cin_pad, clx_stream stream) {
  CLX_REQUIRE(du && dw, "clx_unpack_wgrad_wino: null pointer");
  CLX_REQUIRE(cout > 0 && cin > 0 && rows >= cout && cin_pad >= cin, "clx_unpack_wgrad_wino: bad extents");
  const long long total = (long long)cout * cin;
  wino_unpack_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(du, dw, cout, cin, rows,
                                                                           cin_pad, total);
  CLX_CHECK_LAUNCH("clx_unpack_wgrad_wino");
  return CLX_OK;
}
