// Train-step tail kernels: embedding gather (+ coordinates), OCE pair loss with
// its gradient, the fused gather->loss->scatter kernel and the Adam step.
// All are HBM/L2-bound; loss sums are reduced per wavefront with DPP shuffles,
// per block through LDS, then one f64 atomic per block.
#include "clx_common.h"

namespace {

// (every block ends with three float64 atomics on the SAME three addresses: few, fat blocks)
inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return (int)g;
}

// linear index of coordinate row `co` (x = last axis first) in a (Z, Y, X) grid, with the
// index rules of the advanced indexing the reference uses (unet.py:113-118): -n..-1 wrap
// around once, anything else outside [0, n) is an IndexError -> -1 here.
__device__ __forceinline__ long long coord_index(const long long* co, int ND, int Z, int Y, int X) {
  long long x = co[0], y = co[1];
  long long z = (ND == 3) ? co[2] : 0;
  x += (x < 0) ? X : 0;
  y += (y < 0) ? Y : 0;
  z += (z < 0) ? Z : 0;
  const bool ok = (unsigned long long)x < (unsigned long long)X && (unsigned long long)y < (unsigned long long)Y &&
                  (unsigned long long)z < (unsigned long long)Z;
  return ok ? (z * Y + y) * X + x : -1;
}

// UNetModel.select_and_add_coordinates [cellulus/models/unet.py:108-124]
__global__ void gather_add_fwd_kernel(const float* __restrict__ offsets,
                                      const long long* __restrict__ coords,
                                      float* __restrict__ sel, int P, int ND, int Z, int Y, int X,
                                      long long npix, long long total, int* __restrict__ oob) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / P;
    const long long* co = coords + i * ND;
    const long long idx = coord_index(co, ND, Z, Y, X);
    if (idx < 0) {                      // never dereferenced; the host raises IndexError
      if (oob) atomicAdd(oob, 1);
      for (int c = 0; c < ND; ++c) sel[i * ND + c] = __builtin_nanf("");
      continue;
    }
    const float* ob = offsets + b * ND * npix + idx;
    for (int c = 0; c < ND; ++c) sel[i * ND + c] = ob[(long long)c * npix] + (float)co[c];
  }
}

__global__ void gather_add_bwd_kernel(const float* __restrict__ dsel,
                                      const long long* __restrict__ coords,
                                      float* __restrict__ doffsets, int P, int ND, int Z, int Y, int X,
                                      long long npix, long long total, int* __restrict__ oob) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / P;
    const long long* co = coords + i * ND;
    const long long idx = coord_index(co, ND, Z, Y, X);
    if (idx < 0) {
      if (oob) atomicAdd(oob, 1);
      continue;
    }
    float* ob = doffsets + b * ND * npix + idx;
    for (int c = 0; c < ND; ++c) atomicAdd(ob + (long long)c * npix, dsel[i * ND + c]);
  }
}

// OCELoss.forward [cellulus/criterions/oce_loss.py:45-63], per pair:
//   d = |a - r|_2, oce = 1 - exp(-d^2 / T), reg = w |a|_2
//   d(oce + reg)/da = (2/T) exp(-d^2/T) (a - r) + w a / |a|   (0 at zero norm)
template <int ND>
__device__ __forceinline__ void oce_pair(const float* a, const float* r, float T, float w,
                                         float& oce, float& reg, float* da) {
  float s = 0.f, na = 0.f, diff[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) {
    diff[c] = a[c] - r[c];
    s += diff[c] * diff[c];
    na += a[c] * a[c];
  }
  const float d = sqrtf(s);
  const float e = expf(-(d * d) / T);
  oce = 1.f - e;
  const float nrm = sqrtf(na);
  reg = w * nrm;
  const float k = 2.f * e / T;
  const float inv = (nrm > 0.f) ? w / nrm : 0.f;
#pragma unroll
  for (int c = 0; c < ND; ++c) da[c] = k * diff[c] + inv * a[c];
}

__device__ __forceinline__ void block_accumulate(double oce, double reg, double* sums) {
  __shared__ double red[2][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    oce += __shfl_down(oce, o, 64);
    reg += __shfl_down(reg, o, 64);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[0][wid] = oce; red[1][wid] = reg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double o = 0., g = 0.;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { o += red[0][k]; g += red[1][k]; }
    atomicAdd(sums + 0, o + g);
    atomicAdd(sums + 1, o);
    atomicAdd(sums + 2, g);
  }
}

template <int ND>
__global__ __launch_bounds__(256) void oce_loss_kernel(const float* __restrict__ a,
                                                       const float* __restrict__ r,
                                                       float* __restrict__ da, double* sums,
                                                       long long npairs, float T, float w,
                                                       float gscale) {
  double oce_acc = 0., reg_acc = 0.;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npairs;
       i += (long long)gridDim.x * blockDim.x) {
    float av[ND], rv[ND], g[ND], oce, reg;
#pragma unroll
    for (int c = 0; c < ND; ++c) { av[c] = a[i * ND + c]; rv[c] = r[i * ND + c]; }
    oce_pair<ND>(av, rv, T, w, oce, reg, g);
    oce_acc += oce; reg_acc += reg;
    if (da) {
#pragma unroll
      for (int c = 0; c < ND; ++c) da[i * ND + c] = gscale * g[c];
    }
  }
  block_accumulate(oce_acc, reg_acc, sums);
}

// train.py:170-178 in one pass: gather anchor/reference embeddings, loss,
// scatter-add the anchor gradient into doffsets (planar).
template <int ND>
__global__ __launch_bounds__(256) void oce_pairs_fused_kernel(
    const float* __restrict__ offsets, const long long* __restrict__ anchor,
    const long long* __restrict__ reference, float* __restrict__ doffsets, double* sums,
    int P, int Z, int Y, int X, long long npix, long long total, float T, float w) {
  double oce_acc = 0., reg_acc = 0.;
  int bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / P;
    const long long* ca = anchor + i * ND;
    const long long* cr = reference + i * ND;
    const long long ia = coord_index(ca, ND, Z, Y, X), ir = coord_index(cr, ND, Z, Y, X);
    if (ia < 0 || ir < 0) {             // skipped and counted: sums[3], the host raises IndexError
      ++bad;
      continue;
    }
    const float* ob = offsets + b * ND * npix;
    float av[ND], rv[ND], g[ND], oce, reg;
#pragma unroll
    for (int c = 0; c < ND; ++c) {
      av[c] = ob[(long long)c * npix + ia] + (float)ca[c];
      rv[c] = ob[(long long)c * npix + ir] + (float)cr[c];
    }
    oce_pair<ND>(av, rv, T, w, oce, reg, g);
    oce_acc += oce; reg_acc += reg;
    float* gb = doffsets + b * ND * npix + ia;
#pragma unroll
    for (int c = 0; c < ND; ++c) atomicAdd(gb + (long long)c * npix, g[c]);
  }
  if (bad) atomicAdd(sums + 3, (double)bad);
  block_accumulate(oce_acc, reg_acc, sums);
}

// Reproducible form of oce_pairs_fused_kernel: the anchor gradients are scattered as 2^-40 fixed-point
// integers (64-bit integer atomics: associative, so the order of arrival does not matter), the loss sums
// leave each block as one partial that oce_pairs_det_finish adds in block order.
constexpr double DET_SCALE = 1099511627776.0;        // 2^40: |g| < 2^22 per pair, millions of pairs per pixel fit
constexpr int DET_BLOCKS = 1024;

template <int ND>
__global__ __launch_bounds__(256) void oce_pairs_det_kernel(
    const float* __restrict__ offsets, const long long* __restrict__ anchor,
    const long long* __restrict__ reference, unsigned long long* __restrict__ acc, double* __restrict__ partial,
    int P, int Z, int Y, int X, long long npix, long long total, float T, float w) {
  __shared__ double red[3][4];
  double oce_acc = 0., reg_acc = 0., bad = 0.;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / P;
    const long long* ca = anchor + i * ND;
    const long long* cr = reference + i * ND;
    const long long ia = coord_index(ca, ND, Z, Y, X), ir = coord_index(cr, ND, Z, Y, X);
    if (ia < 0 || ir < 0) { bad += 1.; continue; }
    const float* ob = offsets + b * ND * npix;
    float av[ND], rv[ND], g[ND], oce, reg;
#pragma unroll
    for (int c = 0; c < ND; ++c) {
      av[c] = ob[(long long)c * npix + ia] + (float)ca[c];
      rv[c] = ob[(long long)c * npix + ir] + (float)cr[c];
    }
    oce_pair<ND>(av, rv, T, w, oce, reg, g);
    oce_acc += oce; reg_acc += reg;
    unsigned long long* gb = acc + b * ND * npix + ia;
#pragma unroll
    for (int c = 0; c < ND; ++c)
      atomicAdd(gb + (long long)c * npix, (unsigned long long)__double2ll_rn((double)g[c] * DET_SCALE));
  }
  // fixed-shape reduction: wave shuffles, then the four waves in order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    oce_acc += __shfl_down(oce_acc, o, 64);
    reg_acc += __shfl_down(reg_acc, o, 64);
    bad += __shfl_down(bad, o, 64);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[0][wid] = oce_acc; red[1][wid] = reg_acc; red[2][wid] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double o = 0., r = 0., bd = 0.;
    for (int k = 0; k < 4; ++k) { o += red[0][k]; r += red[1][k]; bd += red[2][k]; }
    partial[4 * blockIdx.x + 0] = o; partial[4 * blockIdx.x + 1] = r; partial[4 * blockIdx.x + 2] = bd;
  }
}

__global__ void oce_pairs_det_finish(const unsigned long long* __restrict__ acc, float* __restrict__ doffsets,
                                     long long n, const double* __restrict__ partial, int nblocks, double* sums) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    doffsets[i] = (float)((double)(long long)acc[i] * (1.0 / DET_SCALE));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double o = 0., r = 0., bd = 0.;
    for (int k = 0; k < nblocks; ++k) { o += partial[4 * k]; r += partial[4 * k + 1]; bd += partial[4 * k + 2]; }
    // ADDS, as clx_oce_pairs_fused does (the caller zeroes `sums` once per step: half batches add up)
    sums[0] += o + r; sums[1] += o; sums[2] += r; sums[3] += bd;
  }
}

// torch.optim.Adam single-tensor step order (weight_decay coupled into grad):
//   g += wd*p; m = lerp(m, g, 1-b1); v = b2*v + (1-b2) g*g;
//   denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m / denom
// `skip_if_positive` (optional): a device double — the bad-coordinate count of the loss kernel — that turns the
// whole step into a no-op when it is > 0, so that the step can be enqueued before the host has looked at it.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, long long n,
                            float one_minus_b1, float b2, float one_minus_b2, float eps,
                            float wd, float step_size, float bc2_sqrt, const double* __restrict__ skip_if_positive) {
  if (skip_if_positive != nullptr && *skip_if_positive > 0.0) return;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float pv = p[i];
    const float gv = g[i] + wd * pv;
    float mv = m[i];
    mv = mv + one_minus_b1 * (gv - mv);
    const float vv = b2 * v[i] + one_minus_b2 * gv * gv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    m[i] = mv;
    v[i] = vv;
    p[i] = pv - step_size * (mv / denom);
  }
}

// ---------------------------------------------------------------------------------------------
// Optional on-device pair sampler (the distribution of ZarrDataset.sample_coordinates /
// sample_offsets_within_radius, cellulus/datasets/zarr_dataset.py:177-242): anchor column d uniform
// on the integers [lo, hi[d]], every anchor repeated `num_refs` times, reference = anchor + an
// offset uniform over the table of admissible offsets (|o|^2 < kappa^2, o != 0).  Counter-based
// generator (splitmix64 of seed, stream, batch row, pair, column): any (seed, stream) reproduces
// its pairs, no state.  NOT the reference's random stream — an opt-in that removes the 38 MB of
// int64 coordinates per step from the host pipeline.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
// uniform integer in [0, n) from 32 random bits (multiply-shift: bias < n / 2^32)
__device__ __forceinline__ unsigned int below(unsigned long long r, unsigned int n) {
  return (unsigned int)(((r >> 32) * (unsigned long long)n) >> 32);
}

__global__ void sample_pairs_kernel(long long* __restrict__ anchor, long long* __restrict__ reference,
                                    const int* __restrict__ offsets, int noffsets, int num_anchors,
                                    int num_refs, int ND, int lo, int hi0, int hi1, int hi2,
                                    unsigned long long seed, unsigned long long stream, long long total) {
  const int hi[3] = {hi0, hi1, hi2};
  const long long P = (long long)num_anchors * num_refs;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / P, p = i - b * P;
    const long long a = p / num_refs;
    const unsigned long long key = splitmix64(seed ^ splitmix64(stream * 0x100000001b3ull + (unsigned long long)b));
    const unsigned long long ra = splitmix64(key ^ (0xa5a5a5a5ull + 2ull * (unsigned long long)a));
    const unsigned long long rb = splitmix64(ra);
    const unsigned long long ro = splitmix64(key ^ (0x5a5a5a5a00000000ull + (unsigned long long)p));
    const int o = (int)below(ro, (unsigned int)noffsets);
    const unsigned long long bits[3] = {ra, ra << 32, rb};       // three independent 32-bit fields
    for (int d = 0; d < ND; ++d) {
      const long long c = lo + (long long)below(bits[d], (unsigned int)(hi[d] - lo + 1));
      anchor[i * ND + d] = c;
      reference[i * ND + d] = c + offsets[o * ND + d];
    }
  }
}

}  // namespace

extern "C" int clx_sample_pairs(long long* anchor, long long* reference, const int* offsets, int noffsets,
                                int B, int num_anchors, int num_refs, int ND, int lo, const int* hi,
                                unsigned long long seed, unsigned long long stream_id, clx_stream stream) {
  CLX_REQUIRE(anchor && reference && offsets && hi, "clx_sample_pairs: null pointer");
  CLX_REQUIRE(B > 0 && num_anchors > 0 && num_refs > 0 && noffsets > 0 && (ND == 2 || ND == 3),
              "clx_sample_pairs: bad extents");
  for (int d = 0; d < ND; ++d) CLX_REQUIRE(hi[d] >= lo, "clx_sample_pairs: empty anchor range");
  const long long total = (long long)B * num_anchors * num_refs;
  sample_pairs_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      anchor, reference, offsets, noffsets, num_anchors, num_refs, ND, lo, hi[0], hi[1], ND == 3 ? hi[2] : hi[1],
      seed, stream_id, total);
  CLX_CHECK_LAUNCH("clx_sample_pairs");
  return CLX_OK;
}

extern "C" int clx_gather_add_fwd(const float* offsets, const long long* coords, float* sel,
                                  int B, int P, int ND, int Z, int Y, int X, int* oob_count,
                                  clx_stream stream) {
  CLX_REQUIRE(offsets && coords && sel, "clx_gather_add_fwd: null pointer");
  CLX_REQUIRE(B > 0 && P >= 0 && (ND == 2 || ND == 3) && Z > 0 && Y > 0 && X > 0,
              "clx_gather_add_fwd: bad extents");
  CLX_REQUIRE(ND == 3 || Z == 1, "clx_gather_add_fwd: Z must be 1 for 2-D");
  if (P == 0) return CLX_OK;
  const long long total = (long long)B * P;
  gather_add_fwd_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      offsets, coords, sel, P, ND, Z, Y, X, (long long)Z * Y * X, total, oob_count);
  CLX_CHECK_LAUNCH("clx_gather_add_fwd");
  return CLX_OK;
}

extern "C" int clx_gather_add_bwd(const float* dsel, const long long* coords, float* doffsets,
                                  int B, int P, int ND, int Z, int Y, int X, int* oob_count,
                                  clx_stream stream) {
  CLX_REQUIRE(dsel && coords && doffsets, "clx_gather_add_bwd: null pointer");
  CLX_REQUIRE(B > 0 && P >= 0 && (ND == 2 || ND == 3) && Z > 0 && Y > 0 && X > 0,
              "clx_gather_add_bwd: bad extents");
  if (P == 0) return CLX_OK;
  const long long total = (long long)B * P;
  gather_add_bwd_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      dsel, coords, doffsets, P, ND, Z, Y, X, (long long)Z * Y * X, total, oob_count);
  CLX_CHECK_LAUNCH("clx_gather_add_bwd");
  return CLX_OK;
}

extern "C" int clx_oce_loss_fwd_bwd(const float* a, const float* r, float* da, double* sums,
                                    long long npairs, int ND, float temperature,
                                    float reg_weight, float grad_scale, clx_stream stream) {
  CLX_REQUIRE(a && r && sums, "clx_oce_loss_fwd_bwd: null pointer");
  CLX_REQUIRE(npairs >= 0 && (ND == 2 || ND == 3), "clx_oce_loss_fwd_bwd: bad extents");
  CLX_REQUIRE(temperature != 0.f, "clx_oce_loss_fwd_bwd: temperature must be non-zero");
  if (npairs == 0) return CLX_OK;
  const int grid = grid_for(npairs, 256);
  if (ND == 2)
    oce_loss_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>(a, r, da, sums, npairs, temperature, reg_weight, grad_scale);
  else
    oce_loss_kernel<3><<<grid, 256, 0, (hipStream_t)stream>>>(a, r, da, sums, npairs, temperature, reg_weight, grad_scale);
  CLX_CHECK_LAUNCH("clx_oce_loss_fwd_bwd");
  return CLX_OK;
}

extern "C" int clx_oce_pairs_fused(const float* offsets, const long long* anchor,
                                   const long long* reference, float* doffsets, double* sums,
                                   int B, int P, int ND, int Z, int Y, int X,
                                   float temperature, float reg_weight, clx_stream stream) {
  CLX_REQUIRE(offsets && anchor && reference && doffsets && sums, "clx_oce_pairs_fused: null pointer");
  CLX_REQUIRE(B > 0 && P >= 0 && (ND == 2 || ND == 3) && Z > 0 && Y > 0 && X > 0,
              "clx_oce_pairs_fused: bad extents");
  CLX_REQUIRE(temperature != 0.f, "clx_oce_pairs_fused: temperature must be non-zero");
  if (P == 0) return CLX_OK;
  const long long total = (long long)B * P, npix = (long long)Z * Y * X;
  const int grid = grid_for(total, 256);
  if (ND == 2)
    oce_pairs_fused_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>(
        offsets, anchor, reference, doffsets, sums, P, Z, Y, X, npix, total, temperature, reg_weight);
  else
    oce_pairs_fused_kernel<3><<<grid, 256, 0, (hipStream_t)stream>>>(
        offsets, anchor, reference, doffsets, sums, P, Z, Y, X, npix, total, temperature, reg_weight);
  CLX_CHECK_LAUNCH("clx_oce_pairs_fused");
  return CLX_OK;
}

extern "C" int clx_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                             long long n, double lr, double beta1, double beta2, double eps,
                             double weight_decay, int step, clx_stream stream) {
  return clx_adam_step_guarded(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step,
                               nullptr, stream);
}

extern "C" int clx_adam_step_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                     long long n, double lr, double beta1, double beta2, double eps,
                                     double weight_decay, int step, const double* skip_if_positive,
                                     clx_stream stream) {
  CLX_REQUIRE(param && grad && exp_avg && exp_avg_sq, "clx_adam_step: null pointer");
  CLX_REQUIRE(n >= 0 && step >= 1, "clx_adam_step: bad n/step");
  if (n == 0) return CLX_OK;
  // host-side scalars in double, as torch computes them in Python floats
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  adam_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(
      param, grad, exp_avg, exp_avg_sq, n, (float)(1.0 - beta1), (float)beta2,
      (float)(1.0 - beta2), (float)eps, (float)weight_decay, step_size, bc2_sqrt, skip_if_positive);
  CLX_CHECK_LAUNCH("clx_adam_step");
  return CLX_OK;
}

extern "C" size_t clx_oce_pairs_det_scratch_bytes(int B, int ND, long long npix) {
  return (size_t)B * ND * npix * sizeof(unsigned long long) + (size_t)DET_BLOCKS * 4 * sizeof(double);
}

extern "C" int clx_oce_pairs_fused_det(const float* offsets, const long long* anchor, const long long* reference,
                                       float* doffsets, double* sums, int B, int P, int ND, int Z, int Y, int X,
                                       float temperature, float reg_weight, void* scratch, clx_stream stream) {
  CLX_REQUIRE(offsets && anchor && reference && doffsets && sums && scratch, "clx_oce_pairs_fused_det: null pointer");
  CLX_REQUIRE(B > 0 && P >= 0 && (ND == 2 || ND == 3) && Z > 0 && Y > 0 && X > 0 && (ND == 3 || Z == 1),
              "clx_oce_pairs_fused_det: bad extents");
  CLX_REQUIRE(temperature > 0.f, "clx_oce_pairs_fused_det: temperature must be positive");
  CLX_REQUIRE(((uintptr_t)scratch & 7) == 0, "clx_oce_pairs_fused_det: scratch must be 8-byte aligned");
  const long long npix = (long long)Z * Y * X, total = (long long)B * P, n = (long long)B * ND * npix;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* acc = (unsigned long long*)scratch;
  double* partial = (double*)(acc + n);
  if (hipMemsetAsync(acc, 0, (size_t)n * sizeof(unsigned long long), st) != hipSuccess) {
    clx_set_error("clx_oce_pairs_fused_det: memset failed");
    return CLX_ERR_LAUNCH;
  }
  long long g = (total + 255) / 256;
  const int grid = (int)(g < 1 ? 1 : g > DET_BLOCKS ? DET_BLOCKS : g);
  if (ND == 2)
    oce_pairs_det_kernel<2><<<grid, 256, 0, st>>>(offsets, anchor, reference, acc, partial, P, Z, Y, X, npix, total,
                                                  temperature, reg_weight);
  else
    oce_pairs_det_kernel<3><<<grid, 256, 0, st>>>(offsets, anchor, reference, acc, partial, P, Z, Y, X, npix, total,
                                                  temperature, reg_weight);
  long long gf = (n + 255) / 256;
  oce_pairs_det_finish<<<(int)(gf > 2048 ? 2048 : gf), 256, 0, st>>>(acc, doffsets, n, partial, grid, sums);
  CLX_CHECK_LAUNCH("clx_oce_pairs_fused_det");
  return CLX_OK;
}
