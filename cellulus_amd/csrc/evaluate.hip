// Joint histogram of (prediction id, ground-truth id) pairs: the integer core of
// cellulus/evaluate.py:72-100, which builds the IoU / IoG tables with one pair of
// full-image mask passes per (prediction, ground truth) combination.  Every entry of
// those tables is |P_j ∩ G_k| and the row / column sums of the same table, so one
// pass over the two label maps replaces #pred x #gt passes.  HBM-bound: 8 B per pixel.
#include "clx_common.h"

namespace {

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// present[id] = 1 for every id that occurs (np.unique without the sort: the ids come out
// ordered because the table is indexed by id)
__global__ __launch_bounds__(256) void presence_kernel(const int32_t* __restrict__ lab, long long n, int nid,
                                                       int32_t* __restrict__ present, int32_t* __restrict__ bad) {
  const long long n4 = n >> 2;
  const i32x4* l4 = reinterpret_cast<const i32x4*>(lab);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const i32x4 v = l4[i];
    int prev = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int id = v[k];
      if (id == prev) continue;                       // label maps are piecewise constant
      prev = id;
      if ((unsigned)id < (unsigned)nid) present[id] = 1; else *bad = 1;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int id = lab[(n4 << 2) + threadIdx.x];
    if ((unsigned)id < (unsigned)nid) present[id] = 1; else *bad = 1;
  }
}

// joint[prow[p]][gcol[g]] += 1.  A thread walks 8 consecutive pixels and flushes one atomic per
// run of equal (row, column): inside an object that is one atomic per 8 pixels.
__global__ __launch_bounds__(256) void joint_kernel(const int32_t* __restrict__ pred, const int32_t* __restrict__ gt,
                                                    long long n, const int32_t* __restrict__ prow,
                                                    const int32_t* __restrict__ gcol, int ncol,
                                                    unsigned long long* __restrict__ joint) {
  const long long n8 = n >> 3;
  const i32x4* p4 = reinterpret_cast<const i32x4*>(pred);
  const i32x4* g4 = reinterpret_cast<const i32x4*>(gt);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const i32x4 pa = p4[2 * i], pb = p4[2 * i + 1], ga = g4[2 * i], gb = g4[2 * i + 1];
    long long cell = -1;
    unsigned int run = 0;
    int lp = -1, lg = -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int p = k < 4 ? pa[k] : pb[k - 4], g = k < 4 ? ga[k] : gb[k - 4];
      if (p == lp && g == lg) { ++run; continue; }
      if (run) atomicAdd(&joint[cell], (unsigned long long)run);
      lp = p; lg = g;
      cell = (long long)prow[p] * ncol + gcol[g];
      run = 1;
    }
    atomicAdd(&joint[cell], (unsigned long long)run);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long long i = (n8 << 3) + threadIdx.x;
    atomicAdd(&joint[(long long)prow[pred[i]] * ncol + gcol[gt[i]]], 1ull);
  }
}

}  // namespace

extern "C" int clx_label_presence(const int32_t* labels, long long n, int nid, int32_t* present, int32_t* bad,
                                  clx_stream stream) {
  CLX_REQUIRE(labels && present && bad && n > 0 && nid > 0, "clx_label_presence: bad arguments");
  CLX_REQUIRE(((uintptr_t)labels & 15) == 0, "clx_label_presence: labels must be 16-byte aligned");
  presence_kernel<<<grid_for(n / 4 + 1, 256), 256, 0, (hipStream_t)stream>>>(labels, n, nid, present, bad);
  CLX_CHECK_LAUNCH("clx_label_presence");
  return CLX_OK;
}

extern "C" int clx_joint_histogram(const int32_t* pred, const int32_t* gt, long long n, const int32_t* pred_row,
                                   const int32_t* gt_col, int ncol, unsigned long long* joint, clx_stream stream) {
  CLX_REQUIRE(pred && gt && pred_row && gt_col && joint && n > 0 && ncol > 0, "clx_joint_histogram: bad arguments");
  CLX_REQUIRE((((uintptr_t)pred | (uintptr_t)gt) & 15) == 0, "clx_joint_histogram: label maps must be 16-byte aligned");
  joint_kernel<<<grid_for(n / 8 + 1, 256), 256, 0, (hipStream_t)stream>>>(pred, gt, n, pred_row, gt_col, ncol, joint);
  CLX_CHECK_LAUNCH("clx_joint_histogram");
  return CLX_OK;
}
