// Bare v_mfma_f32_32x32x2_f32 loop, operands in registers: what the matrix pipes deliver on this
// part with nothing else in the way (occupancy: `waves` per SIMD).  Build: hipcc --offload-arch=gfx950
// -O3 -o tools/exp/mfma_peak tools/exp/mfma_peak.hip ; run: tools/exp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(int blocks_per_cu, int iters) {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = cus * blocks_per_cu;
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.0f, 1e-3f);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.0f, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 /*waves*/ * iters * 4.0 * NACC * (2.0 * 32 * 32 * 2);
    printf("accumulators %d  blocks/CU %d  iters %d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, iters, ms,
           flops / ms / 1e9);
  }
  hipFree(out);
}

int main() {
  run<4>(1, 20000);      // one wave per SIMD, ~0.9 s of MFMAs at nominal clock... keep it short
  run<4>(2, 10000);
  run<4>(3, 8000);
  run<1>(2, 40000);
  return 0;
}
