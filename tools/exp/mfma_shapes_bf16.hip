// Bare bf16 MFMA loops on RANDOM operands held in registers, 64 x 64 outputs per wave (the split-precision product's wave
// tile: six products per fragment pair), two waves per SIMD: v_mfma_f32_32x32x16_bf16 (4 tiles x 16 accumulators) against
// v_mfma_f32_16x16x32_bf16 (16 tiles x 4) — MI355X_MICROARCH.md says the chip holds a higher clock on the second shape.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/mfma_shapes_bf16 tools/exp/mfma_shapes_bf16.hip ; run it on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 6 fragments of A and 6 of B per lane (2 tiles x 3 pieces), random bf16 bits with sane exponents
__device__ u32x4 frag(const unsigned int* rnd, int k) {
  u32x4 v;
  for (int e = 0; e < 4; ++e) {
    unsigned int w = rnd[(threadIdx.x * 12 + k) * 4 + e];
    // two bf16: sign random, exponent 120..127, mantissa random
    unsigned int lo = (w & 0x807fu) | ((120u + ((w >> 8) & 7u)) << 7), hi = ((w >> 16) & 0x807fu) | ((120u + ((w >> 24) & 7u)) << 7);
    v[e] = lo | (hi << 16);
  }
  return v;
}

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void loop(const unsigned int* rnd, float* out, int iters) {
  u32x4 a[2][3], b[2][3];
  for (int i = 0; i < 2; ++i)
    for (int q = 0; q < 3; ++q) { a[i][q] = frag(rnd, i * 3 + q); b[i][q] = frag(rnd, 6 + i * 3 + q); }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int t = 0; t < 6; ++t) {
            const int pa = t == 0 ? 2 : t == 1 ? 0 : t == 2 ? 1 : t == 3 ? 1 : 0, pb = t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 1 : t == 3 ? 0 : t == 4 ? 1 : 0;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][pa]), __builtin_bit_cast(bf16x8, b[j][pb]), acc[i][j], 0, 0, 0);
          }
      // keep the operands "changing" a little so that nothing is hoisted (one VALU per iteration)
      a[0][0][0] ^= (unsigned)it & 1u;
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  } else {
    // the same 64 x 64 x (16 k) x 6 products as 16 tiles of 16 x 16 with K = 32: two k16 steps per instruction, so an
    // iteration here covers TWO iterations of the other loop: 4 x 4 tiles x 6 products
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < 6; ++t) {
            const int pa = t == 0 ? 2 : t == 1 ? 0 : t == 2 ? 1 : t == 3 ? 1 : 0, pb = t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 1 : t == 3 ? 0 : t == 4 ? 1 : 0;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 1][pa] ^ (unsigned)(i >> 1)), __builtin_bit_cast(bf16x8, b[j & 1][pb] ^ (unsigned)(j >> 1)), acc[i][j], 0, 0, 0);
          }
      a[0][0][0] ^= (unsigned)it & 2u;
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int SHAPE>
static void run(const unsigned int* rnd, int iters) {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out;
  hipMalloc(&out, (size_t)cus * 512 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  loop<SHAPE><<<cus, 512>>>(rnd, out, iters);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    loop<SHAPE><<<cus, 512>>>(rnd, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)cus * 8 * iters * 24.0 * (2.0 * 32 * 32 * 16);      // bf16 FLOPs; f32-equivalent = / 6
    printf("shape %dx%d: %d iterations %.3f ms  %.0f TFLOP/s bf16 = %.1f f32-equivalent\n", SHAPE, SHAPE, iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 6);
  }
  hipFree(out);
}

// argument "const": every operand the same value (what a timing-only ablation of a product kernel multiplies) instead of
// random bits — the same instruction stream at a lower switching power
int main(int argc, char** argv) {
  const bool constant = argc > 1 && argv[1][0] == 'c';
  unsigned int* h = (unsigned int*)malloc(512 * 12 * 4 * 4);
  srand(1);
  for (int i = 0; i < 512 * 12 * 4; ++i) h[i] = constant ? 0u : (unsigned)rand() ^ ((unsigned)rand() << 16);
  printf("operands: %s\n", constant ? "constant" : "random");
  unsigned int* rnd;
  hipMalloc(&rnd, 512 * 12 * 4 * 4);
  hipMemcpy(rnd, h, 512 * 12 * 4 * 4, hipMemcpyHostToDevice);
  run<32>(rnd, 40000);
  run<16>(rnd, 40000);
  run<32>(rnd, 40000);
  run<16>(rnd, 40000);
  return 0;
}
