// The K-loop core of conv_igemm_kernel<128,128,2,2> in isolation: 4 waves, each a 2x2 grid of 32x32
// accumulators, per chunk 4 groups of [4 ds_read_b128 -> 16 v_mfma_f32_32x32x2_f32], operand
// fragments double-buffered one group ahead, one barrier per chunk; LDS content static (no global
// loads, no LDS stores).  Variants switch the barrier / the fragment reads off to price them.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/core_loop tools/exp/core_loop.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDS_LD = 36, BM = 128, BN = 128;

template <bool BARRIER, bool READS, int LDS_FLOATS>
__global__ __launch_bounds__(256, 2) void core(float* out, int chunks) {
  __shared__ float smem[LDS_FLOATS];
  for (int i = threadIdx.x; i < 2 * (BM + BN) * LDS_LD; i += 256) smem[i] = 1e-3f * (i & 7);
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int a_base = (wm * 64 + li) * LDS_LD + 4 * lh;
  const int b_base = 2 * BM * LDS_LD + (wn * 64 + li) * LDS_LD + 4 * lh;
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f32x4 af[2][2], bf[2][2];
  for (int s = 0; s < 2; ++s)
    for (int a = 0; a < 2; ++a) { af[s][a] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + 1); bf[s][a] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f}; }
  auto load_frags = [&](int buf, int q, int slot) {
    if (!READS) return;
#pragma unroll
    for (int a = 0; a < 2; ++a)
      af[slot][a] = *reinterpret_cast<const f32x4*>(&smem[buf * BM * LDS_LD + a_base + a * 32 * LDS_LD + 8 * q]);
#pragma unroll
    for (int c = 0; c < 2; ++c)
      bf[slot][c] = *reinterpret_cast<const f32x4*>(&smem[buf * BN * LDS_LD + b_base + c * 32 * LDS_LD + 8 * q]);
  };
  auto mfma_group = [&](int slot) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][a][e], bf[slot][c][e], acc[a][c], 0, 0, 0);
  };
  int buf = 0;
  load_frags(buf, 0, 0);
  for (int ch = 0; ch < chunks; ++ch) {
    load_frags(buf, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(0);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(buf, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(1);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(buf, 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(1);
    __builtin_amdgcn_sched_barrier(0);
    if (BARRIER) __syncthreads();
    buf ^= 1;
    load_frags(buf, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool BARRIER, bool READS, int LDS_FLOATS>
static void run(const char* name, int blocks_per_cu, int chunks) {
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = cus * blocks_per_cu;
  float* out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  core<BARRIER, READS, LDS_FLOATS><<<blocks, 256>>>(out, chunks);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    core<BARRIER, READS, LDS_FLOATS><<<blocks, 256>>>(out, chunks);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)blocks * 4 * chunks * 64.0 * (2.0 * 32 * 32 * 2);
  printf("%-44s blocks/CU %d: %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, best, flops / best / 1e9);
  (void)hipFree(out);
}

static int main_core() {
  constexpr int TWO = 2 * (BM + BN) * LDS_LD;          // 73.7 KB: two blocks per CU
  constexpr int ONE = TWO + 4096;                      // > 80 KB: one block per CU
  run<true, true, TWO>("barrier + fragment reads (the kernel's core)", 2, 2000);
  run<false, true, TWO>("no barrier", 2, 2000);
  run<true, false, TWO>("no fragment reads", 2, 2000);
  run<false, false, TWO>("neither (MFMAs only)", 2, 2000);
  run<true, true, ONE>("barrier + reads, ONE block per CU", 1, 4000);
  run<false, true, ONE>("no barrier, ONE block per CU", 1, 4000);
  return 0;
}

// ---- the same core plus the operand traffic of a 1x1 convolution: per chunk every thread issues
// 4 + 4 global_load_dwordx4 (A rows from a large activation, B rows from a small weight panel) and
// 8 ds_write_b128 into the other LDS buffer.  MODE 0: placed as in the kernel (A loads before group 0,
// B loads before group 1, LDS stores before group 3, sched_barrier(0) fences); MODE 1: no fences at
// all (compiler's order); MODE 2: sched_group_barrier pipelines — one VMEM / DS-write per few MFMAs.
template <int MODE>
__global__ __launch_bounds__(256, 2) void full(const float* __restrict__ A, const float* __restrict__ B, float* out,
                                               int chunks, int ldk, long long rows_total) {
  __shared__ float smem[2 * (BM + BN) * LDS_LD];
  float (*As)[BM * LDS_LD] = reinterpret_cast<float (*)[BM * LDS_LD]>(smem);
  float (*Bs)[BN * LDS_LD] = reinterpret_cast<float (*)[BN * LDS_LD]>(smem + 2 * BM * LDS_LD);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  const int a_base = (wm * 64 + li) * LDS_LD + 4 * lh;
  const int b_base = (wn * 64 + li) * LDS_LD + 4 * lh;
  const float* arow[4];
  const float* brow[4];
  for (int j = 0; j < 4; ++j) {
    arow[j] = A + ((long long)blockIdx.x * BM % rows_total + lrow + 32 * j) * ldk + lcol;
    brow[j] = B + (long long)(lrow + 32 * j) * ldk + lcol;
  }
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f32x4 ra[4], rw[4], af[2][2], bf[2][2];
  int c0 = 0;
  auto load_a = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const f32x4*>(arow[j] + c0);
  };
  auto load_b = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) rw[j] = *reinterpret_cast<const f32x4*>(brow[j] + c0);
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&As[buf][(lrow + 32 * j) * LDS_LD + lcol]) = ra[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + 32 * j) * LDS_LD + lcol]) = rw[j];
  };
  auto load_frags = [&](int buf, int q, int slot) {
#pragma unroll
    for (int a = 0; a < 2; ++a) af[slot][a] = *reinterpret_cast<const f32x4*>(&As[buf][a_base + a * 32 * LDS_LD + 8 * q]);
#pragma unroll
    for (int c = 0; c < 2; ++c) bf[slot][c] = *reinterpret_cast<const f32x4*>(&Bs[buf][b_base + c * 32 * LDS_LD + 8 * q]);
  };
  auto mfma_group = [&](int slot) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][a][e], bf[slot][c][e], acc[a][c], 0, 0, 0);
  };
  load_a(); load_b(); store_chunk(0);
  __syncthreads();
  int buf = 0;
  load_frags(buf, 0, 0);
  for (int ch = 1; ch < chunks; ++ch) {
    c0 = (c0 + 32) % ldk;
    if (MODE == 0) {
      load_a(); load_frags(buf, 1, 1);
      __builtin_amdgcn_sched_barrier(0); mfma_group(0); __builtin_amdgcn_sched_barrier(0);
      load_b(); load_frags(buf, 2, 0);
      __builtin_amdgcn_sched_barrier(0); mfma_group(1); __builtin_amdgcn_sched_barrier(0);
      load_frags(buf, 3, 1);
      __builtin_amdgcn_sched_barrier(0); mfma_group(0); __builtin_amdgcn_sched_barrier(0);
      store_chunk(buf ^ 1);
      __builtin_amdgcn_sched_barrier(0); mfma_group(1); __builtin_amdgcn_sched_barrier(0);
    } else {
      load_a(); load_b();
      load_frags(buf, 1, 1); mfma_group(0);
      load_frags(buf, 2, 0); mfma_group(1);
      load_frags(buf, 3, 1); mfma_group(0);
      store_chunk(buf ^ 1); mfma_group(1);
      if (MODE == 2) {
        // 64 MFMAs, 8 VMEM reads, 12 DS reads, 8 DS writes per chunk: spread them out
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS read (12 in total; the rest fall where they may)
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // 3 MFMA
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 DS write
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        }
      }
    }
    __syncthreads();
    buf ^= 1;
    load_frags(buf, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void fill_random(float* x, long long n, unsigned seed) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    x[i] = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f);
  }
}

template <int MODE>
static void run_full(const char* name, int chunks, bool random_data = false) {
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = cus * 2 * 8;                     // 8 rounds of tiles
  const int ldk = 768;
  const long long rows = 1 << 19;                     // 1.6 GB activation
  float *A, *B, *out;
  (void)hipMalloc(&A, (size_t)rows * ldk * 4);
  (void)hipMalloc(&B, (size_t)128 * ldk * 4);
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  (void)hipMemset(A, 0, (size_t)rows * ldk * 4);
  (void)hipMemset(B, 0, (size_t)128 * ldk * 4);
  if (random_data) {
    fill_random<<<4096, 256>>>(A, rows * ldk, 1u);
    fill_random<<<64, 256>>>(B, 128ll * ldk, 2u);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  full<MODE><<<blocks, 256>>>(A, B, out, chunks, ldk, rows);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    full<MODE><<<blocks, 256>>>(A, B, out, chunks, ldk, rows);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)blocks * 4 * (chunks - 1) * 64.0 * (2.0 * 32 * 32 * 2);
  printf("%-60s chunks %3d: %.3f ms  %.1f TFLOP/s\n", name, chunks, best, flops / best / 1e9);
  (void)hipFree(A); (void)hipFree(B); (void)hipFree(out);
}

int main() {
  main_core();
  for (int chunks : {24, 216}) {
    run_full<0>("with operand traffic, kernel placement (fenced clumps)", chunks);
    run_full<1>("with operand traffic, compiler's order", chunks);
    run_full<2>("with operand traffic, sched_group_barrier pipeline", chunks);
    run_full<0>("RANDOM operands, kernel placement (fenced clumps)", chunks, true);
    run_full<2>("RANDOM operands, sched_group_barrier pipeline", chunks, true);
  }
  return 0;
}
