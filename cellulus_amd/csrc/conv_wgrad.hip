// f32 MFMA weight-gradient kernel for gfx950.
//
//   dwpack[tap][n][c] += sum_{p in slice} dy[p][n] * in[pix(p) (+) tap][c]
//   dbias[n]          += sum_p dy[p][n]
//
// GEMM view per tap: rows = output channels n (from dy), cols = input channels
// c (gathered input), contraction over the output pixels p (hundreds of
// thousands), split over `nslices` blocks that combine with float atomics
// (the [tap][n][c] layout makes every atomic wave-instruction two 128-byte
// row segments — the full-rate shape of MI355X_MICROARCH "Global float
// atomics").  Both operands are pixel-major in HBM, so the LDS tiles are
// [32 pixels][channels] and an MFMA fragment is a conflict-free ds_read_b32.
//
// Replaces the autograd weight/bias gradient of nn.Conv{2,3}d on the
// reference train step (cellulus/train.py:178).
#include "clx_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

// pixels per chunk: 32, but 24 for the 128x128 tile — 48 instead of 64 KB of LDS, i.e. three blocks per CU
template <int BMN, int BNC> constexpr int bkp() { return BMN == 128 && BNC == 128 ? 24 : 32; }

__device__ __attribute__((aligned(16))) float g_wgrad_zero16[4] = {0.f, 0.f, 0.f, 0.f};

struct SrcP {
  const float* ptr;
  int C, ld, D, H, W, oz, oy, ox, fz, fy, fx;
  FastDiv dfz, dfy, dfx;   // nearest-upsampling factors as fast divisions (factor 1 = identity)
};

struct WgradP {
  int nsrc;
  SrcP src[2];
  int B, ID, IH, IW, KD, KH, KW, PD, PH, PW, OD, OH, OW;
  int N, M, Ctot;
  FastDiv dOW, dOH, dOD;
  const float* dy;
  const float* zeros;   // 16 zero bytes in global memory (target of out-of-range loads)
  int ld_dy;
  float* dwp;
  float* dbias;
  int tiles_n, tiles_c, taps, nslices, chunks_per_slice;
  long long bs_x, bs_dy, bs_out;   // per-batch strides in floats
  // 1-D grid: blocks [0, n_main) are batches [0, batch_split) cut into `nslices` pixel slices;
  // blocks [n_main, ..) are the remaining batches — the launch's last, partial round of
  // co-resident blocks — cut into `nslices_tail` >= nslices shorter slices so that the partial
  // round fills the chip and ends early instead of holding a few CUs for a whole block time
  int n_main, batch_split, nslices_tail, chunks_per_slice_tail;
  // reproducible mode: one counter per (batch, output tile); slice s adds its partial sums when the counter
  // reads s and then advances it — the float additions of a tile happen in slice order
  int* turns;
};

// KS: the k-pairs of a chunk are dealt out to KS groups of waves (WAVES_M * WAVES_N * KS = 4), every group holding its own
// partial sums of the same output tile — they all end in the float-atomic combine anyway.  For the 64 x 64 tile
// (KS = 2: two waves of 64 x 32 instead of four of 32 x 32) that is one ds_read_b64 + one ds_read_b32 per two MFMAs
// where the 2 x 2 arrangement reads two b32 per MFMA.  MEASURED AND NOT DISPATCHED: the 3-D step's weight gradients 4.72
// against 4.60 ms (twice the atomics per tile, and the kernel is not bound by its LDS reads); KS = 1 everywhere.
// LIN: output pixel m reads input pixel m of a single, uncropped, un-upsampled source (1 x 1(x1) layers, the per-xi products
// of the 2-D Winograd layers): the row address is m * ld — no decode of the pixel index by three fast divisions, no
// bounds, no upsampling divisions: ~45 VALU operations per staged row and chunk less.
// LIN = 2: the per-xi products of the 3-D Winograd layers — a (KD, 1, 1) "convolution" over [planes][tiles]: only the plane
// index is decoded (two fast divisions instead of six).
template <int BMN, int BNC, int WAVES_M, int WAVES_N, int KS = 1, int LIN = 0>
__global__ __launch_bounds__(256, (BMN == 128 && BNC == 128) ? 3 : 2) void conv_wgrad_kernel(const WgradP p) {
  constexpr int BKP = bkp<BMN, BNC>();
  constexpr int TM = BMN / WAVES_M / 32;
  constexpr int TN = BNC / WAVES_N / 32;
  constexpr int A_F4 = BMN / 4, A_RPP = 256 / A_F4, A_PASSES = BKP / A_RPP;
  constexpr int B_F4 = BNC / 4, B_RPP = 256 / B_F4, B_PASSES = BKP / B_RPP;
  static_assert(WAVES_M * WAVES_N * KS == 4, "4 waves");
  constexpr int NIT = BKP / 2 / KS;           // k-pairs per wave and chunk
  static_assert(NIT * KS * 2 == BKP && NIT % 8 == 0 || KS == 1, "k-pairs divide");

  __shared__ float Ys[2][BKP * BMN];
  __shared__ float Xs[2][BKP * BNC];

  const int T = p.tiles_n * p.tiles_c * p.taps;
  const bool tail = (int)blockIdx.x >= p.n_main;
  const int ns = tail ? p.nslices_tail : p.nslices;
  const int cps = tail ? p.chunks_per_slice_tail : p.chunks_per_slice;
  // (reproducible mode: blocks keep their dispatch order, so that the slice a block waits for — a lower block
  //  index — is always already running; the XCD remap could put it behind the waiting block)
  const int u = p.turns != nullptr ? (tail ? (int)blockIdx.x - p.n_main : (int)blockIdx.x)
                : tail             ? xcd_remap((int)blockIdx.x - p.n_main, (int)gridDim.x - p.n_main)
                                   : xcd_remap((int)blockIdx.x, p.n_main);
  const int per_batch = T * ns;
  const int batch = u / per_batch + (tail ? p.batch_split : 0);
  const int v = u % per_batch;
  const int slice = v / T;
  int t = v - slice * T;
  const int tile_c = t % p.tiles_c; t /= p.tiles_c;
  const int tile_n = t % p.tiles_n;
  const int tap = t / p.tiles_n;
  const int tx = tap % p.KW, ty = (tap / p.KW) % p.KH, tz = tap / (p.KW * p.KH);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ks = wid / (WAVES_M * WAVES_N);
  const int wm = (wid % (WAVES_M * WAVES_N)) / WAVES_N, wn = wid % WAVES_N;

  // loader coordinates
  const int a_row = tid / A_F4, a_col = (tid % A_F4) * 4;
  const int b_row = tid / B_F4, b_col = (tid % B_F4) * 4;
  const int n_g = tile_n * BMN + a_col;       // first of this thread's 4 dy columns
  const bool n_ok = n_g < p.N;
  const int c_g = tile_c * BNC + b_col;       // first of this thread's 4 input channels
  const bool c_ok = c_g < p.Ctot;
  const int s = (p.nsrc == 2 && c_g >= p.src[0].C) ? 1 : 0;
  SrcP S = p.src[s];
  S.ptr += batch * p.bs_x;
  const float* dyp = p.dy + batch * p.bs_dy;
  const int c_l = c_g - (s ? p.src[0].C : 0);

  const int chunk0 = slice * cps;
  int nchunks = (p.M + BKP - 1) / BKP - chunk0;
  if (nchunks > cps) nchunks = cps;

  f32x4 ra[A_PASSES], rb[B_PASSES];
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const bool do_bias = (p.dbias != nullptr) && tile_c == 0 && tap == 0 && batch == 0;

  // dY rows are linear in the pixel index: a chunk's rows come through a buffer descriptor that starts at the chunk's
  // first row and ends with the tensor — a lane's byte offset inside the chunk is a constant, rows past M fall outside
  // the descriptor and read as zeros, and a load costs no vector instruction (the 64-bit address + zero-pointer select
  // of the plain form: ~6 per load, on the issue port the MFMAs use).  Columns past N read into the next row: their
  // products land in rows of dW that are never stored.
  typedef int i32x4_ __attribute__((ext_vector_type(4)));
  int dyoff[A_PASSES];
#pragma unroll
  for (int j = 0; j < A_PASSES; ++j) dyoff[j] = ((a_row + j * A_RPP) * p.ld_dy + n_g) * 4;
  auto chunk_rsrc = [&](const float* base, int ld, int chunk) {
    const long long row0 = (long long)(chunk0 + chunk) * BKP;
    const long long left = ((long long)p.M - row0) * ld * 4;
    const int records = left <= 0 ? 0 : left > 0x7fffffffll ? 0x7fffffff : (int)left;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + row0 * ld), 0, records, 0x00020000);
  };
  auto load_dy = [&](int chunk) {
    const __amdgpu_buffer_rsrc_t r = chunk_rsrc(dyp, p.ld_dy, chunk);
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j)
      ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, dyoff[j], 0, 0));
  };
  int xoff[B_PASSES];               // LIN == 1: the same for the input rows
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) xoff[j] = ((b_row + j * B_RPP) * S.ld + c_l) * 4;
  // Every chunk decodes its B_PASSES pixel rows from the linear pixel index with fast divisions:
  // straight-line code.  (An incremental walk with `while` wrap-arounds is fewer instructions, but
  // its divergent loops made the compiler put s_waitcnt vmcnt(<=3) in front of every row — the
  // x loads then waited for the dy loads issued four MFMA groups earlier, every chunk.)
  uint32_t wlin[B_PASSES];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) wlin[j] = (uint32_t)chunk0 * BKP + (uint32_t)(b_row + j * B_RPP);
  auto load_x = [&](int chunk) {
    if constexpr (LIN == 1) {
      const __amdgpu_buffer_rsrc_t r = chunk_rsrc(S.ptr, S.ld, chunk);
#pragma unroll
      for (int j = 0; j < B_PASSES; ++j)
        rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, xoff[j], 0, 0));
      return;
    }
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
      const uint32_t m = wlin[j] + (uint32_t)chunk * BKP;
      if constexpr (LIN == 1) {
        const bool ok = c_ok && m < (uint32_t)p.M;
        rb[j] = *reinterpret_cast<const f32x4*>(ok ? S.ptr + (size_t)m * S.ld + c_l : p.zeros);
        continue;
      }
      if constexpr (LIN == 2) {       // OH = IH = 1, one plain source of ID planes x OW tiles
        const uint32_t t = fdiv(m, p.dOW);
        const int ox = (int)(m - t * p.OW);
        const uint32_t ob = fdiv(t, p.dOD);
        const int lz = (int)(t - ob * p.OD) + tz - p.PD;
        const bool ok = c_ok && m < (uint32_t)p.M && (unsigned)lz < (unsigned)p.ID;
        const int pix = ((int)ob * p.ID + lz) * p.OW + ox;
        rb[j] = *reinterpret_cast<const f32x4*>(ok ? S.ptr + (size_t)pix * S.ld + c_l : p.zeros);
        continue;
      }
      const uint32_t q1 = fdiv(m, p.dOW);
      const int ox = (int)(m - q1 * p.OW);
      const uint32_t q2 = fdiv(q1, p.dOH);
      const int oy = (int)(q1 - q2 * p.OH);
      const uint32_t q3 = fdiv(q2, p.dOD);
      const int oz = (int)(q2 - q3 * p.OD);
      const int ob = (int)q3;
      const int lz = oz + tz - p.PD, ly = oy + ty - p.PH, lx = ox + tx - p.PW;
      const bool ok = c_ok && m < (uint32_t)p.M && (unsigned)lz < (unsigned)p.ID &&
                      (unsigned)ly < (unsigned)p.IH && (unsigned)lx < (unsigned)p.IW;
      // (garbage for out-of-range rows: their pointer is replaced by the zero buffer below)
      const int sz = (int)fdiv((uint32_t)(lz + S.oz), S.dfz);
      const int sy = (int)fdiv((uint32_t)(ly + S.oy), S.dfy);
      const int sx = (int)fdiv((uint32_t)(lx + S.ox), S.dfx);
      const int pix = ((ob * S.D + sz) * S.H + sy) * S.W + sx;   // < 2^31 (validated on the host)
      rb[j] = *reinterpret_cast<const f32x4*>(ok ? S.ptr + (size_t)pix * S.ld + c_l : p.zeros);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) {
      *reinterpret_cast<f32x4*>(&Ys[buf][(a_row + j * A_RPP) * BMN + a_col]) = ra[j];
      if (do_bias) bsum += ra[j];
    }
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      *reinterpret_cast<f32x4*>(&Xs[buf][(b_row + j * B_RPP) * BNC + b_col]) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  // two accumulator sets along an operand: its fragments as ONE ds_read_b64 per k-pair — lane i reads channels
  // 2i, 2i + 1 of its pixel, the first feeds accumulator set 0, the second set 1 (MFMA row i of set a is channel
  // 2i + a; the k order and the row order of an MFMA are free).  Half the LDS read instructions: 1-3 % per launch.
  constexpr bool PAIR_A = TM == 2, PAIR_B = TN == 2;
  const int a_base = lh * BMN + wm * TM * 32 + (PAIR_A ? 2 * li : li);
  const int b_base = lh * BNC + wn * TN * 32 + (PAIR_B ? 2 * li : li);

  // fragments of k-pair k2+1 are read from LDS while the MFMAs of k-pair k2 issue; the global
  // loads of the next chunk and their LDS stores are slotted between MFMA groups.
  float af[2][TM], bf[2][TN];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto load_frags = [&](int buf, int k2, int slot) {
    if constexpr (PAIR_A) {
      const f32x2 va = *reinterpret_cast<const f32x2*>(&Ys[buf][a_base + 2 * k2 * BMN]);
      af[slot][0] = va[0]; af[slot][TM - 1] = va[1];
    } else {
#pragma unroll
      for (int a = 0; a < TM; ++a) af[slot][a] = Ys[buf][a_base + 2 * k2 * BMN + a * 32];
    }
    if constexpr (PAIR_B) {
      const f32x2 vb = *reinterpret_cast<const f32x2*>(&Xs[buf][b_base + 2 * k2 * BNC]);
      bf[slot][0] = vb[0]; bf[slot][TN - 1] = vb[1];
    } else {
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[slot][b] = Xs[buf][b_base + 2 * k2 * BNC + b * 32];
    }
  };
  if (nchunks > 0) {
    load_dy(0);
    load_x(0);
    store_chunk(0);
    __syncthreads();
    int buf = 0;
    load_frags(0, ks, 0);
    // `MORE` is a compile-time flag, the last chunk runs after the loop: with a run-time
    // `if (more)` around the loads and the LDS stores the compiler's wait-count pass sees paths
    // on which a load is issued and never consumed, and guards later writes of those registers
    // with s_waitcnt vmcnt(<=3) — the x loads then waited for the dy loads issued four MFMA
    // groups earlier, in every chunk.
    auto chunk_body = [&](int ch, auto more_tag) {
      constexpr bool MORE = decltype(more_tag)::value;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {          // this wave's k-pairs: it * KS + ks
        if constexpr (MORE) {
          if (it == 0) load_dy(ch + 1);
          if (it == NIT / 4) load_x(ch + 1);
          if (it == 3 * NIT / 4) store_chunk(buf ^ 1);
        }
        if (it + 1 < NIT) load_frags(buf, (it + 1) * KS + ks, (it + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[it & 1][a], bf[it & 1][b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (MORE) {
        __syncthreads();
        buf ^= 1;
        load_frags(buf, ks, 0);
      }
    };
    for (int ch = 0; ch + 1 < nchunks; ++ch) chunk_body(ch, std::true_type{});
    chunk_body(nchunks - 1, std::false_type{});
  }

  // ---- combine: float atomics into dwpack[tap][n][c]
  float* dst = p.dwp + batch * p.bs_out + (size_t)tap * p.N * p.Ctot;
  int* my_turn = nullptr;
  if (p.turns != nullptr) {
    // lower slices of a tile have lower block indices: they were dispatched no later than this block
    my_turn = p.turns + (size_t)batch * T + (v - slice * T);
    if (tid == 0)
      while (__hip_atomic_load(my_turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != slice) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int c = PAIR_B ? tile_c * BNC + wn * TN * 32 + 2 * li + b : tile_c * BNC + (wn * TN + b) * 32 + li;
      if (c < p.Ctot) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int n = PAIR_A ? tile_n * BMN + wm * TM * 32 + 2 * row + a : tile_n * BMN + (wm * TM + a) * 32 + row;
          if (n < p.N) atomicAdd(dst + (size_t)n * p.Ctot + c, acc[a][b][r]);
        }
      }
    }
  }

  // ---- bias gradient: column sums of this block's dy rows
  if (do_bias) {
    __syncthreads();
    float* red = &Ys[0][0];  // [A_RPP][BMN]
    *reinterpret_cast<f32x4*>(&red[a_row * BMN + a_col]) = bsum;
    __syncthreads();
    if (tid < BMN) {
      float sum = 0.f;
#pragma unroll 4
      for (int r = 0; r < A_RPP; ++r) sum += red[r * BMN + tid];
      const int n = tile_n * BMN + tid;
      if (n < p.N) atomicAdd(p.dbias + n, sum);
    }
  }
  if (my_turn != nullptr) {
    __threadfence();                                   // this block's additions are performed ...
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(my_turn, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... next slice
  }
}


}  // namespace

// co-resident blocks of a kernel on the current device = CUs x blocks per CU (cached)
static int resident_blocks(const void* fn) {
  struct Entry { const void* fn; int dev; int slots; };
  static Entry cache[16];
  static int used = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  for (int i = 0; i < used; ++i)
    if (cache[i].fn == fn && cache[i].dev == dev) return cache[i].slots;
  int cus = 0, per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess) return 0;
  if (per_cu < 1) per_cu = 1;
  const int slots = cus * per_cu;
  if (used < 16) cache[used++] = Entry{fn, dev, slots};
  return slots;
}

static const float* wgrad_zero_buffer() {
  static const float* cache[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (cache[dev] == nullptr) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_wgrad_zero16)) != hipSuccess) return nullptr;
    cache[dev] = (const float*)ptr;
  }
  return cache[dev];
}

// tiles of one batch element of the weight-gradient launch (n tiles x c tiles x taps)
static int wgrad_tiles(const clx_conv_desc* d) {
  auto wide = [](int n) { return n > 64 && (double)(cdiv(n, 128) * 128) / n <= 1.15; };
  const int Ctot = d->src[0].C + (d->nsrc == 2 ? d->src[1].C : 0);
  return cdiv(d->N, wide(d->N) ? 128 : 64) * cdiv(Ctot, wide(Ctot) ? 128 : 64) * d->KD * d->KH * d->KW;
}

int clx_wgrad_launch(const clx_conv_desc* d, const float* dy, int ld_dy, float* dwpack, float* dbias,
                     int batch, long long bs_x, long long bs_dy, long long bs_out, hipStream_t st) {
  WgradP p;
  p.turns = d->det_turns;
  p.nsrc = d->nsrc;
  for (int s = 0; s < 2; ++s) {
    const clx_src& S = d->src[s < d->nsrc ? s : 0];
    p.src[s] = SrcP{S.ptr, S.C, S.ld, S.D, S.H, S.W, S.oz, S.oy, S.ox, S.fz, S.fy, S.fx,
                    make_fastdiv((uint32_t)(S.fz > 0 ? S.fz : 1)), make_fastdiv((uint32_t)(S.fy > 0 ? S.fy : 1)),
                    make_fastdiv((uint32_t)(S.fx > 0 ? S.fx : 1))};
  }
  p.B = d->B; p.ID = d->ID; p.IH = d->IH; p.IW = d->IW;
  p.KD = d->KD; p.KH = d->KH; p.KW = d->KW;
  p.PD = d->PD; p.PH = d->PH; p.PW = d->PW;
  p.OD = d->ID + 2 * d->PD - d->KD + 1;
  p.OH = d->IH + 2 * d->PH - d->KH + 1;
  p.OW = d->IW + 2 * d->PW - d->KW + 1;
  CLX_REQUIRE(p.OD > 0 && p.OH > 0 && p.OW > 0, "clx_conv_wgrad: empty output");
  const long long M = (long long)d->B * p.OD * p.OH * p.OW;
  CLX_REQUIRE(M < (1ll << 31), "clx_conv_wgrad: too many pixels");
  p.N = d->N; p.M = (int)M;
  p.Ctot = d->src[0].C + (d->nsrc == 2 ? d->src[1].C : 0);
  p.dOW = make_fastdiv(p.OW); p.dOH = make_fastdiv(p.OH); p.dOD = make_fastdiv(p.OD);
  p.dy = dy; p.ld_dy = ld_dy; p.dwp = dwpack; p.dbias = dbias;
  p.bs_x = bs_x; p.bs_dy = bs_dy; p.bs_out = bs_out;
  p.zeros = wgrad_zero_buffer();
  CLX_REQUIRE(p.zeros != nullptr, "clx_conv_wgrad: cannot resolve the device zero buffer");
  p.taps = d->KD * d->KH * d->KW;
  // output pixel m = input pixel m of one plain source?
  static const bool lin_env = getenv("CLX_WGRAD_LINEAR") == nullptr || atoi(getenv("CLX_WGRAD_LINEAR")) != 0;
  const bool linear = lin_env && p.taps == 1 && d->nsrc == 1 && d->PD == 0 && d->PH == 0 && d->PW == 0 &&
                      d->src[0].fz == 1 && d->src[0].fy == 1 && d->src[0].fx == 1 &&
                      d->src[0].oz == 0 && d->src[0].oy == 0 && d->src[0].ox == 0 &&
                      d->src[0].D == p.OD && d->src[0].H == p.OH && d->src[0].W == p.OW;
  // ... or input plane (oz + tz - PD), same tile, of one plain source of single-row planes?
  const bool zlinear = lin_env && !linear && d->KH == 1 && d->KW == 1 && d->nsrc == 1 && d->PH == 0 && d->PW == 0 &&
                       d->IH == 1 && d->src[0].fz == 1 && d->src[0].fy == 1 && d->src[0].fx == 1 &&
                       d->src[0].oz == 0 && d->src[0].oy == 0 && d->src[0].ox == 0 &&
                       d->src[0].D == d->ID && d->src[0].H == 1 && d->src[0].W == p.OW && d->IW == p.OW;
  const int lin_mode = linear ? 1 : zlinear ? 2 : 0;

  // 128-wide tiles unless padding the extent up to a multiple of 128 wastes > 15 % of the MFMAs
  auto wide = [](int n) { return n > 64 && (double)(cdiv(n, 128) * 128) / n <= 1.15; };
  const bool big_n = wide(p.N), big_c = wide(p.Ctot);
  const int bmn = big_n ? 128 : 64, bnc = big_c ? 128 : 64;
  p.tiles_n = cdiv(p.N, bmn);
  p.tiles_c = cdiv(p.Ctot, bnc);
  const int T = p.tiles_n * p.tiles_c * p.taps;
  const int Tall = T * batch;
  if (p.turns != nullptr && hipMemsetAsync(p.turns, 0, (size_t)Tall * sizeof(int), st) != hipSuccess) {
    clx_set_error("clx_conv_wgrad: clearing the turn counters failed");
    return CLX_ERR_LAUNCH;
  }
  const int BKP = big_n && big_c ? bkp<128, 128>() : 32;
  const int total_chunks = cdiv(p.M, BKP);
  // Split-K so that the grid is a whole number of "rounds" of co-resident blocks: a grid of
  // 4 rounds + a few blocks would run 5 rounds (the tail alone costs 20 %).
  const void* fn = big_n && big_c ? (const void*)conv_wgrad_kernel<128, 128, 2, 2>
                   : big_n        ? (const void*)conv_wgrad_kernel<128, 64, 4, 1>
                   : big_c        ? (const void*)conv_wgrad_kernel<64, 128, 1, 4>
                                  : (const void*)conv_wgrad_kernel<64, 64, 2, 2>;
  const int slots = resident_blocks(fn);
  CLX_REQUIRE(slots > 0, "clx_conv_wgrad: occupancy query failed");
  int nslices = 1;
  if (Tall < 4 * slots) {
    const int rounds = Tall <= slots ? (Tall * 4 <= slots ? 1 : 2) : 4;
    nslices = rounds * slots / Tall;
    if (Tall * 4 <= slots) nslices = 4 * slots / Tall;   // tiny output: still aim for >= 4 blocks per slot
  }
  const int max_slices = total_chunks / 8 > 0 ? total_chunks / 8 : 1;
  if (nslices > max_slices) nslices = max_slices;
  if (nslices < 1) nslices = 1;
  p.chunks_per_slice = cdiv(total_chunks, nslices);
  p.nslices = cdiv(total_chunks, p.chunks_per_slice);
  // the last, partial round of co-resident blocks: its batches get more, shorter slices
  p.batch_split = batch; p.nslices_tail = p.nslices; p.chunks_per_slice_tail = p.chunks_per_slice;
  {
    const int per_batch = T * p.nslices;
    const long long blocks = (long long)per_batch * batch;
    const int full = (int)(blocks / slots);
    const long long rest = blocks - (long long)full * slots;
    static const bool enabled = getenv("CLX_WGRAD_TAIL_SPLIT") == nullptr || atoi(getenv("CLX_WGRAD_TAIL_SPLIT")) != 0;
    if (enabled && batch > 1 && full >= 1 && rest > 0 && rest * 10 < (long long)slots * 9) {
      int split = (int)(((long long)full * slots) / per_batch);          // batches that fit the full rounds
      if (split >= batch) split = batch - 1;
      const int tail_tiles = (batch - split) * T;
      int ns_tail = slots / tail_tiles;
      const int max_tail = total_chunks / 4 > 0 ? total_chunks / 4 : 1;
      if (ns_tail > max_tail) ns_tail = max_tail;
      if (split > 0 && ns_tail > p.nslices) {
        p.batch_split = split;
        p.chunks_per_slice_tail = cdiv(total_chunks, ns_tail);
        p.nslices_tail = cdiv(total_chunks, p.chunks_per_slice_tail);
      }
    }
  }
  p.n_main = T * p.nslices * p.batch_split;
  const dim3 grid(p.n_main + T * p.nslices_tail * (batch - p.batch_split)), block(256);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (clx_prof_enabled())
    clx_prof_events(CLX_PROF_WGRAD, 2.0 * p.M * p.N * p.Ctot * p.taps * batch, &e0, &e1);
#define CLX_WG(BMN_, BNC_, WM_, WN_)                                                                           \
  do {                                                                                                         \
    if (lin_mode == 1) CLX_LAUNCH_TIMED((conv_wgrad_kernel<BMN_, BNC_, WM_, WN_, 1, 1>), grid, block, st, e0, e1, p);      \
    else if (lin_mode == 2) CLX_LAUNCH_TIMED((conv_wgrad_kernel<BMN_, BNC_, WM_, WN_, 1, 2>), grid, block, st, e0, e1, p); \
    else CLX_LAUNCH_TIMED((conv_wgrad_kernel<BMN_, BNC_, WM_, WN_>), grid, block, st, e0, e1, p);                          \
  } while (0)
  if (big_n && big_c) CLX_WG(128, 128, 2, 2);
  else if (big_n) CLX_WG(128, 64, 4, 1);
  else if (big_c) CLX_WG(64, 128, 1, 4);
  else CLX_WG(64, 64, 2, 2);
#undef CLX_WG
  return CLX_OK;
}

extern "C" int clx_conv_wgrad(const clx_conv_desc* d, const float* dy, int ld_dy,
                              float* dwpack, float* dbias, clx_stream stream) {
  CLX_REQUIRE(d && dy && dwpack, "clx_conv_wgrad: null pointer");
  CLX_REQUIRE(d->nsrc == 1 || d->nsrc == 2, "clx_conv_wgrad: nsrc must be 1 or 2");
  CLX_REQUIRE(d->N > 0 && d->N % 4 == 0 && ld_dy % 4 == 0 && ld_dy >= d->N,
              "clx_conv_wgrad: N and ld_dy must be multiples of 4 (N=%d ld_dy=%d)", d->N, ld_dy);
  CLX_REQUIRE(((uintptr_t)dy & 15) == 0, "clx_conv_wgrad: dy must be 16-byte aligned");
  for (int s = 0; s < d->nsrc; ++s) {
    const clx_src& S = d->src[s];
    CLX_REQUIRE(S.ptr && S.C > 0 && S.C % 4 == 0 && S.ld % 4 == 0 && S.ld >= S.C &&
                    ((uintptr_t)S.ptr & 15) == 0,
                "clx_conv_wgrad: bad source %d", s);
    CLX_REQUIRE(S.fz >= 1 && S.fy >= 1 && S.fx >= 1 && S.oz >= 0 && S.oy >= 0 && S.ox >= 0,
                "clx_conv_wgrad: bad crop/upsample of source %d", s);
    CLX_REQUIRE((long long)d->B * S.D * S.H * S.W < (1ll << 31),
                "clx_conv_wgrad: source %d has too many pixels", s);
  }
  CLX_REQUIRE(d->algo == CLX_ALGO_DIRECT || d->algo == CLX_ALGO_WINOGRAD || d->algo == CLX_ALGO_WINOGRAD4,
              "clx_conv_wgrad: bad algo");
  if (d->algo != CLX_ALGO_DIRECT) return clx_wino_wgrad(d, dy, ld_dy, dwpack, dbias, (hipStream_t)stream);
  // opt-in precision: a 1x1 layer over one plain source, both channel counts multiples of 128 — planes of x (left by the
  // forward pass, or split here) and of dY (split here; the bias gradient is that pass's column sums)
  if (d->precision == CLX_PREC_F32X3BF16 && d->aplanes != nullptr && d->dyplanes != nullptr && d->det_turns == nullptr &&
      d->nsrc == 1 && d->KD == 1 && d->KH == 1 && d->KW == 1 && d->PD == 0 && d->PH == 0 && d->PW == 0 &&
      d->N % 128 == 0 && d->src[0].C % 128 == 0 && ld_dy >= d->N &&
      (long long)d->B * d->ID * d->IH * d->IW * (d->N > d->src[0].C ? d->N : d->src[0].C) * 6 < (1ll << 32) - (1 << 24)) {
    const clx_src& S = d->src[0];
    if (S.fz == 1 && S.fy == 1 && S.fx == 1 && S.oz == 0 && S.oy == 0 && S.ox == 0 && S.D == d->ID && S.H == d->IH && S.W == d->IW) {
      const long long M = (long long)d->B * d->ID * d->IH * d->IW;
      int rc = 0;
      if (!d->aplanes_valid) rc = clx_sp_split(S.ptr, S.ld, M, S.C, d->aplanes, nullptr, 0, (hipStream_t)stream);
      if (rc) return rc;
      if (!d->dyplanes_valid) rc = clx_sp_split(dy, ld_dy, M, d->N, d->dyplanes, dbias, d->N, (hipStream_t)stream);
      if (rc) return rc;
      rc = clx_sp_wgrad_launch(d->dyplanes, d->aplanes, M, d->N, S.C, 1, 0, 0, 0, dwpack, S.C, (hipStream_t)stream);
      if (rc) return rc;
      CLX_CHECK_LAUNCH("clx_conv_wgrad(split precision)");
      return CLX_OK;
    }
  }
  if (clx_smallc_applicable(d) && d->det_turns == nullptr) {
    clx_smallc_wgrad(d, dy, ld_dy, dwpack, dbias, (hipStream_t)stream);
    CLX_CHECK_LAUNCH("clx_conv_wgrad(small-channel)");
    return CLX_OK;
  }
  const int rc = clx_wgrad_launch(d, dy, ld_dy, dwpack, dbias, 1, 0, 0, 0, (hipStream_t)stream);
  if (rc) return rc;
  CLX_CHECK_LAUNCH("clx_conv_wgrad");
  return CLX_OK;
}

// Winograd layers launch a^2 batched products (xi), the sub-pixel / direct layers one: an upper bound over the
// forms clx_conv_wgrad can take for this descriptor
extern "C" size_t clx_conv_wgrad_turns_bytes(const clx_conv_desc* d) {
  if (d == nullptr) return 0;
  clx_conv_desc one = *d;
  one.KD = one.KH = one.KW = 1;
  const size_t direct = (size_t)wgrad_tiles(d), per_xi = (size_t)wgrad_tiles(&one) * d->KD;
  const size_t most = direct > 36 * per_xi ? direct : 36 * per_xi;
  return (most + 64) * sizeof(int);
}
