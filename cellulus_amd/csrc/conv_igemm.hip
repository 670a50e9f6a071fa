// f32 MFMA implicit-GEMM convolution for gfx950 (forward and data gradient).
//
//   out[m][n] = act( sum_{tap, c} in[pix(m) (+) tap][c] * wpack[n][tap][c] + bias[n] )
//
// GEMM view: M = B*OD*OH*OW output pixels, N = output channels,
// K = taps x Ctot.  A rows are gathered on the fly (valid / zero-padded conv,
// optional crop offset and nearest upsample per source, two concatenated
// sources), B rows are the K-contiguous packed weights.  Both tiles are staged
// through LDS as [rows][32 + 4 pad] so every MFMA operand fragment is one
// conflict-free ds_read_b128: lane (i = l&31, h = l>>5) reads k = 8q+4h..+3 of
// row i and feeds the 4 components to 4 v_mfma_f32_32x32x2_f32 (the k order
// inside a chunk is permuted identically for A and B, which a sum allows).
//
// Replaces nn.Conv{2,3}d(+ReLU) forward/backward-data of the reference U-Net
// (cellulus/models/unet.py:24-63, funlib ConvPass) — exact f32 arithmetic.
#include "clx_common.h"
#include <stdlib.h>

#ifdef IG_STAMP
// diagnostic build (tools/build_variant.sh ... -DIG_STAMP): per block, wall-clock stamps (100 MHz counter) at start, first
// MFMA, end of the K loop, end of the block, and the hardware id — read back with clx_debug_stamps
__device__ unsigned long long g_stamps[8 * 32768];
extern "C" int clx_debug_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#define STAMP(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 32768) g_stamps[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

// Shader clock under load: the middle block of every launch adds the shader-clock ticks (s_memtime) and the 100-MHz wall-clock
// ticks (s_memrealtime) of its own life to two counters — their ratio is the clock the matrix cores actually ran at
// (bench.py: roofline.shader_clock_mhz; the 157.3 TFLOP/s peak is 2.4 GHz).  Two atomics per launch.
__device__ unsigned long long g_clk_ticks[2];
extern "C" int clx_profile_clock(double* shader_ticks, double* wall_ticks_100mhz, int reset) {
  unsigned long long h[2] = {0ull, 0ull};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk_ticks), sizeof(h)) != hipSuccess) {
    clx_set_error("clx_profile_clock: copy failed");
    return CLX_ERR_LAUNCH;
  }
  double sc = 0.0, sw = 0.0;                     // the split-precision products' counters (gemm_sp.hip)
  if (clx_sp_clock_read(&sc, &sw, reset) != CLX_OK) {
    clx_set_error("clx_profile_clock: copy failed");
    return CLX_ERR_LAUNCH;
  }
  if (shader_ticks) *shader_ticks = (double)h[0] + sc;
  if (wall_ticks_100mhz) *wall_ticks_100mhz = (double)h[1] + sw;
  if (reset) {
    const unsigned long long z[2] = {0ull, 0ull};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_clk_ticks), z, sizeof(z)) != hipSuccess) {
      clx_set_error("clx_profile_clock: reset failed");
      return CLX_ERR_LAUNCH;
    }
  }
  return CLX_OK;
}

#ifndef CLX_FLUSH_FORM
#define CLX_FLUSH_FORM 0
#endif

namespace {

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int BK = 32;       // K chunk (floats)
// LDS row of the 128x128 kernel: one K chunk + 4 floats of padding (conflict-free b128 reads).
// The 128x64 kernel stores unpadded rows and XOR-swizzles the 16-byte column index with
// (row >> 1) & 7 instead: ds_read_b128 is serviced in 16-lane groups over the 16 16-byte slots of a
// 256-byte line (MI355X_MICROARCH.md, LDS); a fragment read touches 16 rows at one column, slot =
// 8 (row & 1) + (col ^ (row >> 1) & 7) — all distinct; a store's 8-lane group is one row, its 8
// columns permuted.  32 instead of 36 floats per row = 48 KB per block: THREE blocks per CU
// (132 VGPRs), which is what the short-K, narrow-N layers of the 3-D network need (26.5 -> 25.3 ms
// per step); on the 128x128 kernel (two blocks per CU either way) the swizzle costs 3 %.
template <int BN> constexpr int lds_ld() { return BN == 64 ? 32 : 36; }

struct SrcP {
  const float* ptr;
  int C, ld, D, H, W, oz, oy, ox, fz, fy, fx;
};

struct ConvP {
  int nsrc;
  SrcP src[2];
  int B, ID, IH, IW, KD, KH, KW, PD, PH, PW, OD, OH, OW;
  int N, M, Ctot, Ktot;
  FastDiv dOW, dOH, dOD;
  const float* wpack;
  const float* bias;
  const float* mask;
  const float* zeros;   // 16 zero bytes in global memory (target of out-of-range loads)
  float* out;
  unsigned int* gate_out;          // ReLU gates of `out` as bits (optional)
  const unsigned int* mask_bits;   // gate bits read instead of `mask` (optional)
  int ld_gate, ld_mask_bits;
  int relu, ld_mask, ld_out, accumulate;
  int nbm, nbn;
  long long bs_in, bs_w, bs_out;   // per-batch strides in floats (gridDim.y batches; 0 = none)
  int flush_every;                 // chunks of 32 products per fresh accumulator (two-level summation over K); 0 = never
};

// FAST: no zero rows exist (no padding; channel counts are multiples of the 32-wide K chunk) and every offset fits 32 bits:
// rows past M / N are CLAMPED into the tensor instead of zeroed (their results are never stored), so a staged load is
// `uniform base + 32-bit lane offset` — the chunk's channel offset moves the scalar base — and costs no vector
// instruction, where the general form spends ~6 per load on a 64-bit address and the zero-pointer select (the weight
// gradient gained 9 % from losing its per-row index arithmetic; the K loop's other instructions share the issue port
// with the MFMAs).
// FAST = 2: the same for PADDED convolutions (the data gradients): the rows are fetched with buffer loads whose range check
// returns zeros for the offset 0x80000000 that a row outside the image gets (tensor < 2 GB).
template <int BM, int BN, int WAVES_M, int WAVES_N, int FAST = 0>
__global__ __launch_bounds__(256, BN == 64 ? 3 : 2) void conv_igemm_kernel(const ConvP p) {
  constexpr int LDS_LD = lds_ld<BN>();
  constexpr bool SWZ = LDS_LD == 32;
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  constexpr int A_PASSES = BM / 32;
  constexpr int B_PASSES = BN / 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(TM * TN >= 2, "the accumulator flush relies on another tile's MFMAs between a tile's last MFMA and its flush");

  // one LDS arena: A/B double buffers during the K loop, the C tile in the epilogue
  // C tile rows: 16 lanes read one 64-float row of the narrow tile, and ds_read_b128 serves the lane groups
  // {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}: two consecutive rows must present complementary 16-byte slots,
  // i.e. a row stride of a whole 256-byte line (with + 4 floats a tenth of this kernel's LDS cycles were
  // conflicts of these reads); the 128-wide tile keeps a lane group inside one row, where the pad is harmless
  constexpr int LDC = BN == 64 ? BN : BN + 4;
  constexpr int SMEM_AB = 2 * (BM + BN) * LDS_LD;
  constexpr int SMEM_C = BM * LDC;
  __shared__ float smem[SMEM_AB > SMEM_C ? SMEM_AB : SMEM_C];
  float (*As)[BM * LDS_LD] = reinterpret_cast<float (*)[BM * LDS_LD]>(smem);
  float (*Bs)[BN * LDS_LD] = reinterpret_cast<float (*)[BN * LDS_LD]>(smem + 2 * BM * LDS_LD);

  const bool clk_block = blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && threadIdx.x == 0;   // mid-launch
  unsigned long long clk_c0 = 0, clk_w0 = 0;
  if (clk_block) { clk_c0 = clock64(); clk_w0 = wall_clock64(); }
  STAMP(0);
#ifdef IG_STAMP
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 32768) {
    unsigned int hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_stamps[blockIdx.x * 8 + 4] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  const int v = xcd_remap(blockIdx.x, p.nbm * p.nbn);
  const int tile_n = v % p.nbn;
  const int tile_m = v / p.nbn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int lrow = tid >> 3;        // 0..31
  const int lcol = (tid & 7) * 4;   // float offset inside the 32-wide chunk
  const int lcol_lds = SWZ ? ((tid & 7) ^ ((lrow >> 1) & 7)) * 4 : lcol;   // its place in the LDS row

  // ---- decode this thread's A rows (output pixels) once
  int rb[A_PASSES], rz[A_PASSES], ry[A_PASSES], rx[A_PASSES];
#pragma unroll
  for (int j = 0; j < A_PASSES; ++j) {
    uint32_t m = (uint32_t)(m0 + lrow + 32 * j);
    if (FAST == 1 && m >= (uint32_t)p.M) m = (uint32_t)p.M - 1u;
    if (m >= (uint32_t)p.M) {
      rb[j] = -1; rz[j] = ry[j] = rx[j] = 0;
    } else {
      uint32_t t = fdiv(m, p.dOW);
      rx[j] = (int)(m - t * p.OW);
      uint32_t t2 = fdiv(t, p.dOH);
      ry[j] = (int)(t - t2 * p.OH);
      uint32_t t3 = fdiv(t2, p.dOD);
      rz[j] = (int)(t2 - t3 * p.OD);
      rb[j] = (int)t3;
    }
  }
  // ---- B rows (output channels)
  const float* wrow[B_PASSES];
  uint32_t woff[B_PASSES];          // FAST: byte offset of the row + this lane's column inside a chunk
  const float* const wbase = p.wpack + blockIdx.y * p.bs_w;
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) {
    int n = n0 + lrow + 32 * j;
    wrow[j] = (n < p.N) ? wbase + (size_t)n * p.Ktot : nullptr;
    woff[j] = ((uint32_t)(n < p.N ? n : p.N - 1) * (uint32_t)p.Ktot + (uint32_t)lcol) * 4u;      // bytes
  }

  // ---- K-loop state: (tap, source, channel chunk)
  int tz = 0, ty = 0, tx = 0, tap = 0, s = 0, c0 = 0;
  long long aoff[A_PASSES];   // float offset of the row in the current source, -1 = zero row
  uint32_t aoff32[A_PASSES];  // FAST: the same + this lane's column inside a chunk, always valid
  const float* sptr = p.src[0].ptr + blockIdx.y * p.bs_in;
  int sC = p.src[0].C, cbase = 0;

  auto set_source = [&]() {
    const SrcP& S = p.src[s];
    sptr = S.ptr + blockIdx.y * p.bs_in;
    sC = S.C;
    cbase = (s == 0) ? 0 : p.src[0].C;
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) {
      const int lz = rz[j] + tz - p.PD, ly = ry[j] + ty - p.PH, lx = rx[j] + tx - p.PW;
      const bool ok = rb[j] >= 0 && (unsigned)lz < (unsigned)p.ID &&
                      (unsigned)ly < (unsigned)p.IH && (unsigned)lx < (unsigned)p.IW;
      int sz = lz + S.oz, sy = ly + S.oy, sx = lx + S.ox;
      if (S.fz > 1) sz /= S.fz;
      if (S.fy > 1) sy /= S.fy;
      if (S.fx > 1) sx /= S.fx;
      const long long pix = (((long long)rb[j] * S.D + sz) * S.H + sy) * S.W + sx;
      aoff[j] = ok ? pix * S.ld : -1;
      if (FAST == 1) aoff32[j] = ((uint32_t)pix * (uint32_t)S.ld + (uint32_t)lcol) * 4u;      // bytes
      if (FAST == 2) aoff32[j] = ok ? ((uint32_t)pix * (uint32_t)S.ld + (uint32_t)lcol) * 4u : 0x80000000u;
    }
  };

  f32x4 ra[A_PASSES], rw[B_PASSES];
  auto load_a = [&]() {
    if constexpr (FAST == 2) {
      typedef int i32x4_ __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sptr), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int j = 0; j < A_PASSES; ++j) {
        const i32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)aoff32[j], c0 * 4, 0);
        ra[j] = __builtin_bit_cast(f32x4, v);
      }
      return;
    }
    if constexpr (FAST == 1) {
      const char* const base = reinterpret_cast<const char*>(sptr + c0);             // uniform
#pragma unroll
      for (int j = 0; j < A_PASSES; ++j) ra[j] = *reinterpret_cast<const f32x4*>(base + aoff32[j]);
      return;
    }
    const int c = c0 + lcol;
    const bool cv = c < sC;
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) {
      // out-of-range rows/channels read 16 zero bytes: no branch, and no use of the
      // loaded value before the LDS store (a select here would force vmcnt(0) at once)
      const bool ok = cv && aoff[j] >= 0;
      ra[j] = *reinterpret_cast<const f32x4*>(ok ? sptr + aoff[j] + c : p.zeros);
    }
  };
  auto load_b = [&]() {
    if constexpr (FAST != 0) {
      const char* const base = reinterpret_cast<const char*>(wbase + (tap * p.Ctot + cbase + c0));      // uniform
#pragma unroll
      for (int j = 0; j < B_PASSES; ++j) rw[j] = *reinterpret_cast<const f32x4*>(base + woff[j]);
      return;
    }
    const int c = c0 + lcol;
    const bool cv = c < sC;
    const int kflat = tap * p.Ctot + cbase + c;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
      const bool ok = cv && wrow[j] != nullptr;
      rw[j] = *reinterpret_cast<const f32x4*>(ok ? wrow[j] + kflat : p.zeros);
    }
  };
  auto load_chunk = [&]() { load_a(); load_b(); };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j)
      *reinterpret_cast<f32x4*>(&As[buf][(lrow + 32 * j) * LDS_LD + lcol_lds]) = ra[j];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + 32 * j) * LDS_LD + lcol_lds]) = rw[j];
  };
  // returns false when the K loop is exhausted
  auto advance = [&]() -> bool {
    c0 += BK;
    if (c0 < sC) return true;
    c0 = 0;
    ++s;
    if (s == p.nsrc) {
      s = 0;
      ++tap;
      if (++tx == p.KW) { tx = 0; if (++ty == p.KH) { ty = 0; ++tz; } }
      if (tz == p.KD) return false;
    }
    set_source();
    return true;
  };

  // Two-level summation over K: `acc` holds the products of ONE chunk of 32, `tot` the chunks before it.  One
  // accumulator walking the whole contraction rounds every product against a partial sum that keeps growing when the
  // terms are coherent — post-ReLU activations are non-negative —: measured on the benchmark network at a trained
  // network's output scale, 1.6x (K = 256) to 3.1x (K = 2304) the rms error of the reference's CPU path (oneDNN blocks
  // its contraction), and the whole of the network's excess distance from the float64 oracle (tools/exp/
  // layer_profile_error.py).  With a fresh accumulator per chunk the error of a K = 256 layer is 0.45x
  // the single chain's (emulated in float64 on real activations).  The adds of a tile's 16 registers are issued in
  // front of the tile's first MFMA of the next chunk, which starts from zero: they read results that completed three
  // MFMAs ago and hide under the MFMAs around them.
  f32x16 acc[TM][TN], tot[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.f; tot[a][b][r] = 0.f; }

  const int li = lane & 31, lh = lane >> 5;
  // bias of this lane's accumulator columns: fetched here, used in the epilogue
  float bias_v[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int nb_ = n0 + (wn * TN + b) * 32 + li;
    bias_v[b] = (p.bias && nb_ < p.N) ? p.bias[nb_] : 0.f;
  }
  const int a_base = (wm * TM * 32 + li) * LDS_LD + (SWZ ? 0 : 4 * lh);
  const int b_base = (wn * TN * 32 + li) * LDS_LD + (SWZ ? 0 : 4 * lh);
  // k-group q reads 16-byte column 2q + lh of its row, swizzled: (2q + lh) ^ (row >> 1 & 7)
  const int fsw = lh ^ ((li >> 1) & 7);

  set_source();
  load_chunk();
  store_chunk(0);
  __syncthreads();
  STAMP(1);

  int buf = 0;
  bool more = advance();
  // MFMA operand fragments are double-buffered in registers: the ds_read_b128s of
  // k-group q+1 are in flight while the 4*TM*TN MFMAs of group q issue.
  f32x4 af[2][TM], bf[2][TN];
  auto load_frags = [&](int b, int q, int slot) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
      af[slot][a] = *reinterpret_cast<const f32x4*>(&As[b][a_base + a * 32 * LDS_LD + (SWZ ? 4 * ((2 * q) ^ fsw) : 8 * q)]);
#pragma unroll
    for (int c = 0; c < TN; ++c)
      bf[slot][c] = *reinterpret_cast<const f32x4*>(&Bs[b][b_base + c * 32 * LDS_LD + (SWZ ? 4 * ((2 * q) ^ fsw) : 8 * q)]);
  };
  auto mfma_group = [&](int slot) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int c = 0; c < TN; ++c)
          acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][a][e], bf[slot][c][e], acc[a][c], 0, 0, 0);
  };
  // the first group of a chunk: every tile's previous chunk goes into `tot` and the tile restarts from zero, IN PLACE —
  // spelled as instructions: written as `tot += acc; acc = mfma(a, b, 0)` the compiler rotated the adds to the bottom of
  // the loop behind a COPY of the 64 accumulator registers (a third set: 338 registers wanted, 82 spilled, 124 -> 69
  // TFLOP/s).  The tile's 32 instructions sit in front of its first MFMA and under the MFMAs of the tiles around it.
  // CLX_IGEMM_FLUSH (read once, default 2): chunks per fresh accumulator — 1: every 32 products; 2: every 64 (half the
  // flush instructions, the same error on real activations: emulation 2.1e-8 / 2.5e-8 rms against 4.6e-8 for one
  // chain at K = 256); 0: one chain over all of K (rounds 1-4)
  int since_flush = 0;
  auto mfma_group_first = [&](int slot) {
    const bool flush = p.flush_every > 0 && ++since_flush >= p.flush_every;
    if (flush) since_flush = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int c = 0; c < TN; ++c) {
          if (e == 0 && flush) {
            // (the asm block reads MFMA results as VALU operands, and nobody inserts the MFMA -> VALU wait states for an
            //  asm block: 18 for a 16-pass v_mfma_f32_32x32x2_f32.  This tile's last MFMA is at least TM * TN - 1 MFMAs of the
            //  other tiles back — the static_assert below keeps it that way — and the s_nop covers the rest for the first tile.)
            if (a == 0 && c == 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 1" ::: "memory");
#pragma unroll
#if CLX_FLUSH_FORM == 1        // four 32-bit instructions per pair (the first version of this flush)
            for (int r = 0; r < 16; ++r) {
              float t = tot[a][c][r], x = acc[a][c][r];
              asm volatile("v_add_f32 %0, %0, %1\n\tv_mov_b32 %1, 0" : "+v"(t), "+v"(x));
              tot[a][c][r] = t;
              acc[a][c][r] = x;
            }
#else                          // one packed add + one 64-bit move per pair
            for (int r = 0; r < 16; r += 2) {
              typedef float f32x2 __attribute__((ext_vector_type(2)));
              f32x2 t = {tot[a][c][r], tot[a][c][r + 1]}, x = {acc[a][c][r], acc[a][c][r + 1]};
              asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_mov_b64 %1, 0" : "+v"(t), "+v"(x));
              tot[a][c][r] = t[0]; tot[a][c][r + 1] = t[1];
              acc[a][c][r] = x[0]; acc[a][c][r + 1] = x[1];
            }
#endif
          }
          acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][a][e], bf[slot][c][e], acc[a][c], 0, 0, 0);
        }
  };
  // Per chunk: 4 groups of 4*TM*TN MFMAs.  Everything else is slotted between the
  // groups so the matrix pipe never waits for a long non-MFMA stretch: global loads of
  // the next chunk before groups 0/1, their LDS stores (other buffer) before group 3,
  // operand fragments one group ahead.  Only barrier + first fragment read are exposed.
  // The steady-state loop has NO conditional around the loads and stores: with `if (more)`
  // around each of them the compiler's wait-count pass sees paths on which a load was issued
  // and never consumed, and guards every later write of those registers with s_waitcnt
  // vmcnt(<=3) — the B loads then waited for the A loads issued one MFMA group earlier, every
  // chunk.  The last chunk (nothing left to prefetch) runs after the loop.
  load_frags(buf, 0, 0);
#ifdef IG_STAMP
  unsigned long long stamp_prev_ = wall_clock64(), stamp_max_ = 0, stamp_min_ = ~0ull;
#endif
  while (more) {
    load_a();
    load_frags(buf, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group_first(0);
    __builtin_amdgcn_sched_barrier(0);
    load_b();
    load_frags(buf, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(1);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(buf, 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(0);
    __builtin_amdgcn_sched_barrier(0);
    store_chunk(buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(1);
    __builtin_amdgcn_sched_barrier(0);     // keep the barrier BEHIND the last group: the MFMAs
    __syncthreads();                       // already issued absorb the skew between the waves
#ifdef IG_STAMP
    {
      const unsigned long long now_ = wall_clock64();
      const unsigned long long dt_ = now_ - stamp_prev_;
      stamp_prev_ = now_;
      if (dt_ > stamp_max_) stamp_max_ = dt_;
      if (dt_ < stamp_min_) stamp_min_ = dt_;
    }
#endif
    buf ^= 1;
    load_frags(buf, 0, 0);
    more = advance();
  }
  load_frags(buf, 1, 1);
  __builtin_amdgcn_sched_barrier(0);
  mfma_group_first(0);
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
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] += tot[a][b];

  STAMP(2);
#ifdef IG_STAMP
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 32768) g_stamps[blockIdx.x * 8 + 7] = (stamp_max_ << 32) | (stamp_min_ & 0xffffffffull);
#endif
  // ---- epilogue: bias in registers, transpose through LDS so that every lane stores (and reads the ReLU-gate mask
  // as) 16-byte channel runs of one output pixel.  Written in PHASES over eight rows-of-four at a time — all LDS
  // reads, then each optional operand (previous output, gate bits, float mask) as one batch of loads, then the
  // arithmetic, then the stores: with the options tested inside one loop body the compiler put an s_waitcnt vmcnt(0)
  // behind every optional load (on gfx9 that also waits for the STORES before it) and never had two LDS reads in
  // flight — 3.4-4.9 us of a 43-us tile at K = 256, plus 1.5-5 us for four bias loads waited for one by one
  // (tools/exp/tile_stamps.py); the bias is now fetched before the K loop.
  // Plain products (no bias / ReLU / gates / accumulate: the per-xi products of the Winograd layers) on whole tiles
  // leave the accumulators as they are — per register two full 128-byte lines, no LDS round trip, no barriers
  // (+ 1 % on the step.  With bias / ReLU / gate words by ballot / gate bits folded into this path as well the step was
  //  1.5 % SLOWER than with the transposed path below for those launches: 225.0 against 228.3 crops/s)
  if (!p.bias && !p.relu && !p.accumulate && !p.mask && !p.mask_bits && !p.gate_out && m0 + BM <= p.M && n0 + BN <= p.N) {
    float* const ob = p.out + blockIdx.y * p.bs_out;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = n0 + (wn * TN + b) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          ob[(size_t)row * p.ld_out + col] = acc[a][b][r];
        }
      }
    STAMP(3);
    if (clk_block) { atomicAdd(&g_clk_ticks[0], clock64() - clk_c0); atomicAdd(&g_clk_ticks[1], wall_clock64() - clk_w0); }
    return;
  }
  __syncthreads();
  STAMP(5);
  float* Cs = smem;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = (wn * TN + b) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        Cs[row * LDC + col] = acc[a][b][r] + bias_v[b];
      }
    }
  }
  __syncthreads();
  STAMP(6);
  constexpr int F4_PER_ROW = BN / 4;
  constexpr int ITERS = BM * F4_PER_ROW / 256;
  constexpr int PH = 8;                       // rows-of-four per phase
  static_assert(ITERS % PH == 0, "phases");
  const int c4 = (tid % F4_PER_ROW) * 4;      // idx % F4_PER_ROW does not depend on the iteration (256 % F4_PER_ROW == 0)
  const int n = n0 + c4;
  const bool n_live = n < p.N;
  const bool full4 = n + 3 < p.N;
#pragma unroll 1
  for (int h0 = 0; h0 < ITERS; h0 += PH) {
    f32x4 val[PH];
    int mrow[PH];
    bool live[PH];
#pragma unroll
    for (int j = 0; j < PH; ++j) {
      const int row = (tid + 256 * (h0 + j)) / F4_PER_ROW;
      mrow[j] = m0 + row;
      live[j] = mrow[j] < p.M && n_live;
      val[j] = *reinterpret_cast<const f32x4*>(&Cs[row * LDC + c4]);          // (inside the tile: always readable)
    }
    float* const out_base = p.out + blockIdx.y * p.bs_out + n;
    if (p.accumulate) {         // out = act(conv + bias + out): n + 3 < ld_out always (ld_out % 4 == 0)
      f32x4 prev[PH];
#pragma unroll
      for (int j = 0; j < PH; ++j)
        prev[j] = *reinterpret_cast<const f32x4*>(live[j] ? out_base + (size_t)mrow[j] * p.ld_out : p.zeros);
#pragma unroll
      for (int j = 0; j < PH; ++j) val[j] += prev[j];
    }
    if (p.relu) {
#pragma unroll
      for (int j = 0; j < PH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) val[j][e] = fmaxf(val[j][e], 0.f);
    }
    if (p.mask_bits) {          // gate of the producer's ReLU, one bit per channel
      unsigned int w[PH];
#pragma unroll
      for (int j = 0; j < PH; ++j)
        w[j] = live[j] ? p.mask_bits[(size_t)mrow[j] * p.ld_mask_bits + (n >> 5)] : 0u;
#pragma unroll
      for (int j = 0; j < PH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) val[j][e] = ((w[j] >> ((n & 31) + e)) & 1u) ? val[j][e] : 0.f;
    }
    if (p.gate_out) {
      // eight consecutive lanes hold the 32 channels of one word (F4_PER_ROW is 16 or 32, so a group of
      // eight never straddles a row): OR their nibbles together, the first lane of the group stores
#pragma unroll
      for (int j = 0; j < PH; ++j) {
        unsigned int nib = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) nib |= (live[j] && n + e < p.N && val[j][e] > 0.f) ? (1u << e) : 0u;
        unsigned int word = nib << (4 * (lane & 7));
        word |= __shfl_xor(word, 1, 64);
        word |= __shfl_xor(word, 2, 64);
        word |= __shfl_xor(word, 4, 64);
        if ((lane & 7) == 0 && mrow[j] < p.M && n < p.ld_out) p.gate_out[(size_t)mrow[j] * p.ld_gate + (n >> 5)] = word;
      }
    }
    if (p.mask && full4) {
      f32x4 mk[PH];
#pragma unroll
      for (int j = 0; j < PH; ++j)
        mk[j] = *reinterpret_cast<const f32x4*>(live[j] ? p.mask + (size_t)mrow[j] * p.ld_mask + n : p.zeros);
#pragma unroll
      for (int j = 0; j < PH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) val[j][e] = (!live[j] || mk[j][e] > 0.f) ? val[j][e] : 0.f;
    }
    if (full4) {
#pragma unroll
      for (int j = 0; j < PH; ++j)
        if (live[j]) *reinterpret_cast<f32x4*>(out_base + (size_t)mrow[j] * p.ld_out) = val[j];
    } else if (n_live) {        // the last, partial group of four channels of a row (N % 4 != 0)
#pragma unroll 1
      for (int j = 0; j < PH; ++j) {
        if (!live[j]) continue;
        float* dst = out_base + (size_t)mrow[j] * p.ld_out;
        for (int e = 0; e < 4 && n + e < p.N; ++e) {
          float x = val[j][e];
          if (p.mask) x = (p.mask[(size_t)mrow[j] * p.ld_mask + n + e] > 0.f) ? x : 0.f;
          dst[e] = x;
        }
      }
    }
  }
  STAMP(3);
  if (clk_block) { atomicAdd(&g_clk_ticks[0], clock64() - clk_c0); atomicAdd(&g_clk_ticks[1], wall_clock64() - clk_w0); }
}

}  // namespace

// address of g_zero16 on the current device (module data, resolved once per device)
static const float* zero_buffer() {
  static const float* cache[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (cache[dev] == nullptr) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_zero16)) != hipSuccess) return nullptr;
    cache[dev] = (const float*)ptr;
  }
  return cache[dev];
}

static int conv_validate(const clx_conv_desc* d, const char* who) {
  CLX_REQUIRE(d != nullptr, "%s: null descriptor", who);
  CLX_REQUIRE(d->nsrc == 1 || d->nsrc == 2, "%s: nsrc must be 1 or 2", who);
  CLX_REQUIRE(d->B > 0 && d->ID > 0 && d->IH > 0 && d->IW > 0, "%s: bad extent", who);
  const int K[3] = {d->KD, d->KH, d->KW};
  const int P[3] = {d->PD, d->PH, d->PW};
  const int I[3] = {d->ID, d->IH, d->IW};
  for (int i = 0; i < 3; ++i) {
    CLX_REQUIRE(K[i] >= 1 && K[i] <= 3, "%s: kernel extent must be 1, 2 or 3", who);
    CLX_REQUIRE(P[i] >= 0 && P[i] < K[i], "%s: padding must be < kernel extent", who);
    CLX_REQUIRE(I[i] + 2 * P[i] - K[i] + 1 > 0, "%s: input smaller than kernel", who);
  }
  for (int s = 0; s < d->nsrc; ++s) {
    const clx_src& S = d->src[s];
    CLX_REQUIRE(S.ptr != nullptr, "%s: null source %d", who, s);
    CLX_REQUIRE(S.C > 0 && S.C % 4 == 0 && S.ld % 4 == 0 && S.ld >= S.C,
                "%s: source channels/ld must be multiples of 4 (C=%d ld=%d)", who, S.C, S.ld);
    CLX_REQUIRE(((uintptr_t)S.ptr & 15) == 0, "%s: source pointer must be 16-byte aligned", who);
    CLX_REQUIRE(S.fz >= 1 && S.fy >= 1 && S.fx >= 1, "%s: upsample factors >= 1", who);
    CLX_REQUIRE(S.oz >= 0 && S.oy >= 0 && S.ox >= 0, "%s: negative crop", who);
    CLX_REQUIRE((d->ID - 1 + S.oz) / S.fz < S.D && (d->IH - 1 + S.oy) / S.fy < S.H &&
                    (d->IW - 1 + S.ox) / S.fx < S.W,
                "%s: source %d smaller than the logical input", who, s);
  }
  const long long M = (long long)d->B * (d->ID + 2 * d->PD - d->KD + 1) *
                      (d->IH + 2 * d->PH - d->KH + 1) * (d->IW + 2 * d->PW - d->KW + 1);
  CLX_REQUIRE(M < (1ll << 31), "%s: too many output pixels", who);
  return CLX_OK;
}

static void fill_params(const clx_conv_desc* d, ConvP& p) {
  p.nsrc = d->nsrc;
  for (int s = 0; s < 2; ++s) {
    const clx_src& S = d->src[s < d->nsrc ? s : 0];
    p.src[s] = SrcP{S.ptr, S.C, S.ld, S.D, S.H, S.W, S.oz, S.oy, S.ox, S.fz, S.fy, S.fx};
  }
  p.B = d->B; p.ID = d->ID; p.IH = d->IH; p.IW = d->IW;
  p.KD = d->KD; p.KH = d->KH; p.KW = d->KW;
  p.PD = d->PD; p.PH = d->PH; p.PW = d->PW;
  p.OD = d->ID + 2 * d->PD - d->KD + 1;
  p.OH = d->IH + 2 * d->PH - d->KH + 1;
  p.OW = d->IW + 2 * d->PW - d->KW + 1;
  p.N = d->N;
  p.M = d->B * p.OD * p.OH * p.OW;
  p.Ctot = d->src[0].C + (d->nsrc == 2 ? d->src[1].C : 0);
  p.Ktot = p.Ctot * d->KD * d->KH * d->KW;
  p.dOW = make_fastdiv(p.OW); p.dOH = make_fastdiv(p.OH); p.dOD = make_fastdiv(p.OD);
  p.wpack = d->wpack; p.bias = d->bias; p.mask = d->mask; p.out = d->out;
  p.gate_out = d->gate_out; p.ld_gate = d->ld_gate; p.mask_bits = d->mask_bits; p.ld_mask_bits = d->ld_mask_bits;
  p.relu = d->relu; p.ld_mask = d->ld_mask; p.ld_out = d->ld_out;
  p.accumulate = d->accumulate;
  p.bs_in = p.bs_w = p.bs_out = 0;
  p.flush_every = 0;
}

extern "C" int clx_conv_fwd(const clx_conv_desc* d, clx_stream stream) {
  int rc = conv_validate(d, "clx_conv_fwd");
  if (rc) return rc;
  CLX_REQUIRE(d->wpack && d->out, "clx_conv_fwd: null wpack/out");
  CLX_REQUIRE(d->N > 0 && d->ld_out >= d->N, "clx_conv_fwd: bad N/ld_out");
  CLX_REQUIRE(((uintptr_t)d->wpack & 15) == 0, "clx_conv_fwd: wpack must be 16-byte aligned");
  CLX_REQUIRE(d->mask == nullptr || (d->ld_mask >= d->N && d->ld_mask % 4 == 0 &&
                                     ((uintptr_t)d->mask & 15) == 0),
              "clx_conv_fwd: mask must be 16-byte aligned with ld_mask %% 4 == 0 and >= N");
  CLX_REQUIRE(d->ld_out % 4 == 0 && ((uintptr_t)d->out & 15) == 0,
              "clx_conv_fwd: out must be 16-byte aligned with ld_out %% 4 == 0");
  CLX_REQUIRE(d->gate_out == nullptr || (d->relu && d->ld_out % 32 == 0 && d->ld_gate * 32 >= d->ld_out),
              "clx_conv_fwd: gate_out needs relu, ld_out %% 32 == 0 and ld_gate >= ld_out / 32");
  CLX_REQUIRE(d->mask_bits == nullptr || (d->mask == nullptr && d->ld_mask_bits * 32 >= d->N),
              "clx_conv_fwd: mask_bits replaces mask and needs ld_mask_bits >= ceil(N / 32)");
  CLX_REQUIRE(d->algo == CLX_ALGO_DIRECT || d->algo == CLX_ALGO_WINOGRAD || d->algo == CLX_ALGO_WINOGRAD4 ||
                  d->algo == CLX_ALGO_WINOGRAD4_FUSED,
              "clx_conv_fwd: bad algo");
  if (d->algo == CLX_ALGO_WINOGRAD4_FUSED) return clx_wino_fused_fwd(d, (hipStream_t)stream);
  if (d->algo != CLX_ALGO_DIRECT) return clx_wino_fwd(d, (hipStream_t)stream);
  // pool_out / tile_list / adjoint are honoured by the Winograd paths only: a caller whose plan and dispatch disagree
  // would otherwise be left with a pooled buffer nobody wrote, or a dense recomputation, and no error
  CLX_REQUIRE(d->pool_out == nullptr && d->tile_list == nullptr && !d->adjoint,
              "clx_conv_fwd: pool_out / tile_list / adjoint need a Winograd algorithm (algo = %d)", d->algo);
  // the small-channel kernels know neither ReLU-gate form (float mask, gate bits in or out): a
  // layer that asks for one — e.g. a NON-first layer with 4 input channels whose output feeds a
  // bit-gated data gradient — stays on the implicit-GEMM kernel, which writes / applies them
  if (clx_smallc_applicable(d) && d->mask == nullptr && d->mask_bits == nullptr && d->gate_out == nullptr) {
    clx_smallc_fwd(d, (hipStream_t)stream);
    CLX_CHECK_LAUNCH("clx_conv_fwd(small-channel)");
    return CLX_OK;
  }
  rc = clx_igemm_launch(d, 1, 0, 0, 0, (hipStream_t)stream);
  if (rc) return rc;
  CLX_CHECK_LAUNCH("clx_conv_fwd");
  return CLX_OK;
}

// `batch` independent problems of identical geometry: batch b reads src[0].ptr + b*bs_in,
// wpack + b*bs_w and writes out + b*bs_out (mask is not batched).  Used by the Winograd path.
int clx_igemm_launch(const clx_conv_desc* d, int batch, long long bs_in, long long bs_w,
                     long long bs_out, hipStream_t st) {
  // opt-in precision: a 1x1 layer as split pass + product from planes (the Winograd paths call clx_sp_launch themselves,
  // with planes their transforms wrote)
  if (batch == 1 && d->aplanes != nullptr && clx_sp_applicable(d)) {
    const clx_src& S = d->src[0];
    const long long M = (long long)d->B * d->ID * d->IH * d->IW;
    if (!d->aplanes_valid) {
      const int rc = clx_sp_split(S.ptr, S.ld, M, S.C, d->aplanes, nullptr, 0, st);
      if (rc) return rc;
    }
    return clx_sp_launch(d->aplanes, d->wplanes, (int)M, d->N, S.C, M, 1, 0, 0, 0, d, st);
  }
  ConvP p;
  fill_params(d, p);
  p.zeros = zero_buffer();
  CLX_REQUIRE(p.zeros != nullptr, "clx_conv_fwd: cannot resolve the device zero buffer");
  p.bs_in = bs_in; p.bs_w = bs_w; p.bs_out = bs_out;
  static const int flush_env = getenv("CLX_IGEMM_FLUSH") ? atoi(getenv("CLX_IGEMM_FLUSH")) : 2;
  p.flush_every = flush_env > 0 ? flush_env : 0;
  // 128-wide N tiles unless padding N up to a multiple of 128 wastes > 20 % of the MFMAs
  const bool wide = d->N > 64 && (double)(cdiv(d->N, 128) * 128) / d->N <= 1.2;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (clx_prof_enabled())
    clx_prof_events(wide ? CLX_PROF_IGEMM_WIDE : CLX_PROF_IGEMM_NARROW, 2.0 * p.M * p.N * p.Ktot * batch, &e0, &e1);
  // no zero rows and 32-bit offsets?  (valid convolutions over channel counts that are multiples of the K chunk)
  static const bool fast_env = getenv("CLX_IGEMM_FAST") == nullptr || atoi(getenv("CLX_IGEMM_FAST")) != 0;
  int fast = fast_env && (long long)p.N * p.Ktot < (1ll << 30) ? 1 : 0;
  const bool padded = d->PD != 0 || d->PH != 0 || d->PW != 0;
  for (int s = 0; s < d->nsrc && fast; ++s) {
    const clx_src& S = d->src[s];
    // ONE batch's tensor: the batch stride goes into the 64-bit base pointer (sptr = S.ptr + blockIdx.y * bs_in), only the
    // offsets inside a batch are 32-bit.  (The whole batched tensor counted here until round 4: the 36 x tiles x C operand
    // of a Winograd layer over eight 526^2 inference copies is 1.3 G floats, and those launches took the general path.)
    const long long floats = (long long)d->B * S.D * S.H * S.W * S.ld;
    // byte offsets in 32 bits; with padding the rows come through buffer loads and 0x80000000 must lie outside the tensor
    if (S.C % BK != 0 || floats >= (padded ? (1ll << 29) : (1ll << 30))) fast = 0;
  }
  if (fast && padded) fast = 2;
  static const bool dbg = getenv("CLX_IGEMM_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "igemm fast=%d wide=%d M=%d N=%d K=%d batch=%d pad=%d%d%d k=%d%d%d nsrc=%d C0=%d\n", (int)fast, (int)wide, p.M, p.N, p.Ktot, batch, d->PD, d->PH, d->PW, d->KD, d->KH, d->KW, d->nsrc, d->src[0].C);
  if (wide) {
    p.nbm = cdiv(p.M, 128); p.nbn = cdiv(p.N, 128);
    if (fast == 1) CLX_LAUNCH_TIMED((conv_igemm_kernel<128, 128, 2, 2, 1>), dim3(p.nbm * p.nbn, batch), dim3(256), st, e0, e1, p);
    else if (fast == 2) CLX_LAUNCH_TIMED((conv_igemm_kernel<128, 128, 2, 2, 2>), dim3(p.nbm * p.nbn, batch), dim3(256), st, e0, e1, p);
    else CLX_LAUNCH_TIMED((conv_igemm_kernel<128, 128, 2, 2>), dim3(p.nbm * p.nbn, batch), dim3(256), st, e0, e1, p);
  } else {
    // (256-row tiles for narrow N — a wave owning 64 x 64 like the wide kernel's, two 80-KB blocks per CU — measured in round
    //  4: 106 against 111 TFLOP/s over the 3-D step's fifteen launches, 91 against 96 on the 2-D step's three: three blocks
    //  per CU matter more to these short-K launches than the fragment reads saved)
    p.nbm = cdiv(p.M, 128); p.nbn = cdiv(p.N, 64);
    if (fast == 1) CLX_LAUNCH_TIMED((conv_igemm_kernel<128, 64, 4, 1, 1>), dim3(p.nbm * p.nbn, batch), dim3(256), st, e0, e1, p);
    else if (fast == 2) CLX_LAUNCH_TIMED((conv_igemm_kernel<128, 64, 4, 1, 2>), dim3(p.nbm * p.nbn, batch), dim3(256), st, e0, e1, p);
    else CLX_LAUNCH_TIMED((conv_igemm_kernel<128, 64, 4, 1>), dim3(p.nbm * p.nbn, batch), dim3(256), st, e0, e1, p);
  }
  return CLX_OK;
}
