// The noisy copies of the infer-mode forward (cellulus/models/unet.py:73-100) differ from the image in p_salt_pepper of
// their pixels (1 % by default): behind the first k x k convolution a copy's activations differ from the clean image's
// in the dilated set of those pixels (8.6 % for 3 x 3), and the 1 x 1 layers that follow keep that set.  A row of a
// 1 x 1 layer's output depends on the same row of its input and on nothing else — the matrix cores accumulate an output
// element over K in the same order wherever its row sits in a tile — so computing the 1 x 1 layers once on the clean
// image and again on the CHANGED rows of each copy gives the bits of the dense computation (tests/test_gpu_unet.py
// compares the two paths with torch.equal).  These are the row movers around that: the list of changed rows, gather,
// scatter, and the broadcast of the clean rows.
#include "clx_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Changed rows in two passes over bits (one thread per output pixel with its whole window — 18 loads and an
// index decode each — took 1.4 ms for 32 copies of a 528^2 tile; this form 0.1 ms):
//   diff_bits_kernel    one wavefront per 64 input pixels of a row: bit = the copy differs from the clean image there
//                       (any channel); (T, ID, IH, ceil(IW / 64)) 64-bit words
//   dilate_rows_kernel  one thread per 64 OUTPUT pixels of a row: OR of the window's rows, shifted over the window's
//                       width; the set bits are appended to the chunk's list — one atomic per wavefront and chunk — as
//                       (copy inside the chunk) * npix_out + output pixel, in no particular order: every consumer
//                       addresses by the row's value.
__global__ __launch_bounds__(256) void diff_bits_kernel(const float* __restrict__ clean, const float* __restrict__ noisy,
                                                        int C, long long npix_in, int IW, int nw, long long nwords_copy,
                                                        long long total_words, unsigned long long* __restrict__ bits) {
  const int lane = threadIdx.x & 63;
  for (long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < total_words;
       w += ((long long)gridDim.x * blockDim.x) >> 6) {
    const long long t = w / nwords_copy, r = w - t * nwords_copy;     // r = (z * IH + y) * nw + word
    const long long line = r / nw;
    const int x = (int)(r - line * nw) * 64 + lane;
    bool diff = false;
    if (x < IW) {
      const long long q = line * IW + x;
      for (int c = 0; c < C; ++c) diff |= clean[c * npix_in + q] != noisy[(t * C + c) * npix_in + q];
    }
    const unsigned long long word = __ballot(diff);
    if (lane == 0) bits[w] = word;
  }
}

__global__ __launch_bounds__(256) void dilate_rows_kernel(const unsigned long long* __restrict__ bits, int T, int IH,
                                                          int nw, long long nwords_copy, int KD, int KH, int KW, int OD,
                                                          int OH, int OW, int onw, int chunk, int* __restrict__ rows,
                                                          int* __restrict__ counts, long long cap,
                                                          unsigned long long* __restrict__ row_bits) {
  const long long npix_out = (long long)OD * OH * OW;
  const long long per_copy = (long long)OD * OH * onw, total = (long long)T * per_copy;
  const int lane = threadIdx.x & 63;
  const long long rounded = (total + 63) / 64 * 64;              // every lane reaches the ballots
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += (long long)gridDim.x * blockDim.x) {
    unsigned long long acc = 0ull;
    int t = 0, wd = 0;
    long long line = 0;
    if (i < total) {
      t = (int)(i / per_copy);
      const long long r = i - (long long)t * per_copy;
      line = r / onw;                                             // oz * OH + oy
      wd = (int)(r - line * onw);
      const int oy = (int)(line % OH), oz = (int)(line / OH);
      for (int dz = 0; dz < KD; ++dz)
        for (int dy = 0; dy < KH; ++dy) {
          const unsigned long long* row = bits + (long long)t * nwords_copy + ((long long)(oz + dz) * IH + (oy + dy)) * nw;
          const unsigned long long lo = row[wd], hi = wd + 1 < nw ? row[wd + 1] : 0ull;
          acc |= lo;
          for (int dx = 1; dx < KW; ++dx) acc |= (lo >> dx) | (hi << (64 - dx));
        }
      const int left = OW - wd * 64;                              // output pixels of this word inside the row
      if (left < 64) acc &= (1ull << left) - 1ull;
      row_bits[i] = acc;                                          // [copy][oz * OH + oy][word]: clx_changed_tiles reads these
    }
    const int n = __popcll(acc);
    int pending = n > 0 ? t / chunk : -1;
    long long base = 0;
    while (true) {                                                // (a wavefront's words may straddle two chunks)
      const unsigned long long want = __ballot(pending >= 0);
      if (want == 0ull) break;
      const int leader = __builtin_ctzll(want);
      const int cur = __shfl(pending, leader, 64);
      const bool in = pending == cur;
      int incl = in ? n : 0;                                      // inclusive prefix of the counts of this chunk's lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      const int wave_total = __shfl(incl, 63, 64);
      int start = 0;
      if (lane == leader) start = atomicAdd(&counts[cur], wave_total);
      start = __shfl(start, leader, 64);
      if (in) {
        base = (long long)start + incl - n;
        pending = -1;
      }
    }
    if (n > 0) {
      const int ck = t / chunk;
      const long long first = (long long)(t - ck * chunk) * npix_out + line * OW + (long long)wd * 64;
      long long pos = base;
      for (unsigned long long m = acc; m != 0ull; m &= m - 1ull, ++pos)
        if (pos < cap) rows[(long long)ck * cap + pos] = (int)(first + __builtin_ctzll(m));
    }
  }
}

// Output tiles (tile x tile, 2-D) of a (WH, WW) valid convolution over the rows whose bits dilate_rows_kernel left: a
// tile is CHANGED if any row of its (tile + WH - 1) x (tile + WW - 1) window is.  One thread per (copy, ty, tx); the
// chunk's list takes ((copy inside the chunk) * th + ty) * tw + tx, in no particular order.
__global__ __launch_bounds__(256) void changed_tiles_kernel(const unsigned long long* __restrict__ row_bits, int T, int OH,
                                                            int OW, int onw, int WH, int WW, int tile, int th, int tw,
                                                            int chunk, int* __restrict__ tiles, int* __restrict__ counts,
                                                            long long cap) {
  const long long per_copy = (long long)th * tw, total = (long long)T * per_copy;
  const int lane = threadIdx.x & 63;
  const long long rounded = (total + 63) / 64 * 64;              // every lane reaches the ballots
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += (long long)gridDim.x * blockDim.x) {
    bool changed = false;
    int t = 0;
    long long r = 0;
    if (i < total) {
      t = (int)(i / per_copy);
      r = i - (long long)t * per_copy;
      const int ty = (int)(r / tw), tx = (int)(r - (long long)ty * tw);
      const int y0 = ty * tile, y1 = min(y0 + tile + WH - 1, OH);
      const int x0 = tx * tile, nbits = min(x0 + tile + WW - 1, OW) - x0;        // (<= 64: checked by the launcher)
      const int wd = x0 >> 6, sh = x0 & 63;
      const unsigned long long mask = nbits >= 64 ? ~0ull : (1ull << nbits) - 1ull;
      unsigned long long any = 0ull;
      for (int y = y0; y < y1; ++y) {
        const unsigned long long* row = row_bits + ((long long)t * OH + y) * onw;
        unsigned long long v = row[wd] >> sh;
        if (sh != 0 && wd + 1 < onw) v |= row[wd + 1] << (64 - sh);
        any |= v & mask;
      }
      changed = any != 0ull;
    }
    int pending = changed ? t / chunk : -1;
    while (true) {                                                // (a wavefront's tiles may straddle two chunks)
      const unsigned long long want = __ballot(pending >= 0);
      if (want == 0ull) break;
      const int leader = __builtin_ctzll(want);
      const int cur = __shfl(pending, leader, 64);
      const unsigned long long mine = __ballot(pending == cur);
      int base = 0;
      if (lane == leader) base = atomicAdd(&counts[cur], __popcll(mine));
      base = __shfl(base, leader, 64);
      if (pending == cur) {
        const long long pos = base + __popcll(mine & ((1ull << lane) - 1ull));
        if (pos < cap) tiles[(long long)cur * cap + pos] = (int)((long long)(t - cur * chunk) * per_copy + r);
        pending = -1;
      }
    }
  }
}

// dst[r][0 .. width) = src[rows[r]][0 .. width)   (width % 4 == 0; one 16-byte group per thread)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int ld_src,
                                                          const int* __restrict__ rows, long long n, int w4,
                                                          float* __restrict__ dst, int ld_dst) {
  const long long total = n * w4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / w4;
    const int c = (int)(i - r * w4) * 4;
    *reinterpret_cast<f32x4*>(dst + r * ld_dst + c) = *reinterpret_cast<const f32x4*>(src + (long long)rows[r] * ld_src + c);
  }
}

// dst[rows[r]][0 .. width) = src[r][0 .. width)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, int ld_src,
                                                           const int* __restrict__ rows, long long n, int w4,
                                                           float* __restrict__ dst, int ld_dst) {
  const long long total = n * w4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / w4;
    const int c = (int)(i - r * w4) * 4;
    *reinterpret_cast<f32x4*>(dst + (long long)rows[r] * ld_dst + c) = *reinterpret_cast<const f32x4*>(src + r * ld_src + c);
  }
}

// dst[k][i] = src[i] for k < copies (n4 groups of 16 bytes): one read, `copies` writes
__global__ __launch_bounds__(256) void broadcast_rows_kernel(const f32x4* __restrict__ src, long long n4,
                                                             f32x4* __restrict__ dst, int copies) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = src[i];
    for (int k = 0; k < copies; ++k) dst[(long long)k * n4 + i] = v;
  }
}

inline int rows_grid(long long total) {
  long long g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

// workspace: [one bit per input pixel and copy][one bit per output row and copy, at most as many]
static size_t input_bit_words(int T, int ID, int IH, int IW) { return (size_t)T * ID * IH * ((IW + 63) / 64); }
extern "C" size_t clx_changed_rows_workspace(int T, int ID, int IH, int IW) {
  return 2 * input_bit_words(T, ID, IH, IW) * sizeof(unsigned long long);
}

extern "C" int clx_changed_rows(const float* clean, const float* noisy, int T, int C, int ID, int IH, int IW, int KD,
                                int KH, int KW, int chunk, int* rows, int* counts, long long cap, void* workspace,
                                clx_stream stream) {
  CLX_REQUIRE(clean && noisy && rows && counts && workspace, "clx_changed_rows: null pointer");
  CLX_REQUIRE(T > 0 && C > 0 && ID > 0 && IH > 0 && IW > 0, "clx_changed_rows: bad extents");
  CLX_REQUIRE(KD >= 1 && KH >= 1 && KW >= 1 && KD <= ID && KH <= IH && KW <= IW && KW < 64, "clx_changed_rows: bad window");
  CLX_REQUIRE(chunk >= 1 && cap >= 1, "clx_changed_rows: bad chunk / capacity");
  CLX_REQUIRE(((uintptr_t)workspace & 7) == 0, "clx_changed_rows: workspace must be 8-byte aligned");
  const int OD = ID - KD + 1, OH = IH - KH + 1, OW = IW - KW + 1;
  const long long npix_out = (long long)OD * OH * OW;
  CLX_REQUIRE((long long)chunk * npix_out < (1ll << 31), "clx_changed_rows: a chunk's rows must fit 31 bits");
  hipStream_t st = (hipStream_t)stream;
  const int nchunks = (T + chunk - 1) / chunk;
  CLX_REQUIRE(hipMemsetAsync(counts, 0, sizeof(int) * nchunks, st) == hipSuccess, "clx_changed_rows: hipMemsetAsync failed");
  unsigned long long* bits = (unsigned long long*)workspace;
  const int nw = (IW + 63) / 64, onw = (OW + 63) / 64;
  const long long nwords_copy = (long long)ID * IH * nw, total_words = (long long)T * nwords_copy;
  diff_bits_kernel<<<rows_grid(total_words * 64), 256, 0, st>>>(clean, noisy, C, (long long)ID * IH * IW, IW, nw, nwords_copy,
                                                                total_words, bits);
  dilate_rows_kernel<<<rows_grid((long long)T * OD * OH * onw), 256, 0, st>>>(bits, T, IH, nw, nwords_copy, KD, KH, KW, OD, OH,
                                                                            OW, onw, chunk, rows, counts, cap,
                                                                            bits + input_bit_words(T, ID, IH, IW));
  CLX_CHECK_LAUNCH("clx_changed_rows");
  return CLX_OK;
}

extern "C" int clx_changed_tiles(const void* workspace, int T, int ID, int IH, int IW, int KD, int KH, int KW, int WH,
                                 int WW, int tile, int chunk, int* tiles, int* counts, long long cap, clx_stream stream) {
  CLX_REQUIRE(workspace && tiles && counts, "clx_changed_tiles: null pointer");
  CLX_REQUIRE(T > 0 && ID > 0 && IH > 0 && IW > 0 && KD >= 1 && KH >= 1 && KW >= 1, "clx_changed_tiles: bad extents");
  const int OD = ID - KD + 1, OH = IH - KH + 1, OW = IW - KW + 1;
  CLX_REQUIRE(OD == 1, "clx_changed_tiles: 2-D layers only (one output plane)");
  CLX_REQUIRE(WH >= 1 && WW >= 1 && WH <= OH && WW <= OW && tile >= 1 && tile + WW - 1 <= 64,
              "clx_changed_tiles: bad window / tile");
  CLX_REQUIRE(chunk >= 1 && cap >= 1, "clx_changed_tiles: bad chunk / capacity");
  const int th = (OH - WH + 1 + tile - 1) / tile, tw = (OW - WW + 1 + tile - 1) / tile;
  CLX_REQUIRE((long long)chunk * th * tw < (1ll << 31), "clx_changed_tiles: a chunk's tiles must fit 31 bits");
  hipStream_t st = (hipStream_t)stream;
  const int nchunks = (T + chunk - 1) / chunk;
  CLX_REQUIRE(hipMemsetAsync(counts, 0, sizeof(int) * nchunks, st) == hipSuccess, "clx_changed_tiles: hipMemsetAsync failed");
  const unsigned long long* row_bits = (const unsigned long long*)workspace + input_bit_words(T, ID, IH, IW);
  changed_tiles_kernel<<<rows_grid((long long)T * th * tw), 256, 0, st>>>(row_bits, T, OH, OW, (OW + 63) / 64, WH, WW, tile, th,
                                                                          tw, chunk, tiles, counts, cap);
  CLX_CHECK_LAUNCH("clx_changed_tiles");
  return CLX_OK;
}

extern "C" int clx_gather_rows(const float* src, int ld_src, const int* rows, long long n, int width, float* dst,
                               int ld_dst, clx_stream stream) {
  CLX_REQUIRE(n >= 0, "clx_gather_rows: bad count");
  if (n == 0) return CLX_OK;
  CLX_REQUIRE(src && rows && dst, "clx_gather_rows: null pointer");
  CLX_REQUIRE(width > 0 && width % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && width <= ld_src && width <= ld_dst,
              "clx_gather_rows: widths must be multiples of 4 floats");
  CLX_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "clx_gather_rows: 16-byte alignment");
  gather_rows_kernel<<<rows_grid(n * (width / 4)), 256, 0, (hipStream_t)stream>>>(src, ld_src, rows, n, width / 4, dst, ld_dst);
  CLX_CHECK_LAUNCH("clx_gather_rows");
  return CLX_OK;
}

extern "C" int clx_scatter_rows(const float* src, int ld_src, const int* rows, long long n, int width, float* dst,
                                int ld_dst, clx_stream stream) {
  CLX_REQUIRE(n >= 0, "clx_scatter_rows: bad count");
  if (n == 0) return CLX_OK;
  CLX_REQUIRE(src && rows && dst, "clx_scatter_rows: null pointer");
  CLX_REQUIRE(width > 0 && width % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && width <= ld_src && width <= ld_dst,
              "clx_scatter_rows: widths must be multiples of 4 floats");
  CLX_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "clx_scatter_rows: 16-byte alignment");
  scatter_rows_kernel<<<rows_grid(n * (width / 4)), 256, 0, (hipStream_t)stream>>>(src, ld_src, rows, n, width / 4, dst, ld_dst);
  CLX_CHECK_LAUNCH("clx_scatter_rows");
  return CLX_OK;
}

extern "C" int clx_broadcast_rows(const float* src, long long nfloats, float* dst, int copies, clx_stream stream) {
  CLX_REQUIRE(src && dst, "clx_broadcast_rows: null pointer");
  CLX_REQUIRE(nfloats > 0 && nfloats % 4 == 0 && copies >= 1, "clx_broadcast_rows: bad extents");
  CLX_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "clx_broadcast_rows: 16-byte alignment");
  broadcast_rows_kernel<<<rows_grid(nfloats / 4), 256, 0, (hipStream_t)stream>>>(
      reinterpret_cast<const f32x4*>(src), nfloats / 4, reinterpret_cast<f32x4*>(dst), copies);
  CLX_CHECK_LAUNCH("clx_broadcast_rows");
  return CLX_OK;
}
