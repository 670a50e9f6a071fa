// Host-side decoder for the chunks zarr-python writes by default (Blosc container, LZ4 codec):
// the raw data of the reference's example containers (docs/examples/2d/01-data.py:35-50 ->
// zarr.open(...)[...] = array with the default compressor Blosc(cname="lz4", clevel=5,
// shuffle=SHUFFLE)), read by cellulus/datasets/zarr_dataset.py:104-121 through gunpowder.
// numcodecs / c-blosc are C libraries of the reference's environment and absent here; this file
// is the LZ4 block decoder (format: lz4_Block_format.md) and the byte un-shuffle, the container
// header / block table is parsed by cellulus_amd/utils/zarr_io.py.  Plain C++ — no GPU code.
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "clx_common.h"

// LZ4 block: sequences of [token][literal length ext][literals][offset LE16][match length ext].
// Returns the number of bytes written, or a negative value on malformed / overflowing input.
extern "C" long long clx_lz4_decompress(const unsigned char* src, long long src_bytes, unsigned char* dst,
                                        long long dst_capacity) {
  if (!src || !dst || src_bytes < 0 || dst_capacity < 0) return -1;
  const unsigned char* ip = src;
  const unsigned char* const iend = src + src_bytes;
  unsigned char* op = dst;
  unsigned char* const oend = dst + dst_capacity;
  while (ip < iend) {
    const unsigned token = *ip++;
    size_t lit = token >> 4;
    if (lit == 15) {
      unsigned b;
      do {
        if (ip >= iend) return -2;
        b = *ip++;
        lit += b;
      } while (b == 255);
    }
    if ((size_t)(iend - ip) < lit || (size_t)(oend - op) < lit) return -3;
    memcpy(op, ip, lit);
    ip += lit;
    op += lit;
    if (ip >= iend) break;                 // the last sequence holds literals only
    if (iend - ip < 2) return -4;
    const size_t offset = (size_t)ip[0] | ((size_t)ip[1] << 8);
    ip += 2;
    if (offset == 0 || offset > (size_t)(op - dst)) return -5;
    size_t mlen = token & 15;
    if (mlen == 15) {
      unsigned b;
      do {
        if (ip >= iend) return -6;
        b = *ip++;
        mlen += b;
      } while (b == 255);
    }
    mlen += 4;
    if ((size_t)(oend - op) < mlen) return -7;
    const unsigned char* match = op - offset;
    if (offset >= mlen) {
      memcpy(op, match, mlen);
      op += mlen;
    } else {                               // overlapping copy replicates the pattern byte by byte
      for (size_t k = 0; k < mlen; ++k) op[k] = match[k];
      op += mlen;
    }
  }
  return (long long)(op - dst);
}

// BloscLZ stream (c-blosc 1.x blosclz.c, a FastLZ descendant; codec 0 of a Blosc chunk): control bytes < 32 start
// a run of ctrl + 1 literals; others are matches — length (ctrl >> 5) + 2, 255-extended when the field is 7;
// distance ((ctrl & 31) << 8) + next byte + 1, or, when both fields are saturated, a 16-bit big-endian distance
// beyond 8191.  The first control byte is masked to a literal run.  Returns the bytes written, or a negative
// value on malformed / overflowing input.
extern "C" long long clx_blosclz_decompress(const unsigned char* src, long long src_bytes, unsigned char* dst,
                                            long long dst_capacity) {
  if (!src || !dst || src_bytes < 0 || dst_capacity < 0) return -1;
  if (src_bytes == 0) return 0;
  const unsigned char* ip = src;
  const unsigned char* const iend = src + src_bytes;
  unsigned char* op = dst;
  unsigned char* const oend = dst + dst_capacity;
  unsigned ctrl = (*ip++) & 31u;
  for (;;) {
    if (ctrl >= 32) {
      long long len = (long long)(ctrl >> 5) - 1;
      long long ofs = (long long)(ctrl & 31u) << 8;
      unsigned code;
      if (len == 7 - 1) {
        do {
          if (ip + 1 >= iend) return -2;
          code = *ip++;
          len += code;
        } while (code == 255);
      } else if (ip + 1 >= iend) {
        return -3;
      }
      code = *ip++;
      len += 3;
      long long dist = ofs + code;
      if (code == 255 && ofs == (31ll << 8)) {
        if (ip + 1 >= iend) return -4;
        dist = ((long long)ip[0] << 8) + ip[1] + 8191;
        ip += 2;
      }
      dist += 1;
      if (len > oend - op) return -5;
      if (dist > op - dst) return -6;
      const bool last = ip >= iend;
      if (!last) ctrl = *ip++;
      const unsigned char* match = op - dist;
      for (long long k = 0; k < len; ++k) op[k] = match[k];     // byte by byte: overlapping matches replicate
      op += len;
      if (last) break;
    } else {
      const long long run = (long long)ctrl + 1;
      if (run > oend - op || run > iend - ip) return -7;
      memcpy(op, ip, (size_t)run);
      op += run;
      ip += run;
      if (ip >= iend) break;
      ctrl = *ip++;
    }
  }
  return (long long)(op - dst);
}

// Blosc byte shuffle, inverse: src holds `typesize` planes of n / typesize bytes (plane j = byte j
// of every element), then the n % typesize left-over bytes verbatim.
extern "C" int clx_unshuffle_bytes(const unsigned char* src, unsigned char* dst, long long n, int typesize) {
  if (!src || !dst || n < 0 || typesize < 1) return CLX_ERR_ARG;
  const long long ne = n / typesize;
  for (int j = 0; j < typesize; ++j) {
    const unsigned char* plane = src + (long long)j * ne;
    for (long long i = 0; i < ne; ++i) dst[i * typesize + j] = plane[i];
  }
  const long long done = ne * typesize;
  memcpy(dst + done, src + done, (size_t)(n - done));
  return CLX_OK;
}
