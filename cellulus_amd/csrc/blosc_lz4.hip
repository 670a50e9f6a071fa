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

// ---------------------------------------------------------------------------------------------------------
// Writer side: one Blosc chunk as c-blosc 1.x lays it out (read back by numcodecs.Blosc / zarr's default
// compressor): header, block offsets, per block `typesize` byte planes (byte shuffle) compressed one by one as
// LZ4 blocks — or one stream when the planes would be shorter than 128 bytes or the block is the shorter last
// one — each preceded by its int32 length (a stream that does not shrink is stored verbatim with length =
// plain size); a chunk that does not shrink at all is stored verbatim (flag 0x02).
namespace {

inline unsigned rd32(const unsigned char* p) { unsigned v; memcpy(&v, p, 4); return v; }

// greedy LZ4 block compressor (one 4-byte hash probe per position); returns the compressed size, or 0 when
// the output would not fit `cap`
long long lz4_compress_block(const unsigned char* src, long long n, unsigned char* dst, long long cap) {
  static thread_local int table[1 << 14];
  for (int i = 0; i < (1 << 14); ++i) table[i] = -1;
  const unsigned char* ip = src;
  const unsigned char* anchor = src;
  const unsigned char* const iend = src + n;
  const unsigned char* const mflimit = iend - 12;      // a match must not start in the last 12 bytes
  const unsigned char* const matchlimit = iend - 5;    // ... nor cover the last 5
  unsigned char* op = dst;
  unsigned char* const oend = dst + cap;
  auto emit = [&](const unsigned char* lit_end, long long offset, long long mlen) -> bool {
    const long long lit = lit_end - anchor;
    if (op + 1 + lit / 255 + 1 + lit + 2 + mlen / 255 + 1 > oend) return false;
    unsigned char* token = op++;
    if (lit >= 15) {
      *token = 15u << 4;
      long long r = lit - 15;
      for (; r >= 255; r -= 255) *op++ = 255;
      *op++ = (unsigned char)r;
    } else {
      *token = (unsigned char)(lit << 4);
    }
    memcpy(op, anchor, (size_t)lit);
    op += lit;
    if (mlen == 0) return true;                        // last sequence: literals only
    *op++ = (unsigned char)(offset & 255);
    *op++ = (unsigned char)(offset >> 8);
    long long m = mlen - 4;
    if (m >= 15) {
      *token |= 15;
      m -= 15;
      for (; m >= 255; m -= 255) *op++ = 255;
      *op++ = (unsigned char)m;
    } else {
      *token |= (unsigned char)m;
    }
    return true;
  };
  if (n >= 13) {
    while (ip < mflimit) {
      const unsigned h = (rd32(ip) * 2654435761u) >> 18;
      const int ref = table[h];
      table[h] = (int)(ip - src);
      if (ref >= 0 && (ip - src) - ref <= 65535 && rd32(src + ref) == rd32(ip)) {
        const unsigned char* m = src + ref;
        long long mlen = 4;
        while (ip + mlen < matchlimit && m[mlen] == ip[mlen]) ++mlen;
        if (!emit(ip, (ip - src) - ref, mlen)) return 0;
        ip += mlen;
        anchor = ip;
      } else {
        ++ip;
      }
    }
  }
  if (!emit(iend, 0, 0)) return 0;
  return (long long)(op - dst);
}

}  // namespace

// Upper bound of clx_blosc_compress_lz4's output for `nbytes` input bytes.
extern "C" long long clx_blosc_compress_bound(long long nbytes) { return nbytes + 16; }

// src: nbytes of elements of `typesize` bytes; dst: at least clx_blosc_compress_bound(nbytes) bytes.  Returns the
// chunk size, or < 0.  `shuffle` = 1: byte shuffle (zarr's default), 0: none.
extern "C" long long clx_blosc_compress_lz4(const unsigned char* src, long long nbytes, int typesize, int shuffle,
                                            unsigned char* dst, long long dst_capacity) {
  if (!src || !dst || nbytes < 0 || nbytes > 0x7fffffffll - 16 || typesize < 1 || typesize > 255) return -1;
  if (dst_capacity < nbytes + 16) return -2;
  const bool shuf = shuffle != 0 && typesize > 1;
  long long blocksize = 256 * 1024;
  blocksize -= blocksize % typesize;
  if (blocksize <= 0 || blocksize > nbytes) blocksize = nbytes > 0 ? nbytes : 1;
  const long long nblocks = nbytes > 0 ? (nbytes + blocksize - 1) / blocksize : 0;
  const long long leftover = nbytes > 0 ? nbytes % blocksize : 0;
  auto put32 = [](unsigned char* p, long long v) { const int x = (int)v; memcpy(p, &x, 4); };
  dst[0] = 2; dst[1] = 1;                               // format version, LZ4 format version
  dst[2] = (unsigned char)((shuf ? 0x01 : 0x00) | (1 << 5));
  dst[3] = (unsigned char)typesize;
  put32(dst + 4, nbytes); put32(dst + 8, blocksize);
  long long pos = 16 + 4 * nblocks;
  bool fits = pos < nbytes + 16;
  unsigned char* planes = (unsigned char*)malloc((size_t)blocksize);
  if (!planes) return -3;
  for (long long b = 0; b < nblocks && fits; ++b) {
    const bool last_short = b == nblocks - 1 && leftover > 0;
    const long long bsize = last_short ? leftover : blocksize;
    const unsigned char* blk = src + b * blocksize;
    const unsigned char* data = blk;
    if (shuf) {
      const long long ne = bsize / typesize;
      for (int j = 0; j < typesize; ++j)
        for (long long i = 0; i < ne; ++i) planes[(long long)j * ne + i] = blk[i * typesize + j];
      memcpy(planes + ne * typesize, blk + ne * typesize, (size_t)(bsize - ne * typesize));
      data = planes;
    }
    put32(dst + 16 + 4 * b, pos);
    const int nsplits = (typesize <= 16 && blocksize / typesize >= 128 && !last_short) ? typesize : 1;
    const long long neblock = bsize / nsplits;
    for (int s = 0; s < nsplits && fits; ++s) {
      // budget: the chunk must stay below nbytes + 16 in total, else the whole chunk is stored verbatim
      const long long room = nbytes + 16 - (pos + 4);
      if (room <= 0) { fits = false; break; }
      long long c = lz4_compress_block(data + s * neblock, neblock, dst + pos + 4, neblock - 1 < room ? neblock - 1 : room);
      if (c <= 0) {                                      // does not shrink: verbatim stream, if it still fits
        if (room < neblock) { fits = false; break; }
        memcpy(dst + pos + 4, data + s * neblock, (size_t)neblock);
        c = neblock;
      }
      put32(dst + pos, c);
      pos += 4 + c;
    }
  }
  free(planes);
  if (!fits || pos >= nbytes + 16) {                     // stored chunk
    dst[2] |= 0x02;
    memcpy(dst + 16, src, (size_t)nbytes);
    pos = nbytes + 16;
  }
  put32(dst + 12, pos);
  return pos;
}
