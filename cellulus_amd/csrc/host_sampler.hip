// HOST code: the offset draws of the reference's pair sampler with numpy's own random stream.
//
// cellulus/datasets/zarr_dataset.py:185-198 (sample_offsets_within_radius) draws, per crop, ND arrays of
// ND * number_offsets integers with np.random.randint(-radius, radius + 1, size), keeps the positions whose vector
// lies strictly inside the radius and is not zero, and takes the first number_offsets of them.  For the benchmark
// crop that is 787 400 bounded integers: 13 of the 20 ms one loader process spends per crop are numpy's generator,
// 7 ms the filtering temporaries.  This file restates numpy's LEGACY stream — RandomState's MT19937, 32-bit outputs,
// masked rejection (numpy/random/src/mt19937/mt19937.c, src/distributions/distributions.c:
// random_bounded_uint64_fill with use_masked = 1 -> buffered_bounded_masked_uint32; numpy 2.2.x) — so that the SAME
// offsets come out and the global generator is left in the SAME state, in one pass without temporaries of the
// draws' size in int64.  tests/test_cpu_host.py pins it against np.random itself (values and state).
#include <stdint.h>
#include <vector>

#include "clx_common.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;

struct MT {
  uint32_t* key;
  int pos;
  void gen() {
    int kk = 0;
    uint32_t y;
    for (; kk < MT_N - MT_M; ++kk) {
      y = (key[kk] & 0x80000000u) | (key[kk + 1] & 0x7fffffffu);
      key[kk] = key[kk + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    }
    for (; kk < MT_N - 1; ++kk) {
      y = (key[kk] & 0x80000000u) | (key[kk + 1] & 0x7fffffffu);
      key[kk] = key[kk + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    }
    y = (key[MT_N - 1] & 0x80000000u) | (key[0] & 0x7fffffffu);
    key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    pos = 0;
  }
  inline uint32_t next() {
    if (pos == MT_N) gen();
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
};

}  // namespace

extern "C" int clx_sample_offsets_mt19937(unsigned int* key, int* pos, int radius, int ND, long long number_offsets,
                                          long long* offsets, int* rounds) {
  CLX_REQUIRE(key && pos && offsets && (ND == 2 || ND == 3) && radius >= 1 && radius < 16384 && number_offsets >= 0,
              "clx_sample_offsets_mt19937: bad arguments");
  CLX_REQUIRE(*pos >= 0 && *pos <= MT_N, "clx_sample_offsets_mt19937: bad generator position");
  MT mt{key, *pos};
  const uint32_t rng = 2u * (uint32_t)radius;        // randint(-r, r + 1): high - 1 - low
  uint32_t mask = rng;                                // smallest 2^k - 1 >= rng
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  const long long L = (long long)ND * number_offsets;
  const int r2 = radius * radius;
  std::vector<int16_t> draws;
  try {        // (an allocation failure must not cross the C boundary: ctypes would std::terminate the process)
    draws.resize((size_t)ND * (size_t)L);
  } catch (const std::exception& e) {
    clx_set_error("clx_sample_offsets_mt19937: %s", e.what());
    return CLX_ERR_WORKSPACE;
  }
  int trips = 0;
  for (;;) {
    ++trips;
    // the reference's order: one randint call per dimension, ND * number_offsets values each.  Whole blocks of the
    // generator's 624 outputs are tempered, masked and compacted WITHOUT branches (a third of the values is rejected: a
    // branch per value mispredicts every third time); the block in which a dimension's last value falls goes one by one,
    // so the generator stops at exactly the output numpy would have stopped at.
    for (int d = 0; d < ND; ++d) {
      int16_t* out = draws.data() + (size_t)d * L;
      long long k = 0;
      while (k < L) {
        if (mt.pos == MT_N) mt.gen();
        const int avail = MT_N - mt.pos;
        if (L - k >= avail) {                          // even if every value is accepted the block does not overshoot
          const uint32_t* src = key + mt.pos;
          long long kk = k;
          for (int i = 0; i < avail; ++i) {
            uint32_t y = src[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            const uint32_t v = y & mask;
            out[kk] = (int16_t)((int)v - radius);
            kk += v <= rng ? 1 : 0;
          }
          k = kk;
          mt.pos = MT_N;
        } else {
          uint32_t v;
          while ((v = (mt.next() & mask)) > rng) {}
          out[k++] = (int16_t)((int)v - radius);
        }
      }
    }
    long long k = 0;
    const int16_t* d0 = draws.data();
    const int16_t* d1 = draws.data() + (size_t)L;
    const int16_t* d2 = draws.data() + (size_t)(ND == 3 ? 2 : 0) * L;
    // (branch-free here too: row k is written and kept only if it counts — k < number_offsets whenever it is written)
    for (long long i = 0; i < L && k < number_offsets; ++i) {
      const int a = d0[i], b2 = d1[i], c = ND == 3 ? d2[i] : 0;
      const int sq = a * a + b2 * b2 + c * c;
      long long* row = offsets + k * ND;
      row[0] = a;
      row[1] = b2;
      if (ND == 3) row[2] = c;
      k += (sq < r2 && sq > 0) ? 1 : 0;
    }
    if (k >= number_offsets) break;                   // (the reference redraws everything when too few survive)
    CLX_REQUIRE(trips < 990, "clx_sample_offsets_mt19937: too few offsets inside the radius, 990 times over (the reference recurses until Python stops it)");
  }
  *pos = mt.pos;
  if (rounds) *rounds = trips;
  return CLX_OK;
}
