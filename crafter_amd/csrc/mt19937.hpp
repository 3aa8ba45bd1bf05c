// MT19937 exactly as numpy's legacy RandomState consumes it (the reference's one RNG per env,
// engine.py:34).  Verified draw semantics: SURVEY.md A.6.
//   * RandomState(seed)         -> init_genrand, pos = 624
//   * uniform() / random_sample -> ((a >> 5) * 2^26 + (b >> 6)) / 2^53 from two words
//   * randint(0, n)             -> masked rejection on 32-bit words, no draw when n == 1
// The twist itself is a wave primitive (wave_gfx950.hpp: lane-parallel batches).
#pragma once
#include "types.hpp"

namespace crafter {

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

// Exact in binary64: (a >> 5) * 2^26 + (b >> 6) < 2^53, division by 2^53 is a scaling.
__device__ __forceinline__ double mt_double(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// uniform() < p without a double: random_sample() = X * 2^-53 with the 53-bit integer X = (a >> 5) * 2^26 + (b >> 6), exact,
// so X * 2^-53 < p  <=>  X < p * 2^53 (a scaling: exact)  <=>  X < ceil(p * 2^53) for the integer X.  The threshold is a
// compile-time constant; the comparison is three scalar instructions on the rule wave's chain instead of two conversions, a
// multiply, an add, a scaling and a compare in f64 (objects.py:277,298,299,333,336,338,339: every creature, every step).
// Checked for every probability of the rules on both sides of each threshold: tests/test_render_identities.py.
constexpr uint64_t mt_prob53(double p) {
  double t = p * 9007199254740992.0;   // exact: a power of two
  uint64_t f = (uint64_t)t;
  return (double)f < t ? f + 1 : f;
}
__device__ __forceinline__ uint64_t mt_x53(uint32_t a, uint32_t b) { return ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6); }

// RandomState.uniform(32, 127) = 32 + 95 * random_sample() (engine.py:209).  random_sample = X * 2^-53 with the 53-bit
// integer X exact in binary64, so 95 * (X * 2^-53) and X * (95 * 2^-53) round the same real number once: one multiply less.
// X = (a >> 5) * 2^26 + (b >> 6) itself is exact whichever way it is evaluated, so the explicit fma (one instruction for
// the scaling and the sum) changes nothing.
__device__ __forceinline__ double mt_uniform_32_127(uint32_t a, uint32_t b) {
  return 32.0 + __builtin_fma((double)(a >> 5), 67108864.0, (double)(b >> 6)) * (95.0 / 9007199254740992.0);
}

// One element of the twist: new[i] from (mt[i], mt[i+1], mt[i+397]) (indices mod 624).
__device__ __forceinline__ uint32_t mt_twist_word(uint32_t cur, uint32_t nxt, uint32_t far) {
  uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
  return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// init_genrand (numpy mt19937_seed): strictly serial recurrence, run by one lane.
__device__ __forceinline__ void mt_seed_serial(uint32_t* mt, uint32_t seed) {
  for (int i = 0; i < MT_N; i++) {
    mt[i] = seed;
    seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)(i + 1);
  }
}

// CPython >= 3.8 tuple hash of (seed, episode) (Objects/tupleobject.c, xxHash-style), then
// ``% (2**31 - 1)`` with Python's non-negative remainder: the world seed of env.py:74.
// seed_lane is hash(seed) computed by CPython on the host; hash(episode) == episode.
__device__ __forceinline__ uint32_t world_seed(uint64_t seed_lane, uint64_t episode) {
  const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P5 = 2870177450012600261ull;
  uint64_t acc = P5;
  uint64_t lanes[2] = {seed_lane, episode};
  for (int i = 0; i < 2; i++) {
    acc += lanes[i] * P2;
    acc = (acc << 31) | (acc >> 33);
    acc *= P1;
  }
  acc += 2ull ^ (P5 ^ 3527539ull);
  int64_t h = (acc == (uint64_t)-1) ? 1546275796ll : (int64_t)acc;
  int64_t m = h % 2147483647ll;
  if (m < 0) m += 2147483647ll;
  return (uint32_t)m;
}

}  // namespace crafter
