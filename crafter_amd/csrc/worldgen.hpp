// Env.reset(): world reseed, player, terrain and initial creatures -- reference env.py:70-81,
// engine.py:33-39, worldgen.py:10-91; semantics SURVEY.md A.4.
//
// The terrain noise (6.3 OpenSimplex evaluations per cell on average) has no side effects, so
// it is evaluated for all cells in parallel; what has to stay in reference order is the
// consumption of the env's MT19937 stream (worldgen.py:43,45,47,58 for materials, 71,73,75 for
// creatures).  Pass 1 therefore classifies every cell into either a final material or a "pending
// chain" code holding the noise-dependent condition bits; pass 2 walks only the pending cells in
// x-major order and draws; pass 3 does the same for creature placement.
#pragma once
#include <math.h>

#include "env_core.hpp"
#include "simplex.hpp"

namespace crafter {

// per-cell code while generating (lives in the LDS `mat` array)
enum : uint8_t {
  WG_MAT_MASK = 0x0F,
  WG_TREE = 0x10,      // pending: grassland cell with simplex(x, y, 5, 7) > 0 -> one draw
  WG_TUNNEL = 0x40,    // tunnels[x, y] (worldgen.py:40,43)
  WG_PENDING = 0x80,   // low bits = conditions c1..c4 of the mountain chain, or WG_TREE
};

constexpr int WG_TABLES_AT = 1024 + 4 * MT_N;               // perm, pg3, source, ridx (seeding scratch) | next MT state
constexpr int WG_LDS_BYTES = WG_TABLES_AT + kSimplexLdsBytes;   // | noise3's tables (simplex.hpp SimplexLds)

// ---- the exponential of worldgen.py:27, pinned ------------------------------------------------------------------------
// The reference calls np.exp, which is Intel SVML on AVX512 hosts and the C library's exp elsewhere; ocml's is a third
// flavour.  Their last-bit differences decide `start > 0.5` (worldgen.py:36) on cells at distance exactly 4 from the player
// on which the noise vanishes -- about one world in 5000 -- and with it every later uniform() draw of that world.  So both
// the oracle (oracle/exp_cr.py) and the device evaluate the CORRECTLY ROUNDED exponential, by the same sequence of
// double-double operations (exact products, see dd_two_prod): x = k ln2 + r with ln2 in three pieces, exp(r / 256) by its
// Taylor series to degree 8, eight squarings.  ~300 operations per cell, once per cell (a noise3 look-up is ~380 and a
// cell needs six and a half).
struct DD {
  double hi, lo;
};
__device__ __forceinline__ DD dd_two_sum(double a, double b) {
  double s = a + b, bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ DD dd_quick_two_sum(double a, double b) {
  double s = a + b;
  return {s, b - (s - a)};
}
// a * b = p + e exactly.  The oracle obtains e with Veltkamp's split and Dekker's product (17 operations, numpy has no fused
// multiply-add); one fused multiply-add returns the very same number -- a * b - p is representable, both compute it
// exactly (no underflow in exp_cr's range) -- so the results stay bit-identical (tests/test_exp_cr.py, tests/test_gpu_noise.py).
__device__ __forceinline__ DD dd_two_prod(double a, double b) {
  double p = a * b;
  return {p, __builtin_fma(a, b, -p)};
}
__device__ __forceinline__ DD dd_mul(DD a, DD b) {
  DD p = dd_two_prod(a.hi, b.hi);
  double e = p.lo + (a.hi * b.lo + a.lo * b.hi);
  return dd_quick_two_sum(p.hi, e);
}
__device__ __forceinline__ DD dd_add(DD a, DD b) {
  DD s = dd_two_sum(a.hi, b.hi);
  double e = s.lo + (a.lo + b.lo);
  return dd_quick_two_sum(s.hi, e);
}
__device__ __attribute__((noinline)) static double exp_cr(double x) {
  const double INV_LN2 = 0x1.71547652b82fep+0, L1 = 0x1.62e42fefa3800p-1, L2 = 0x1.ef35793c76730p-45, L3 = 0x1.f97b57a079a19p-103;
  const DD C[9] = {{1.0, 0.0},
                   {1.0, 0.0},
                   {0.5, 0.0},
                   {0x1.5555555555555p-3, 0x1.5555555555555p-57},
                   {0x1.5555555555555p-5, 0x1.5555555555555p-59},
                   {0x1.1111111111111p-7, 0x1.1111111111111p-63},
                   {0x1.6c16c16c16c17p-10, -0x1.f49f49f49f49fp-65},
                   {0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-73},
                   {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76}};
  double k = __builtin_rint(x * INV_LN2);
  double a = x - k * L1;   // exact: L1 has 42 significant bits
  DD p2 = dd_two_prod(k, L2);
  DD st = dd_two_sum(a, -p2.hi);
  double t = (st.lo - p2.lo) - k * L3;
  DD r = dd_quick_two_sum(st.hi, t);
  r.hi *= 0.00390625;
  r.lo *= 0.00390625;
  DD acc = C[8];
#pragma unroll
  for (int n = 7; n >= 0; n--) acc = dd_add(dd_mul(acc, r), C[n]);
#pragma unroll
  for (int i = 0; i < 8; i++) acc = dd_mul(acc, acc);
  double out = __builtin_ldexp(acc.hi, (int)k);
  // |x| <= 2^-52: exp(x) = 1 + x + d, 0 < d < 2^-105; when 1 + x is exactly half way between two doubles d decides, upwards
  DD one = dd_two_sum(1.0, x);
  double half_up = one.hi < 1.0 ? 0x1p-54 : 0x1p-53;
  uint64_t up;   // the double above one.hi (positive and finite here: about 1)
  __builtin_memcpy(&up, &one.hi, 8);
  up += 1;
  double above;
  __builtin_memcpy(&above, &up, 8);
  double tiny = one.lo == half_up ? above : one.hi;
  return __builtin_fabs(x) <= 0x1p-52 ? tiny : out;
}

template <class W, class S = uint16_t>   // S: the slot map's element type of the Env it generates into (env_core.hpp)
struct WorldGen {
  Env<W, S>& e;
  uint8_t* perm;     // LDS [256]
  uint8_t* pg3;      // LDS [256]
  uint8_t* source;   // LDS [256] scratch for the seeding shuffle
  uint8_t* ridx;     // LDS [256] shuffle indices
  SimplexLds* tab;   // LDS: noise3's gradient and extra-vertex tables
  uint32_t* mtb;     // LDS [624] the MT19937 state AFTER e.mt (random access to >= 624 future words)

  __device__ __forceinline__ WorldGen(Env<W, S>& env, uint8_t* lds) : e(env) {
    perm = lds;
    pg3 = lds + 256;
    source = lds + 512;
    ridx = lds + 768;
    tab = (SimplexLds*)(lds + WG_TABLES_AT);
    mtb = (uint32_t*)(lds + 1024);
  }

  // OpenSimplex(seed) permutation (SURVEY App. B "Seeding"): the 256 LCG draws are independent
  // given the seed (jump-ahead per thread); only the shuffle itself is a serial chain, one LDS
  // round trip per element.
  __device__ __forceinline__ void seed_simplex(int64_t seed) {
    e.w.block_for(256, [&](int i) { ridx[i] = (uint8_t)simplex_shuffle_index(seed, i); });
    e.w.sync();
    e.w.block_for(256, [&](int i) { source[i] = (uint8_t)i; });
    e.w.sync();
    if (e.w.leader()) {
      // The shuffle is one serial chain over a 256-byte array: perm[i] = source[r_i]; source[r_i] = source[i], i = 255 .. 0.
      // By ONE lane, on the LDS arrays themselves: a wave's DS instructions execute in order, so a step costs one LDS round
      // trip (both reads of a step are in flight together; the indices come four to a dword, fetched a group ahead) --
      // about 120 clocks.  Round 5 kept the array in a lane register (v_readlane / v_writelane, no memory access at all):
      // thirty dependent instructions alternating between the scalar and the vector unit per step, 290 clocks (round 6 probe).
      const uint32_t* r4 = (const uint32_t*)ridx;
      uint32_t next = r4[63];
      for (int g = 63; g >= 0; g--) {
        uint32_t rr = next;
        next = r4[g > 0 ? g - 1 : 0];
#pragma unroll
        for (int k = 3; k >= 0; k--) {
          int i = 4 * g + k;
          int r = (int)((rr >> (8 * k)) & 0xFFu);
          uint8_t v = source[r], t = source[i];   // r <= i: source[i] is read before source[r] changes (the same value if r == i)
          perm[i] = v;
          source[r] = t;
        }
      }
    }
    e.w.sync();
    e.w.block_for(256, [&](int i) { pg3[i] = (uint8_t)(perm[i] % 24); });   // Simplex::contrib
    fill_gradients();
    e.w.sync();
  }

  __device__ __forceinline__ void fill_gradients() {   // callers synchronise
    simplex_fill_tables(tab, [&](int n, auto body) { e.w.block_for(n, body); });
  }

  // worldgen.py:79-91 with a single size: 0 + 1 * noise, / 1
  __device__ __forceinline__ static double S1(const Simplex<W>& sx, double x, double y, double z, double size) {
    return sx.noise3(x / size, y / size, z);
  }

  // the material ids classify() writes, read out of the rules once (not per cell)
  struct ClassIds {
    uint8_t grass, path, sand, water, stone, lava;
  };
  __device__ __forceinline__ ClassIds class_ids() const {
    const Rules& R = e.R;
    ClassIds k;
    k.grass = (uint8_t)R.mat_grass; k.path = (uint8_t)R.mat_path; k.sand = (uint8_t)R.mat_sand;
    k.water = (uint8_t)R.mat_water; k.stone = (uint8_t)R.mat_stone; k.lava = (uint8_t)R.mat_lava;
    return k;
  }

  // worldgen.py:21-31: the three fields every cell's material starts from.
  //
  // Far from the player the `start` term needs neither its noise look-up nor its exponential (round 5; one look-up in 6.5 and
  // the 264-instruction exp_cr, for 91 % of the cells of a 256x256 world -- none of a 64x64 one).  What the skip must not
  // change is any COMPARISON made on the fields (the fields themselves are never output):
  //   * |noise3| <= 7.35: at most ten lattice contributions (2 - r^2)^4 (g . d) / 103 with |g| = sqrt(11^2 + 11^2 + 4^2) and
  //     |d| = r: each at most (16/9)^4 sqrt(2/9) x 16.06 / 103 = 0.7343 (the maximum over r, at r^2 = 2/9);
  //   * so at distance >= 48 (kFarD2) the sigmoid's argument is <= 4 - 48 + 2 x 7.35 = -29.3 and start <= exp(-29.3) <
  //     1.9e-13, whatever flavour of exp: `start > 0.5` is false, water moves by 2 start <= 3.8e-13 and mountain by
  //     4 start + 0.3 x 2 start <= 8.8e-13 (+ a few ulps of re-rounding, 1e-16);
  //   * with start taken as 0 the two fields are therefore within 1e-12 of their true values, and every comparison against
  //     them (water: 0.25, 0.3, 0.35; mountain: 0.15, 0.18, 0.3 -- classify / classify_head / resolve's flags) comes out the
  //     same PROVIDED neither field lies within kGuard = 1e-11 of one of its thresholds: that is checked, per cell, and a
  //     cell that fails the check (about one in 10^10) is evaluated in full.
  // tests/test_worldgen_guard.py: 1,000+ worlds of 256x256 and 64x64 generated both ways by the same code (CRAFTER_WG_NO_SKIP),
  // cell for cell; the oracle never skips.
  static constexpr int kFarD2 = 48 * 48;
  __device__ __forceinline__ static bool clear_of(double v, double t) { return v < t - 1e-11 || v > t + 1e-11; }
  __device__ __forceinline__ static void terrain_fields(const Simplex<W>& sx, int x, int y, int px, int py, double& start, double& water,
                                               double& mountain) {
    double fx = (double)x, fy = (double)y;
    int d2 = (x - px) * (x - px) + (y - py) * (y - py);
    double w0 = (0 + 1 * sx.noise3(fx / 15, fy / 15, 3)) + 0.15 * sx.noise3(fx / 5, fy / 5, 3);
    w0 = w0 + 0.1;
    double m0 = (0 + 1 * sx.noise3(fx / 15, fy / 15, 0)) + 0.3 * sx.noise3(fx / 5, fy / 5, 0);
    m0 /= (1 + 0.3);
#ifndef CRAFTER_WG_NO_SKIP
    if (d2 >= kFarD2) {
      double mf = m0 - (4 * 0.0 + 0.3 * w0);
      if (clear_of(w0, 0.25) && clear_of(w0, 0.3) && clear_of(w0, 0.35) && clear_of(mf, 0.15) && clear_of(mf, 0.18) && clear_of(mf, 0.3)) {
        start = 0.0;
        water = w0;
        mountain = mf;
        return;
      }
    }
#endif
    start = 4 - __builtin_sqrt((double)d2);
    start += 2 * S1(sx, fx, fy, 8, 3);
    start = 1 / (1 + exp_cr(-start));
    water = w0 - 2 * start;
    mountain = m0 - (4 * start + 0.3 * water);
  }

  // worldgen.py:21-61 up to (not including) the uniform() draws
  __device__ __forceinline__ static uint8_t classify(const Simplex<W>& sx, const ClassIds& R, int x, int y, int px, int py) {
    double fx = (double)x, fy = (double)y;
    double start, water, mountain;
    terrain_fields(sx, x, y, px, py, start, water, mountain);
    if (start > 0.5) return (uint8_t)R.grass;
    if (mountain > 0.15) {
      if (S1(sx, fx, fy, 6, 7) > 0.15 && mountain > 0.3) return (uint8_t)R.path;          // cave
      if (S1(sx, (double)(2 * x), fy / 5, 7, 3) > 0.4) return (uint8_t)(R.path | WG_TUNNEL);  // horizontal tunnel
      if (S1(sx, fx / 5, (double)(2 * y), 7, 3) > 0.4) return (uint8_t)(R.path | WG_TUNNEL);  // vertical tunnel
      int c1 = S1(sx, fx, fy, 1, 8) > 0;
      int c2 = S1(sx, fx, fy, 2, 6) > 0.4;
      int c3 = mountain > 0.18;
      int c4 = mountain > 0.3 && S1(sx, fx, fy, 6, 5) > 0.35;
      if (!(c1 | c2 | c3)) return (uint8_t)(c4 ? R.lava : R.stone);
      return (uint8_t)(WG_PENDING | c1 | (c2 << 1) | (c3 << 2) | (c4 << 3));
    }
    if (0.25 < water && water <= 0.35 && S1(sx, fx, fy, 4, 9) > -0.2) return (uint8_t)R.sand;
    if (0.3 < water) return (uint8_t)R.water;
    if (S1(sx, fx, fy, 5, 7) > 0) return (uint8_t)(WG_PENDING | WG_TREE);
    return (uint8_t)R.grass;
  }

  // ---- classification in compacted passes (the world pool's classification kernel) ---------------------------------
  // classify() above evaluates its conditional noise look-ups under divergence: a wave runs every branch any of its
  // lanes takes, ~10 noise3 per wave-cell for 6.3 needed per cell.  Here the five unconditional look-ups are evaluated
  // for all cells first; what each cell still needs is a little state machine (next look-up + condition bits), and
  // every round gathers the cells that need a look-up -- of whatever kind: they all run the same noise3 -- into a dense
  // list, one lane per entry.  Same expressions, same comparisons, same results as classify().
  enum : int { NK_CAVES = 0, NK_TUNH, NK_TUNV, NK_C1, NK_C2, NK_C4, NK_SAND, NK_TREE, NK_DONE = 15 };
  enum : int { NF_M18 = 0x10, NF_M30 = 0x20, NF_A = 0x40, NF_B = 0x80 };   // mountain > 0.18, > 0.3; branch-specific: c1 / water > 0.3, c2

  // after the unconditional look-ups: the cell's final code, or its first conditional look-up (returned in `state`)
  __device__ __forceinline__ static uint8_t classify_head(const Simplex<W>& sx, const ClassIds& R, int x, int y, int px, int py,
                                                 int& state) {
    double start, water, mountain;
    terrain_fields(sx, x, y, px, py, start, water, mountain);
    state = NK_DONE;
    if (start > 0.5) return R.grass;
    if (mountain > 0.15) {
      state = NK_CAVES | (mountain > 0.18 ? NF_M18 : 0) | (mountain > 0.3 ? NF_M30 : 0);
      return 0;
    }
    if (0.25 < water && water <= 0.35) {
      state = NK_SAND | (0.3 < water ? NF_A : 0);
      return 0;
    }
    if (0.3 < water) return R.water;
    state = NK_TREE;
    return 0;
  }
  // the look-up a cell in state `st` needs
  __device__ __forceinline__ static double classify_lookup(const Simplex<W>& sx, int st, int x, int y) {
    int kind = st & 15;
    double fx = (double)(kind == NK_TUNH ? 2 * x : x), fy = (double)(kind == NK_TUNV ? 2 * y : y);
    if (kind == NK_TUNV) fx = (double)x / 5;
    if (kind == NK_TUNH) fy = (double)y / 5;
    double size = (kind == NK_CAVES || kind == NK_TREE) ? 7.0 : (kind == NK_TUNH || kind == NK_TUNV) ? 3.0 : (kind == NK_C1) ? 8.0
                  : (kind == NK_C2) ? 6.0 : (kind == NK_C4) ? 5.0 : 9.0;
    double z = (kind == NK_CAVES || kind == NK_C4) ? 6.0 : (kind == NK_TUNH || kind == NK_TUNV) ? 7.0 : (kind == NK_C1) ? 1.0
               : (kind == NK_C2) ? 2.0 : (kind == NK_SAND) ? 4.0 : 5.0;
    return sx.noise3(fx / size, fy / size, z);
  }
  // applies the look-up's value: the next state, and the final code once the state is NK_DONE
  __device__ __forceinline__ static int classify_advance(const ClassIds& R, int st, double v, uint8_t& code) {
    int kind = st & 15, flags = st & 0xF0;
    switch (kind) {
      case NK_CAVES:
        if (v > 0.15 && (flags & NF_M30)) { code = R.path; return NK_DONE; }
        return NK_TUNH | flags;
      case NK_TUNH:
        if (v > 0.4) { code = (uint8_t)(R.path | WG_TUNNEL); return NK_DONE; }
        return NK_TUNV | flags;
      case NK_TUNV:
        if (v > 0.4) { code = (uint8_t)(R.path | WG_TUNNEL); return NK_DONE; }
        return NK_C1 | flags;
      case NK_C1:
        return NK_C2 | flags | (v > 0 ? NF_A : 0);
      case NK_C2: {
        flags |= (v > 0.4 ? NF_B : 0);
        if (flags & NF_M30) return NK_C4 | flags;   // c4 = mountain > 0.3 && simplex(x, y, 6, 5) > 0.35: short-circuit
        v = 0.0;                                    // c4 = 0
      }  // fall through
      case NK_C4: {
        int c1 = (flags & NF_A) != 0, c2 = (flags & NF_B) != 0, c3 = (flags & NF_M18) != 0;
        int c4 = (kind == NK_C4) && v > 0.35;
        code = !(c1 | c2 | c3) ? (c4 ? R.lava : R.stone) : (uint8_t)(WG_PENDING | c1 | (c2 << 1) | (c3 << 2) | (c4 << 3));
        return NK_DONE;
      }
      case NK_SAND:
        if (v > -0.2) { code = R.sand; return NK_DONE; }
        if (flags & NF_A) { code = R.water; return NK_DONE; }
        return NK_TREE;
      default:  // NK_TREE
        code = v > 0 ? (uint8_t)(WG_PENDING | WG_TREE) : R.grass;
        return NK_DONE;
    }
  }

  // ---- random access into the env's MT19937 stream -------------------------------------------
  // e.mt holds the current state (words [mt_pos, 624) unconsumed), mtb the state after it, so
  // any of the next >= 624 raw words can be read by any lane.  advance() consumes words and
  // rolls the pair forward; when the window is closed e.mt / e.mt_pos are the ordinary
  // RandomState again.
  __device__ __forceinline__ void window_open() {
    e.rng_invalidate();
    if (e.mt_pos >= MT_N) {
      e.w.mt_twist(e.mt);
      e.mt_pos = 0;
    }
    e.w.wave_for(MT_N, [&](int i) { mtb[i] = e.mt[i]; });
    e.w.wsync();
    e.w.mt_twist(mtb);
  }
  __device__ __forceinline__ uint32_t wword(int k) const {
    int idx = e.mt_pos + k;
    return idx < MT_N ? e.mt[idx] : mtb[idx - MT_N];
  }
  __device__ __forceinline__ double wdouble(int d) const { return mt_double(mt_temper(wword(2 * d)), mt_temper(wword(2 * d + 1))); }
  __device__ __forceinline__ void advance(int nwords) {  // nwords <= 624
    e.mt_pos += nwords;
    if (e.mt_pos >= MT_N) {
      e.w.wave_for(MT_N, [&](int i) { e.mt[i] = mtb[i]; });
      e.w.wsync();
      e.w.mt_twist(mtb);
      e.mt_pos -= MT_N;
    }
  }

  __device__ __forceinline__ static int draws_of(int code) {  // uniform() calls the cell makes if no chain stops early
    return (code & WG_TREE) ? 1 : __builtin_popcount(code & 7);
  }

  // ---- the ordered uniform() draws of worldgen.py:43-50,58 (materials) and 71-75 (creatures) ---------------------
  // Which threshold a double of the stream is compared with depends on how many draws the cells before it consumed,
  // but there are only four (three) thresholds.  So every double is tempered and compared with ALL of them exactly
  // once, 64 stream positions at a time (one per lane), and the outcomes are kept as wave-uniform bit masks -- a
  // window of two 64-position chunks (lo: relative positions 0..63, hi: 64..127; the consumed position p stays below
  // 64).  Resolving the cells is then bit arithmetic: 64 cells per round, every lane assumes the lanes before it
  // consume their full chains, picks its bits out of the masks and resolves; the first lane whose chain stopped early
  // (coal / iron / cow / zombie found with draws left), or whose chain would leave the window, ends the round, the
  // lanes after it are re-run.  No LDS access and no floating point inside the rounds.
  struct DrawWindow {
    uint64_t lo[4], hi[4];
    int p;   // doubles consumed since the window's base (e.mt_pos): 0..63 between rounds
  };
  template <int NTHR>
  __device__ __forceinline__ void eval_chunk(int first, const double (&thr)[NTHR], uint64_t (&m)[4]) {
    W& w = e.w;
    w.lane_set(1, 0, 64, [&](int l, int) -> uint32_t {
      double d = wdouble(first + l);
      uint32_t bits = 0;
#pragma unroll
      for (int k = 0; k < NTHR; k++) bits |= (d > thr[k]) ? (1u << k) : 0u;
      return bits;
    });
#pragma unroll
    for (int k = 0; k < 4; k++) m[k] = k < NTHR ? W::uni64(w.lane_ballot(1, 1u << k)) : 0ull;
  }
  template <int NTHR>
  __device__ __forceinline__ void window_begin(DrawWindow& dw, const double (&thr)[NTHR]) {
    dw.p = 0;
    eval_chunk(0, thr, dw.lo);
    eval_chunk(64, thr, dw.hi);
    window_deal<NTHR>(dw);
  }
  // after a round: drop the lo chunk once it is used up (stream base moves 64 doubles = 128 words on)
  template <int NTHR>
  __device__ __forceinline__ void window_roll(DrawWindow& dw, const double (&thr)[NTHR]) {
    if (dw.p >= 64) {
      advance(128);
#pragma unroll
      for (int k = 0; k < 4; k++) dw.lo[k] = dw.hi[k];
      eval_chunk(64, thr, dw.hi);
      dw.p -= 64;
      window_deal<NTHR>(dw);
    }
  }
  __device__ __forceinline__ void window_end(DrawWindow& dw) { advance(2 * dw.p); }
  // bit `pos` (0..127) of threshold mask t; 32-bit selects and one bit-field extract per lane
  __device__ __forceinline__ static uint32_t wbit(const DrawWindow& dw, int t, int pos) {
    uint64_t m = (pos & 64) ? dw.hi[t] : dw.lo[t];
    uint32_t half = (pos & 32) ? (uint32_t)(m >> 32) : (uint32_t)m;
    return (half >> (pos & 31)) & 1u;
  }

  // The window's threshold masks, dealt to the lanes for per-lane bit-field extraction: lane i of register 5 + t holds bits
  // [32 i - 16, 32 i + 16) of mask t (positions outside the window's [0, 128): 0).  Refreshed whenever the masks change.
  __device__ __forceinline__ static uint32_t wword(const DrawWindow& dw, int t, int i) {   // i: 0 .. 63
    uint64_t lo = dw.lo[t], hi = dw.hi[t];
    switch (i) {
      case 0: return (uint32_t)(lo << 16);
      case 1: return (uint32_t)(lo >> 16);
      case 2: return (uint32_t)((lo >> 48) | (hi << 16));
      case 3: return (uint32_t)(hi >> 16);
      case 4: return (uint32_t)(hi >> 48);
      default: return 0u;
    }
  }
  template <int NTHR>
  __device__ __forceinline__ void window_deal(const DrawWindow& dw) {
    W& w = e.w;
#pragma unroll
    for (int t = 0; t < NTHR; t++) w.lane_set(5 + t, 0, 64, [&](int, int l) -> uint32_t { return wword(dw, t, l); });
  }
  // 16 consecutive bits of threshold mask t, ending at window position `end` (>= 0): bit j = position end - 15 + j.  Two
  // register look-ups across the lanes and a funnel shift, no branch (round 6, first form: 64-bit shifts of the masks under
  // four-way divergence -- 60 instructions per field, four fields per lane and round).
  __device__ __forceinline__ uint32_t wfield(int t, int end, int l) const {
    uint32_t s = (uint32_t)(end + 1);   // bit offset into the dealt string (which starts 16 bits before the window)
    int i = (int)(s >> 5);
    uint32_t lo = e.w.lane_fetch(5 + t, i, l), hi = e.w.lane_fetch(5 + t, i + 1, l);
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31u)) & 0xFFFFu;
  }
  __device__ __forceinline__ static uint64_t lanes_below(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

  // One round of ordered draws over the (<= 64) cells of `active`, lane = cell.  What a cell draws is a chain of up to three
  // links -- lane slot 0, bits 0 / 1 / 2: the cell has link a / b / c, compared with threshold 0 / 1 / 2 in this order, and
  // the chain ENDS at the first link that hits (worldgen.py:43-47: coal, iron, diamond; 71-75: cow, zombie, skeleton) -- or
  // a single draw against threshold 3 (bit 3; worldgen.py:58: tree).  Where in the stream a cell draws depends on how many
  // draws every cell before it used, i.e. on where their chains ended.
  //
  // Round 5 let every lane assume that the lanes before it use all their links, and re-ran the round's lanes behind the
  // first chain that ended early -- a fresh pass over the round per early end.  Now an early end costs a dozen instructions.
  // An early end only SHIFTS the draws of the cells behind it, by 1 or 2 positions towards the front; so every lane
  // evaluates its chain for ALL shifts 0 .. 15 at once -- 16 consecutive bits of each threshold mask are the link's
  // outcomes under the 16 shifts, and a chain's logic on them is bitwise (SWAR over the shifts) -- and the round is
  // resolved by a scalar walk from early end to early end: "the first lane behind the last early end whose chain ends early
  // under the shift accumulated so far" is one ballot.  The walk stops where the shift would pass 15, or where a chain would
  // leave the 128-position window; the lanes behind that point wait for the next round (the window has rolled by then).
  //
  // Returns the lanes resolved; for those, lane slot 1 = hit (0: none, 1 / 2 / 3: link a / b / c; single draw: 1) | draws used << 8.
  // Lane registers: 0 links (in), 1 result (out), 3 / 4 outcome vectors, 5 .. 8 the dealt masks (window_deal), 9 first position.
  template <int NTHR>
  __device__ __forceinline__ uint64_t chain_round(const DrawWindow& dw, uint64_t active) {
    W& w = e.w;
    const uint64_t fa = w.lane_ballot(0, 1u) & active, fb = w.lane_ballot(0, 2u) & active, fc = w.lane_ballot(0, 4u) & active,
                   fs = NTHR > 3 ? w.lane_ballot(0, 8u) & active : 0ull;
    const int p = dw.p;
    // where the lane's chain starts if every chain before it runs to its end
    w.lane_set(9, 0, 64, [&](int, int l) -> uint32_t {
      return (uint32_t)w.count_below(fs, l, w.count_below(fc, l, w.count_below(fb, l, w.count_below(fa, l, p))));
    });
    // slot 3: hits of link a (or of the single draw) | hits of link b << 16; slot 4: hits of link c | "ends early" << 16 --
    // bit j of each: under shift 15 - j
    w.lane_set2_all(3, 4, [&](int l) -> uint64_t {
      uint32_t links = (active >> l) & 1ull ? w.lane_get(0, l) : 0u;
      int B = (int)w.lane_get(9, l);
      int a = links & 1u, b = (links >> 1) & 1u, c = (links >> 2) & 1u;
      // (every look-up by every lane, then masked: a cross-lane fetch under divergence would find its source lanes switched off)
      uint32_t Fa = wfield(0, B, l) & (a ? 0xFFFFu : 0u);
      if (NTHR > 3) Fa |= wfield(3, B, l) & ((links & 8u) ? 0xFFFFu : 0u);
      uint32_t Fb = wfield(1, B + a, l) & (b ? 0xFFFFu : 0u);
      uint32_t Fc = wfield(2, B + a + b, l) & (c ? 0xFFFFu : 0u);
      uint32_t ha = Fa, hb = Fb & ~Fa, hc = Fc & ~Fa & ~Fb;
      uint32_t early = ((b + c) != 0 ? ha : 0u) | (c ? hb : 0u);   // link a hit with b or c left; link b hit with c left
      return (uint64_t)(ha | (hb << 16)) | ((uint64_t)(hc | (early << 16)) << 32);
    });
    uint64_t rem = active, st1 = 0, st2 = 0;
    int shift = 0, limit = 64;
    for (;;) {
      uint64_t m = W::uni64(w.lane_ballot(4, 1u << (16 + 15 - shift)) & rem);
      if (!m) break;
      int d = __builtin_ctzll(m);
      // the chain of lane d ends early under the shift so far: by two draws if its link a hit with both b and c left, else by one
      bool two = (w.lane_read(0, d) & 7u) == 7u && ((w.lane_read(3, d) >> (15 - shift)) & 1u) != 0;
      if (two) st2 |= 1ull << d; else st1 |= 1ull << d;
      shift += two ? 2 : 1;
      rem = active & ~lanes_below(d + 1);
      if (shift > 15) {   // the lanes behind d would need a 17th shift: next round
        limit = d + 1;
        break;
      }
    }
    w.lane_set(1, 0, 64, [&](int, int l) -> uint32_t {
      if (!((active >> l) & 1ull) || l >= limit) return 0u;
      int Sl = w.count_below(st2, l, w.count_below(st2, l, w.count_below(st1, l, 0)));
      uint32_t links = w.lane_get(0, l);
      int a = links & 1u, b = (links >> 1) & 1u, c = (links >> 2) & 1u, single = NTHR > 3 ? (links >> 3) & 1u : 0;
      int full = a + b + c + single;
      uint32_t v3 = w.lane_get(3, l), v4 = w.lane_get(4, l);
      int j = 15 - Sl;
      int hit = ((v3 >> j) & 1u) ? 1 : ((v3 >> (16 + j)) & 1u) ? 2 : ((v4 >> j) & 1u) ? 3 : 0;
      int used = hit == 1 ? 1 : hit == 2 ? a + 1 : full;
      int outside = ((int)w.lane_get(9, l) - Sl + full > 128) ? 1 : 0;
      return (uint32_t)(hit | (used << 8) | (outside << 11));
    });
    uint64_t out = W::uni64(w.lane_ballot(1, 1u << 11));
    if (out) limit = imin(limit, __builtin_ctzll(out));
    return active & lanes_below(limit);
  }
  __device__ __forceinline__ int count_used(uint64_t commit) {
    W& w = e.w;
    uint64_t u0 = w.lane_ballot(1, 1u << 8) & commit;
    uint64_t u1 = w.lane_ballot(1, 1u << 9) & commit;
    return __builtin_popcountll(u0) + 2 * __builtin_popcountll(u1);
  }

  // The cells that draw at all, densely: a round of chain_round costs its ~300 instructions whether 5 or 64 of its lanes
  // hold a cell with links (materials: a third of the cells, in clumps; creatures: three fifths), so the cells with links
  // of the MATERIAL pass are listed first -- in cell order, which is the order of the draws -- and its rounds run over the list
  // (round 6: 19 dense rounds per 64x64 world instead of 36 sparse ones; material draws 166 k -> 137 k clocks per world).  The list lives where noise3's tables were (dead once the
  // terrain is classified): chunk-relative 16-bit cell numbers, kListCap per chunk.
  static constexpr int kListCap = (kSimplexLdsBytes / 2) & ~63;
  __device__ __forceinline__ uint16_t* draw_list() const { return (uint16_t*)tab; }
  // lists the cells of [from, cells) with links_of(cell) != 0 until the list is full or the chunk spans 2^16 cells; returns
  // the count, `from` = the first cell not looked at yet
  template <class F>
  __device__ __forceinline__ int build_draw_list(int& from, int cells, F links_of) {
    W& w = e.w;
    uint16_t* list = draw_list();
    const int first = from;
    int count = 0;
    while (from < cells && count + 64 <= kListCap && from - first < 65536 - 64) {
      uint64_t m = W::uni64(w.ballot(from, cells, [&](int i) { return links_of(i) != 0u; }));
      if (m) {
        w.lanes(from, cells, [&](int i, int l) {
          if ((m >> l) & 1ull) list[count + __builtin_popcountll(m & ((1ull << l) - 1ull))] = (uint16_t)(i - first);
        });
        count += __builtin_popcountll(m);
      }
      from += 64;
    }
    w.wsync();
    return count;
  }

  // pass 2: materials (worldgen.py:43-50,58).  Lane slot 0: the cell's links (chain_round) | c4 << 4 (lava instead of stone when
  // nothing hits); lane slot 2: the cell
  __device__ __forceinline__ void resolve_materials(int cells) {
    const Rules& R = e.R;
    W& w = e.w;
    const double thr[4] = {0.85, 0.75, 0.994, 0.8};   // coal, iron, diamond (worldgen.py:43-47), tree (worldgen.py:58)
    DrawWindow dw;
    window_begin(dw, thr);
    auto links_of = [&](int i) -> uint32_t {
      int code = e.mat[i];
      if (!(code & WG_PENDING)) return 0u;
      return (code & WG_TREE) ? 8u : (uint32_t)((code & 7) | ((code & 8) << 1));
    };
    const uint16_t* list = draw_list();
    for (int from = 0; from < cells;) {
      const int first = from;
      const int count = build_draw_list(from, cells, links_of);
      for (int lb = 0; lb < count; lb += 64) {
        w.lane_set(2, lb, count, [&](int k, int) -> uint32_t { return (uint32_t)(first + list[k]); });
        w.lane_set(0, lb, count, [&](int, int l) -> uint32_t { return links_of((int)w.lane_get(2, l)); });
        uint64_t active = lanes_below(count - lb);
        while (active) {
          uint64_t commit = chain_round<4>(dw, active);
          w.lanes(lb, count, [&](int, int l) {
            if (!((commit >> l) & 1ull)) return;
            uint32_t links = w.lane_get(0, l), hit = w.lane_get(1, l) & 0xFFu;
            int m = (links & 8u) ? (hit ? R.mat_tree : R.mat_grass)
                    : hit == 1 ? R.mat_coal : hit == 2 ? R.mat_iron : hit == 3 ? R.mat_diamond : (links & 16u) ? R.mat_lava : R.mat_stone;
            e.mat[w.lane_get(2, l)] = (uint8_t)m;
          });
          w.wsync();
          dw.p += count_used(commit);
          active &= ~commit;
          window_roll(dw, thr);
        }
      }
    }
    window_end(dw);
  }

  // pass 3: creature placement, worldgen.py:64-76, same scheme: links a / b / c = the cow / zombie / skeleton draw the cell
  // can reach; a hit ends the chain.  Over the cells as they lie (three cells in five draw: listing them first, as the material
  // pass does, cost more than the 25 rounds it saved -- creature draws 236 k -> 262 k clocks per world, round 6 probe).
  __device__ __forceinline__ void place_creatures(int cells, int px, int py) {
    const Config& c = e.cfg;
    const Rules& R = e.R;
    W& w = e.w;
    const double thr[3] = {0.985, 0.993, 0.95};   // cow, zombie, skeleton (worldgen.py:71-75)
    DrawWindow dw;
    window_begin(dw, thr);
    for (int base = 0; base < cells; base += 64) {
      w.lane_set(0, base, cells, [&](int i, int) -> uint32_t {
        int code = e.mat[i];
        int mat = code & WG_MAT_MASK;
        if (!((R.walkable_mask >> mat) & 1u)) return 0u;
        int x = i / c.H, y = i - x * c.H;
        int d2 = (x - px) * (x - px) + (y - py) * (y - py);
        uint32_t g = (d2 > 9 && mat == R.mat_grass);                       // dist > 3 and grass
        uint32_t z = (d2 > 100);                                           // dist > 10
        uint32_t sk = (mat == R.mat_path && (code & WG_TUNNEL) != 0);      // tunnel path
        return g | (z << 1) | (sk << 2);
      });
      uint64_t active = W::uni64(w.lane_ballot(0, 7u));
      while (active) {
        uint64_t commit = chain_round<3>(dw, active);
        uint64_t born = W::uni64(w.lane_ballot(1, 0xFFu)) & commit;
        while (born) {  // World.add in cell order (slots and chunk keys are order sensitive)
          int l = __builtin_ctzll(born);
          born &= born - 1;
          int i = base + l;
          int x = i / c.H, y = i - x * c.H;
          int hit = (int)(w.lane_read(1, l) & 0xFF);
          int type = hit == 1 ? T_COW : hit == 2 ? T_ZOMBIE : T_SKELETON;
          e.obj_add(type, x, y, type == T_ZOMBIE ? 5 : 3, 0, 0, 0);
        }
        dw.p += count_used(commit);
        active &= ~commit;
        window_roll(dw, thr);
      }
    }
    window_end(dw);
  }

  // RandomState(seed): init_genrand, a serial recurrence -- run on the scalar unit, 64 words collected in a lane
  // register per LDS store.  Wave 0 only.
  __device__ __forceinline__ void init_mt(uint32_t wseed) {
    uint32_t s = (uint32_t)W::uni((int)wseed);
    for (int i0 = 0; i0 < MT_N; i0 += 64) {
      int nb = MT_N - i0 < 64 ? MT_N - i0 : 64;
      for (int j = 0; j < nb; j++) {
        e.w.lane_put(0, j, s);
        s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)(i0 + j + 1);
      }
      e.w.lanes(i0, MT_N, [&](int i, int l) { e.mt[i] = e.w.lane_get(0, l); });
    }
    e.w.wsync();
  }

  // env.py:70-81
  __device__ __forceinline__ void reset_env(uint64_t* prof = nullptr) {
    auto stamp = [&](int k) {
      if (prof && e.w.leader()) prof[k] = e.w.clock();
    };
    const Config& c = e.cfg;
    const Rules& R = e.R;
    EnvRec* rec = e.rec;
    int cells = c.W * c.H;
    int nch = c.nchunk_x * c.nchunk_y;
    int episode = rec->episode + 1;
    uint32_t wseed = world_seed(rec->seed_lane, (uint64_t)episode);
    e.w.sync();
    e.begin_episode(episode);
    // World.reset engine.py:33-39
    e.w.block_for(cells, [&](int i) {
      if (e.objmap) e.objmap[i] = 0;
      if (e.g_objmap && (const void*)e.g_objmap != (const void*)e.objmap) e.g_objmap[i] = 0;
    });
    e.w.block_for(nch, [&](int i) { e.chunk_seen[i] = 0; e.chunk_order[i] = 0; });
    e.clear_creature_counts();
    if (e.w.wave0()) {
      init_mt(wseed);
      e.st(&rec->nchunks_seen, 0);
      Obj z;
      z.type = T_NONE; z.health = 0; z.fx = 0; z.fy = 0; z.x = 0; z.y = 0; z.aux = 0; z.pad = 0;
      e.st(e.objs, z);
    }
    e.mt_pos = MT_N;
    e.nobj = 1;
    e.dirty_slots = 0;
    e.w.sync();
    int px = c.W / 2, py = c.H / 2;
    if (e.w.wave0()) e.obj_add(T_PLAYER, px, py, 0, 0, 1, 0);  // facing (0, 1) objects.py:72; slot 1
    e.w.sync();
    // worldgen.py:11: OpenSimplex(seed=randint(0, 2**31 - 1))
    uint32_t sseed = 0;
    if (e.w.wave0()) sseed = e.randint(2147483647u);
    sseed = e.w.bcast_from_wave0(sseed);
    stamp(9);
    seed_simplex((int64_t)sseed);
    stamp(10);
    // pass 1: classify every cell (parallel over the whole workgroup)
    Simplex<W> sx{perm, pg3, tab};
    const ClassIds ids = class_ids();
    e.w.block_for(cells, [&](int i) {
      int x = i / c.H, y = i - x * c.H;
      e.mat[i] = classify(sx, ids, x, y, px, py);
    });
    e.w.sync();
    stamp(11);
    if (e.w.wave0()) {
      window_open();
      resolve_materials(cells);
      stamp(12);
      place_creatures(cells, px, py);
      stamp(13);
    }
    e.w.sync();
    // strip the generation flags, publish the material map
    e.w.block_for(cells, [&](int i) {
      uint8_t m = e.mat[i] & WG_MAT_MASK;
      e.mat[i] = m;
      if (e.g_mat != e.mat) e.g_mat[i] = m;
    });
    e.w.sync();
  }
};

}  // namespace crafter
