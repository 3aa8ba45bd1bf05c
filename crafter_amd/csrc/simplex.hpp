// 3-D OpenSimplex (K. Spencer 2014, legacy OpenSimplex) in binary64 for gfx950.
//
// This is the arithmetic the reference obtains from the un-vendored PyPI package ``opensimplex``
// (call sites worldgen.py:11,84-87); the algorithm is restated from its public-domain
// publication as specified in SURVEY.md App. B.  Operation order is part of the spec: every
// expression below is written in the published association order and the library is compiled
// with -ffp-contract=off (v_mul_f64 / v_add_f64, never v_fma_f64), because the terrain
// thresholds are compared in full double precision.
#pragma once
#include <stdint.h>

namespace crafter {

// ---- The two "extra" vertices of a point, as a table ------------------------------------------------------------------
// Every displacement the published code writes down has the form ((d0 - i_pre) - s * SQUISH) - post along each axis:
// i_pre in -1..2 is the vertex's lattice offset as it appears inside the expression (d0 + 1 is d0 - (-1)), s the sum of
// the vertex's three offsets (s * SQUISH: 0, SQ, 2 * SQ, 3 * SQ -- exactly the constants of the source, the products by
// 0, 1, 2 are exact and 3 * SQ is the same rounded product), and post is non-zero only for the two places where the
// source subtracts AFTER the squish term (`dy_ext -= 1` in the second tetrahedron, `dx_ext1 -= 2` in the octahedron);
// subtracting 0.0 changes nothing.  So the three regions of the simplectic honeycomb only decide small integers: which
// lattice vertices contribute, and (i_pre, post) per axis.  The regions' own vertices are corners of the unit cube and are
// summed by one pass over the eight corners (see noise3); only the two "extra" vertices of a region are data:
// code = 4 bits per axis, (i_pre + 1) | post << 2.
//
// Which two they are is decided by SIX comparisons per region (the published code's own comparisons, with its own
// strictness), so a point's extras are a table look-up: index = region << 6 | six outcome bits (simplex_extras below is
// the published decision tree restated over those bits; make_simplex_tables() evaluates it for all 192 indices at compile
// time and numbers the distinct vertices).  A wavefront, whose lanes sit in all three regions, used to run all three
// decision trees one after the other; now every lane computes its bits without a branch and reads its two vertices.
constexpr uint32_t sx_field(int i_pre, int post = 0) { return (uint32_t)(i_pre + 1) | ((uint32_t)post << 2); }
constexpr uint32_t sx_vertex(int i, int j, int k) { return sx_field(i) | (sx_field(j) << 4) | (sx_field(k) << 8); }

// region 0: tetrahedron at (0,0,0) (in_sum <= 1); bits: 0 xins >= yins, 1 zins > yins, 2 zins > xins, then with (as, bs) the
//   two largest of the three as the source picks them: 3 wins > as, 4 wins > bs, 5 bs > as       (wins = 1 - in_sum)
// region 1: tetrahedron at (1,1,1) (in_sum >= 2): the same six with every comparison mirrored (<= for >=, < for >;
//   wins = 3 - in_sum) -- noise3 evaluates them as region 0's comparisons on the negated quantities
// region 2: octahedron; bits: 0 xins + yins > 1, 1 xins + zins > 1, 2 yins + zins > 1, then with the scores (as, bs, sc)
//   of the source: 3 as <= bs, 4 as < sc, 5 bs < sc
constexpr void simplex_extras(int index, uint32_t& e0, uint32_t& e1) {
  int region = index >> 6;
  bool c0 = index & 1, c1 = index & 2, c2 = index & 4, c3 = index & 8, c4 = index & 16, c5 = index & 32;
  if (region == 0) {
    int ap = 1, bp = 2;
    if (c0 && c1) bp = 4; else if (!c0 && c2) ap = 4;
    if (c3 || c4) {
      int c = c5 ? bp : ap;
      int xe0 = 0, xe1 = 0, ye0 = 0, ye1 = 0, ze0 = 0, ze1 = 0;
      if ((c & 1) == 0) { xe0 = -1; xe1 = 0; } else { xe0 = xe1 = 1; }
      if ((c & 2) == 0) {
        ye0 = ye1 = 0;
        if ((c & 1) == 0) ye1 = -1; else ye0 = -1;
      } else {
        ye0 = ye1 = 1;
      }
      if ((c & 4) == 0) { ze0 = 0; ze1 = -1; } else { ze0 = ze1 = 1; }
      e0 = sx_vertex(xe0, ye0, ze0);
      e1 = sx_vertex(xe1, ye1, ze1);
    } else {
      int c = ap | bp;
      e0 = sx_vertex((c & 1) ? 1 : 0, (c & 2) ? 1 : 0, (c & 4) ? 1 : 0);
      e1 = sx_vertex((c & 1) ? 1 : -1, (c & 2) ? 1 : -1, (c & 4) ? 1 : -1);
    }
  } else if (region == 1) {
    int ap = 6, bp = 5;
    if (c0 && c1) bp = 3; else if (!c0 && c2) ap = 3;
    if (c3 || c4) {
      int c = c5 ? bp : ap;
      uint32_t x0 = (c & 1) ? sx_field(2) : sx_field(0), x1 = (c & 1) ? sx_field(1) : sx_field(0);
      uint32_t y0 = sx_field(0), y1 = sx_field(0);
      if (c & 2) {
        y0 = y1 = sx_field(1);
        if (c & 1) y1 = sx_field(1, 1); else y0 = sx_field(1, 1);   // dy_ext = dy0 - 1 - 3 * SQ, then `dy_ext -= 1`
      }
      uint32_t z0 = (c & 4) ? sx_field(1) : sx_field(0), z1 = (c & 4) ? sx_field(2) : sx_field(0);
      e0 = x0 | (y0 << 4) | (z0 << 8);
      e1 = x1 | (y1 << 4) | (z1 << 8);
    } else {
      int c = ap & bp;
      e0 = sx_vertex((c & 1) ? 1 : 0, (c & 2) ? 1 : 0, (c & 4) ? 1 : 0);
      e1 = sx_vertex((c & 1) ? 2 : 0, (c & 2) ? 2 : 0, (c & 4) ? 2 : 0);
    }
  } else {
    int ap = c0 ? 3 : 4, bp = c1 ? 5 : 2;
    bool af = c0, bf = c1;
    if (c3 && c4) { ap = c2 ? 6 : 1; af = c2; }
    else if (!c3 && c5) { bp = c2 ? 6 : 1; bf = c2; }
    if (af == bf) {
      if (af) {  // both closest points on the (1,1,1) side
        e0 = sx_vertex(1, 1, 1);
        int c = ap & bp;
        e1 = (c & 1) ? sx_vertex(2, 0, 0) : (c & 2) ? sx_vertex(0, 2, 0) : sx_vertex(0, 0, 2);
      } else {  // both on the (0,0,0) side
        e0 = sx_vertex(0, 0, 0);
        int c = ap | bp;
        e1 = ((c & 1) == 0) ? sx_vertex(-1, 1, 1) : ((c & 2) == 0) ? sx_vertex(1, -1, 1) : sx_vertex(1, 1, -1);
      }
    } else {  // one point on each side
      int c1_ = af ? ap : bp, c2_ = af ? bp : ap;
      e0 = ((c1_ & 1) == 0) ? sx_vertex(-1, 1, 1) : ((c1_ & 2) == 0) ? sx_vertex(1, -1, 1) : sx_vertex(1, 1, -1);
      // dx_ext1 = dx0 - 2 * SQ on every axis, then `-= 2` on one of them
      e1 = (c2_ & 1) ? (sx_field(0, 2) | (sx_field(0) << 4) | (sx_field(0) << 8))
           : (c2_ & 2) ? (sx_field(0) | (sx_field(0, 2) << 4) | (sx_field(0) << 8))
                       : (sx_field(0) | (sx_field(0) << 4) | (sx_field(0, 2) << 8));
    }
  }
}

struct SimplexTables {
  static constexpr int kMaxVerts = 64;
  uint16_t pair[192];        // region index -> vertex numbers of the two extras (low byte, high byte)
  uint16_t code[kMaxVerts];  // vertex number -> code
  int nv;
};
constexpr SimplexTables make_simplex_tables() {
  SimplexTables t{};
  for (int index = 0; index < 192; index++) {
    uint32_t e[2] = {0, 0};
    simplex_extras(index, e[0], e[1]);
    int id[2] = {0, 0};
    for (int s = 0; s < 2; s++) {
      int v = 0;
      while (v < t.nv && t.code[v] != e[s]) v++;
      if (v == t.nv) t.code[t.nv++] = (uint16_t)e[s];
      id[s] = v;
    }
    t.pair[index] = (uint16_t)(id[0] | (id[1] << 8));
  }
  return t;
}
__device__ const SimplexTables kSimplexTables = make_simplex_tables();
constexpr int kSimplexVerts = make_simplex_tables().nv;
static_assert(kSimplexVerts <= SimplexTables::kMaxVerts, "vertex table too small");

// What noise3 reads besides the permutation (LDS on the device; simplex_fill_tables writes it once per workgroup):
// Row strides are chosen for the 64 four-byte banks of the gfx950 LDS: a wavefront's lanes hold different gradient / vertex
// numbers, each of noise3's look-ups is ONE 8-byte read per lane at a fixed offset inside the lane's row, and two rows
// collide when their starts are a multiple of 256 bytes apart.  With rows of 32 bytes (gradients k and k + 8) and 64 bytes
// (vertices v and v + 4) that was the rule, not the exception: 61 % of the classification kernel's LDS-active cycles were
// bank-conflict cycles (round 5, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).  40-byte rows put the 24 gradients on 24
// different (even) banks -- 10 k mod 64 is injective on 0..23 -- and 72-byte rows do the same for the vertices (18 v mod 64,
// v < 32); lanes that ask for the same row are a broadcast.
struct SimplexLds {
  double grad[24][5];              // gradient k: its three components (+ two unused: the stride)
  double vert[kSimplexVerts][9];   // vertex v: (double)i_pre per axis, s * SQ, (double)post per axis, then (low dword) the lattice offsets i | j << 8 | k << 16 as signed bytes (+ one unused)
  uint16_t pair[192];
};
static_assert(kSimplexVerts <= 32, "72-byte vertex rows are bank-conflict free for fewer than 32 vertices only");
constexpr int kSimplexLdsBytes = (int)((sizeof(SimplexLds) + 15) / 16 * 16);

// The 24 gradients are the sign/axis permutations of (11, 4, 4) in this order:
//   k = 3 * q + a:  axis a carries the 11;  x is negative unless q & 1;  y negative if q & 2;  z negative if q & 4.
__host__ __device__ inline double simplex_gradient(int k, int axis) {
  int a = k % 3, q = k / 3;
  double m = (a == axis) ? 11.0 : 4.0;
  bool neg = axis == 0 ? !(q & 1) : axis == 1 ? (q & 2) != 0 : (q & 4) != 0;
  return neg ? -m : m;
}
// f(n, body): body(i) for i in [0, n), spread over the caller's threads (block_for of a wave policy, or a plain loop)
template <class For>
__device__ __forceinline__ void simplex_fill_tables(SimplexLds* t, For each) {
  each(24, [&](int k) {
    t->grad[k][0] = simplex_gradient(k, 0);
    t->grad[k][1] = simplex_gradient(k, 1);
    t->grad[k][2] = simplex_gradient(k, 2);
    t->grad[k][3] = 0.0;
    t->grad[k][4] = 0.0;
  });
  each(kSimplexVerts, [&](int v) {
    uint32_t c = kSimplexTables.code[v];
    int ipx = (int)(c & 3) - 1, ppx = (int)((c >> 2) & 3);
    int ipy = (int)((c >> 4) & 3) - 1, ppy = (int)((c >> 6) & 3);
    int ipz = (int)((c >> 8) & 3) - 1, ppz = (int)((c >> 10) & 3);
    int i = ipx + ppx, j = ipy + ppy, k = ipz + ppz;
    const double SQ = 1.0 / 3.0;
    double* r = t->vert[v];
    r[0] = (double)ipx; r[1] = (double)ipy; r[2] = (double)ipz;
    r[3] = (double)(i + j + k) * SQ;
    r[4] = (double)ppx; r[5] = (double)ppy; r[6] = (double)ppz;
    uint64_t ijk = (uint64_t)(uint8_t)(int8_t)i | ((uint64_t)(uint8_t)(int8_t)j << 8) | ((uint64_t)(uint8_t)(int8_t)k << 16);
    __builtin_memcpy(&r[7], &ijk, 8);
    r[8] = 0.0;
  });
  each(192, [&](int i) { t->pair[i] = kSimplexTables.pair[i]; });
}

// W supplies assume_lds(): on gfx950 it tells the compiler the tables live in LDS, so the lookups
// become ds_read_* instead of flat loads even though noise3 is a real (non-inlined) function.
template <class W>
struct Simplex {
  const uint8_t* perm;     // [256]
  const uint8_t* pg3;      // [256] gradient number k = perm[i] % 24
  const SimplexLds* tab;   // gradients, extra-vertex tables

  __device__ int gradient_of(int xsv, int ysv, int zsv) const {
    return pg3[(perm[(perm[xsv & 0xFF] + ysv) & 0xFF] + zsv) & 0xFF];
  }
  // One vertex's contribution.  No branch: a wavefront nearly always holds a lane that needs it, and straight-line code lets
  // the ten contributions of a point overlap.  Adding +0.0 for a vertex that does not contribute leaves the sum as it is
  // (the sum starts at +0.0 and a sum of doubles is never -0.0 unless every term is).
  __device__ __forceinline__ void contrib(double& value, bool member, int k, double dx, double dy, double dz) const {
    double attn = 2 - dx * dx - dy * dy - dz * dz;
    const double* g = tab->grad[k];
    bool in = member && attn > 0;
    attn *= attn;
    double term = attn * attn * (g[0] * dx + g[1] * dy + g[2] * dz);
    value += in ? term : 0.0;
  }

  __device__ __attribute__((noinline)) double noise3(double x, double y, double z) const {
    W::assume_lds(perm);
    W::assume_lds(pg3);
    W::assume_lds(tab);
    const double SQ = 1.0 / 3.0;
    const double ST = -1.0 / 6.0;
    double so = (x + y + z) * ST;
    double xs = x + so, ys = y + so, zs = z + so;
    double fx = __builtin_floor(xs), fy = __builtin_floor(ys), fz = __builtin_floor(zs);
    int xsb = (int)fx, ysb = (int)fy, zsb = (int)fz;
    double qo = (double)(xsb + ysb + zsb) * SQ;
    double xb = fx + qo, yb = fy + qo, zb = fz + qo;
    double xins = xs - fx, yins = ys - fy, zins = zs - fz;
    double in_sum = xins + yins + zins;
    double dx0 = x - xb, dy0 = y - yb, dz0 = z - zb;
    bool first = in_sum <= 1, second = in_sum >= 2;

    // the region's six comparisons (simplex_extras), for every lane without a branch.  The second tetrahedron's are the
    // first one's on the negated quantities: a <= b is -a >= -b, and negation is exact.
    double X = second ? -xins : xins, Y = second ? -yins : yins, Z = second ? -zins : zins;
    double wins = (second ? 3.0 : 1.0) - in_sum;
    double Wn = second ? -wins : wins;
    bool q0 = X >= Y, q1 = Z > Y, q2 = Z > X;
    double as = (!q0 && q2) ? Z : X, bs = (q0 && q1) ? Z : Y;
    bool r0 = Wn > as, r1 = Wn > bs, r2 = bs > as;
    int tetra = (int)q0 | ((int)q1 << 1) | ((int)q2 << 2) | ((int)r0 << 3) | ((int)r1 << 4) | ((int)r2 << 5);
    double p1 = xins + yins, p2 = xins + zins, p3 = yins + zins;
    bool b1 = p1 > 1, b2 = p2 > 1, b3 = p3 > 1;
    double oa = b1 ? p1 - 1 : 1 - p1, ob = b2 ? p2 - 1 : 1 - p2, oc = b3 ? p3 - 1 : 1 - p3;
    bool t0 = oa <= ob, t1 = oa < oc, t2 = ob < oc;
    int octa = (int)b1 | ((int)b2 << 1) | ((int)b3 << 2) | ((int)t0 << 3) | ((int)t1 << 4) | ((int)t2 << 5);
    int index = first ? tetra : second ? 64 + tetra : 128 + octa;
    uint32_t extras = tab->pair[index];

    // The regions' fixed vertices are corners of the unit cube: (0,0,0) | (1,0,0) (0,1,0) (0,0,1) | (1,1,0) (1,0,1) (0,1,1) |
    // (1,1,1).  The first tetrahedron sums the first four in this order, the second the last four, the octahedron the
    // middle six -- so ONE pass over the eight corners in this order, each counted only for the lanes of its regions, is
    // every region's published order, and the corner offsets are compile-time constants: (d0 - 1) once per axis, the
    // squish terms literals, the first two permutation levels shared between corners (2 + 4 look-ups instead of 8 + 8).
    // The two extra vertices follow.
    double value = 0.0;
    double bx[2] = {dx0, dx0 - 1.0}, by[2] = {dy0, dy0 - 1.0}, bz[2] = {dz0, dz0 - 1.0};
    int px[2] = {perm[xsb & 0xFF], perm[(xsb + 1) & 0xFF]};
    int pxy[2][2] = {{perm[(px[0] + ysb) & 0xFF], perm[(px[0] + ysb + 1) & 0xFF]},
                     {perm[(px[1] + ysb) & 0xFF], perm[(px[1] + ysb + 1) & 0xFF]}};
    auto corner = [&](bool member, int i, int j, int k) {   // i, j, k: literals
      double sq = (double)(i + j + k) * SQ;
      double dx = bx[i], dy = by[j], dz = bz[k];
      if (i + j + k != 0) {   // (d - 0.0 is d)
        dx = dx - sq;
        dy = dy - sq;
        dz = dz - sq;
      }
      contrib(value, member, pg3[(pxy[i][j] + zsb + k) & 0xFF], dx, dy, dz);
    };
    corner(first, 0, 0, 0);
    corner(!second, 1, 0, 0);
    corner(!second, 0, 1, 0);
    corner(!second, 0, 0, 1);
    corner(!first, 1, 1, 0);
    corner(!first, 1, 0, 1);
    corner(!first, 0, 1, 1);
    corner(second, 1, 1, 1);
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const double* r = tab->vert[(extras >> (8 * s)) & 0xFF];
      double dx = ((dx0 - r[0]) - r[3]) - r[4];
      double dy = ((dy0 - r[1]) - r[3]) - r[5];
      double dz = ((dz0 - r[2]) - r[3]) - r[6];
      uint32_t ijk = ((const uint32_t*)r)[14];
      int i = (int)(int8_t)(ijk & 0xFF), j = (int)(int8_t)((ijk >> 8) & 0xFF), k = (int)(int8_t)((ijk >> 16) & 0xFF);
      contrib(value, true, gradient_of(xsb + i, ysb + j, zsb + k), dx, dy, dz);
    }
    return value / 103.0;
  }
};

// r_i of the OpenSimplex seeding shuffle for every i at once: the LCG is advanced
// (3 + (256 - i)) times from the seed, r_i = (state + 31) mod (i + 1), non-negative.
// state is a wrapped int64; the package adds 31 WITHOUT wrapping (Python int), hence the split mod.
// (k steps of the generator at once: s_k = a_k s_0 + c_k mod 2^64 with a_k = M^k, c_k = c (M^(k-1) + ... + 1) -- constants.
// Stepping the generator up to 259 times per index, four indices per lane, was 46 k of the seeding kernel's 172 k clocks: round 6.)
struct LcgJump {
  uint64_t a[260], c[260];
};
constexpr LcgJump make_lcg_jump() {
  LcgJump t{};
  uint64_t a = 1, c = 0;
  for (int k = 0; k < 260; k++) {
    t.a[k] = a;
    t.c[k] = c;
    a = a * 6364136223846793005ull;
    c = c * 6364136223846793005ull + 1442695040888963407ull;
  }
  return t;
}
__device__ const LcgJump kLcgJump = make_lcg_jump();
__device__ inline int simplex_shuffle_index(int64_t seed, int i) {
  int steps = 3 + (256 - i);
  uint64_t s = kLcgJump.a[steps] * (uint64_t)seed + kLcgJump.c[steps];
  int64_t n = i + 1;
  int64_t r = (int64_t)s % n;
  if (r < 0) r += n;
  r = (r + (31 % n)) % n;
  return (int)r;
}

}  // namespace crafter
