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

// W supplies assume_lds(): on gfx950 it tells the compiler the tables live in LDS, so the lookups
// become ds_read_u8 instead of flat loads even though noise3 is a real (non-inlined) function.
template <class W>
struct Simplex {
  const uint8_t* perm;   // [256]
  const uint8_t* pg3;    // [256] gradient number k = perm[i] % 24
  const uint4* grad;     // [24]  gradient k as the HIGH dwords of its three binary64 components (the low dwords are 0), .w unused

  // The 24 gradients are the sign/axis permutations of (11, 4, 4) in this order:
  //   k = 3 * q + a:  axis a carries the 11;  x is negative unless q & 1;  y negative if q & 2;
  //   z negative if q & 4.  11.0 = 0x4026000000000000 and 4.0 = 0x4010000000000000: a component is its high dword.
  //   One 16-byte LDS read per vertex instead of ~22 compare / select / sign instructions.
  __host__ __device__ static uint4 gradient_entry(int k) {
    int a = k % 3, q = k / 3;
    uint4 g;
    g.x = ((a == 0) ? 0x40260000u : 0x40100000u) | ((q & 1) ? 0u : 0x80000000u);
    g.y = ((a == 1) ? 0x40260000u : 0x40100000u) | ((q & 2) ? 0x80000000u : 0u);
    g.z = ((a == 2) ? 0x40260000u : 0x40100000u) | ((q & 4) ? 0x80000000u : 0u);
    g.w = 0;
    return g;
  }
  static constexpr int kGradBytes = 24 * 16;
  __device__ static double from_high(uint32_t hi) {
    uint64_t u = (uint64_t)hi << 32;
    double d;
    __builtin_memcpy(&d, &u, 8);
    return d;
  }
  __device__ int gradient_of(int xsv, int ysv, int zsv) const {
    return pg3[(perm[(perm[xsv & 0xFF] + ysv) & 0xFF] + zsv) & 0xFF];
  }
  __device__ void contrib(double& value, int k, double dx, double dy, double dz) const {
    double attn = 2 - dx * dx - dy * dy - dz * dz;
    if (attn > 0) {
      uint4 g = grad[k];
      attn *= attn;
      value += attn * attn * (from_high(g.x) * dx + from_high(g.y) * dy + from_high(g.z) * dz);
    }
  }

  // Every displacement the published code writes down has the form ((d0 - i_pre) - s * SQUISH) - post along each axis:
  // i_pre in -1..2 is the vertex's lattice offset as it appears inside the expression (d0 + 1 is d0 - (-1)), s the sum of
  // the vertex's three offsets (s * SQUISH: 0, SQ, 2 * SQ, 3 * SQ -- exactly the constants of the source, the products by
  // 0, 1, 2 are exact and 3 * SQ is the same rounded product), and post is non-zero only for the two places where the
  // source subtracts AFTER the squish term (`dy_ext -= 1` in the second tetrahedron, `dx_ext1 -= 2` in the octahedron);
  // subtracting 0.0 changes nothing.  So the three regions of the simplectic honeycomb -- which lanes of one wavefront
  // enter independently -- only decide small integers: which lattice vertices contribute, and (i_pre, post) per axis.
  // The regions' own vertices are corners of the unit cube and are summed by one predicated pass over the eight corners
  // (see noise3); only the two "extra" vertices of a region are data: code = 4 bits per axis, (i_pre + 1) | post << 2.
  __device__ __forceinline__ static uint32_t F(int i_pre, int post = 0) { return (uint32_t)(i_pre + 1) | ((uint32_t)post << 2); }
  __device__ __forceinline__ static uint32_t V3(int i, int j, int k) { return F(i) | (F(j) << 4) | (F(k) << 8); }

  __device__ __attribute__((noinline)) double noise3(double x, double y, double z) const {
    W::assume_lds(perm);
    W::assume_lds(pg3);
    W::assume_lds(grad);
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
    uint32_t e0, e1;   // the two "extra" vertices

    if (in_sum <= 1) {  // tetrahedron at (0,0,0)
      int ap = 1, bp = 2;
      double as = xins, bs = yins;
      if (as >= bs && zins > bs) {
        bs = zins;
        bp = 4;
      } else if (as < bs && zins > as) {
        as = zins;
        ap = 4;
      }
      double wins = 1 - in_sum;
      if (wins > as || wins > bs) {
        int c = (bs > as) ? bp : ap;
        int xe0, xe1, ye0, ye1, ze0, ze1;
        if ((c & 1) == 0) { xe0 = -1; xe1 = 0; } else { xe0 = xe1 = 1; }
        if ((c & 2) == 0) {
          ye0 = ye1 = 0;
          if ((c & 1) == 0) ye1 = -1; else ye0 = -1;
        } else {
          ye0 = ye1 = 1;
        }
        if ((c & 4) == 0) { ze0 = 0; ze1 = -1; } else { ze0 = ze1 = 1; }
        e0 = V3(xe0, ye0, ze0);
        e1 = V3(xe1, ye1, ze1);
      } else {
        int c = ap | bp;
        e0 = V3((c & 1) ? 1 : 0, (c & 2) ? 1 : 0, (c & 4) ? 1 : 0);
        e1 = V3((c & 1) ? 1 : -1, (c & 2) ? 1 : -1, (c & 4) ? 1 : -1);
      }
    } else if (in_sum >= 2) {  // tetrahedron at (1,1,1)
      int ap = 6, bp = 5;
      double as = xins, bs = yins;
      if (as <= bs && zins < bs) {
        bs = zins;
        bp = 3;
      } else if (as > bs && zins < as) {
        as = zins;
        ap = 3;
      }
      double wins = 3 - in_sum;
      if (wins < as || wins < bs) {
        int c = (bs < as) ? bp : ap;
        uint32_t x0 = (c & 1) ? F(2) : F(0), x1 = (c & 1) ? F(1) : F(0);
        uint32_t y0, y1;
        if (c & 2) {
          y0 = y1 = F(1);
          if (c & 1) y1 = F(1, 1); else y0 = F(1, 1);   // dy_ext = dy0 - 1 - 3 * SQ, then `dy_ext -= 1`
        } else {
          y0 = y1 = F(0);
        }
        uint32_t z0 = (c & 4) ? F(1) : F(0), z1 = (c & 4) ? F(2) : F(0);
        e0 = x0 | (y0 << 4) | (z0 << 8);
        e1 = x1 | (y1 << 4) | (z1 << 8);
      } else {
        int c = ap & bp;
        e0 = V3((c & 1) ? 1 : 0, (c & 2) ? 1 : 0, (c & 4) ? 1 : 0);
        e1 = V3((c & 1) ? 2 : 0, (c & 2) ? 2 : 0, (c & 4) ? 2 : 0);
      }
    } else {  // octahedron in between
      double as, bs;
      int ap, bp;
      bool af, bf;
      double p1 = xins + yins;
      if (p1 > 1) { as = p1 - 1; ap = 3; af = true; } else { as = 1 - p1; ap = 4; af = false; }
      double p2 = xins + zins;
      if (p2 > 1) { bs = p2 - 1; bp = 5; bf = true; } else { bs = 1 - p2; bp = 2; bf = false; }
      double p3 = yins + zins;
      if (p3 > 1) {
        double sc = p3 - 1;
        if (as <= bs && as < sc) { as = sc; ap = 6; af = true; }
        else if (as > bs && bs < sc) { bs = sc; bp = 6; bf = true; }
      } else {
        double sc = 1 - p3;
        if (as <= bs && as < sc) { as = sc; ap = 1; af = false; }
        else if (as > bs && bs < sc) { bs = sc; bp = 1; bf = false; }
      }
      if (af == bf) {
        if (af) {  // both closest points on the (1,1,1) side
          e0 = V3(1, 1, 1);
          int c = ap & bp;
          e1 = (c & 1) ? V3(2, 0, 0) : (c & 2) ? V3(0, 2, 0) : V3(0, 0, 2);
        } else {  // both on the (0,0,0) side
          e0 = V3(0, 0, 0);
          int c = ap | bp;
          e1 = ((c & 1) == 0) ? V3(-1, 1, 1) : ((c & 2) == 0) ? V3(1, -1, 1) : V3(1, 1, -1);
        }
      } else {  // one point on each side
        int c1 = af ? ap : bp, c2 = af ? bp : ap;
        e0 = ((c1 & 1) == 0) ? V3(-1, 1, 1) : ((c1 & 2) == 0) ? V3(1, -1, 1) : V3(1, 1, -1);
        // dx_ext1 = dx0 - 2 * SQ on every axis, then `-= 2` on one of them
        e1 = (c2 & 1) ? (F(0, 2) | (F(0) << 4) | (F(0) << 8)) : (c2 & 2) ? (F(0) | (F(0, 2) << 4) | (F(0) << 8)) : (F(0) | (F(0) << 4) | (F(0, 2) << 8));
      }
    }
    // The regions' fixed vertices are corners of the unit cube: (0,0,0) | (1,0,0) (0,1,0) (0,0,1) | (1,1,0) (1,0,1) (0,1,1) |
    // (1,1,1).  The first tetrahedron sums the first four in this order, the second the last four, the octahedron the
    // middle six -- so ONE pass over the eight corners in this order, each predicated on the lane's region, is every
    // region's published order (a skipped contribution adds nothing), and the corner offsets are compile-time constants:
    // (d0 - 1) once per axis, the squish terms literals, the first two permutation levels shared between corners (2 + 4
    // look-ups instead of 8 + 8).  The two extra vertices follow, decoded from their codes.
    bool first = in_sum <= 1, second = in_sum >= 2;
    double value = 0.0;
    double bx[2] = {dx0, dx0 - 1.0}, by[2] = {dy0, dy0 - 1.0}, bz[2] = {dz0, dz0 - 1.0};
    int px[2] = {perm[xsb & 0xFF], perm[(xsb + 1) & 0xFF]};
    int pxy[2][2] = {{perm[(px[0] + ysb) & 0xFF], perm[(px[0] + ysb + 1) & 0xFF]},
                     {perm[(px[1] + ysb) & 0xFF], perm[(px[1] + ysb + 1) & 0xFF]}};
    auto corner = [&](bool member, int i, int j, int k) {   // i, j, k: literals
      if (!member) return;
      double sq = (double)(i + j + k) * SQ;
      double dx = bx[i], dy = by[j], dz = bz[k];
      if (i + j + k != 0) {   // (d - 0.0 is d)
        dx = dx - sq;
        dy = dy - sq;
        dz = dz - sq;
      }
      contrib(value, pg3[(pxy[i][j] + zsb + k) & 0xFF], dx, dy, dz);
    };
    corner(first, 0, 0, 0);
    corner(!second, 1, 0, 0);
    corner(!second, 0, 1, 0);
    corner(!second, 0, 0, 1);
    corner(!first, 1, 1, 0);
    corner(!first, 1, 0, 1);
    corner(!first, 0, 1, 1);
    corner(second, 1, 1, 1);
    uint32_t extra[2] = {e0, e1};
#pragma unroll
    for (int s = 0; s < 2; s++) {
      uint32_t c = extra[s];
      int ipx = (int)(c & 3) - 1, ppx = (int)((c >> 2) & 3);
      int ipy = (int)((c >> 4) & 3) - 1, ppy = (int)((c >> 6) & 3);
      int ipz = (int)((c >> 8) & 3) - 1, ppz = (int)((c >> 10) & 3);
      int i = ipx + ppx, j = ipy + ppy, k = ipz + ppz;
      double sq = (double)(i + j + k) * SQ;
      double dx = ((dx0 - (double)ipx) - sq) - (double)ppx;
      double dy = ((dy0 - (double)ipy) - sq) - (double)ppy;
      double dz = ((dz0 - (double)ipz) - sq) - (double)ppz;
      contrib(value, gradient_of(xsb + i, ysb + j, zsb + k), dx, dy, dz);
    }
    return value / 103.0;
  }
};

// r_i of the OpenSimplex seeding shuffle for every i at once: the LCG is advanced
// (3 + (256 - i)) times from the seed, r_i = (state + 31) mod (i + 1), non-negative.
// state is a wrapped int64; the package adds 31 WITHOUT wrapping (Python int), hence the split mod.
__device__ inline int simplex_shuffle_index(int64_t seed, int i) {
  uint64_t s = (uint64_t)seed;
  int steps = 3 + (256 - i);
  for (int k = 0; k < steps; k++) s = s * 6364136223846793005ull + 1442695040888963407ull;
  int64_t n = i + 1;
  int64_t r = (int64_t)s % n;
  if (r < 0) r += n;
  r = (r + (31 % n)) % n;
  return (int)r;
}

}  // namespace crafter
