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
  const uint8_t* pg3;    // [256] gradient number k = perm[i] % 24, packed (k % 3) | (k / 3) << 2

  // The 24 gradients are the sign/axis permutations of (11, 4, 4) in this order:
  //   k = 3 * q + a:  axis a carries the 11;  x is negative unless q & 1;  y negative if q & 2;
  //   z negative if q & 4.  Computed, not looked up: a table would sit in global memory and cost
  //   three dependent loads per lattice vertex.
  // pg3[] holds the gradient number k = perm % 24 packed as (k % 3) | (k / 3) << 2, so that axis and sign bits
  // come out with a mask and a shift.
  __device__ int gradient_of(int xsv, int ysv, int zsv) const {
    return pg3[(perm[(perm[xsv & 0xFF] + ysv) & 0xFF] + zsv) & 0xFF];
  }
  __device__ static void contrib(double& value, int g, double dx, double dy, double dz) {
    double attn = 2 - dx * dx - dy * dy - dz * dz;
    if (attn > 0) {
      int a = g & 3, q = g >> 2;
      double gx = (a == 0) ? 11.0 : 4.0, gy = (a == 1) ? 11.0 : 4.0, gz = (a == 2) ? 11.0 : 4.0;
      if (!(q & 1)) gx = -gx;
      if (q & 2) gy = -gy;
      if (q & 4) gz = -gz;
      attn *= attn;
      value += attn * attn * (gx * dx + gy * dy + gz * dz);
    }
  }

  // Every displacement the published code writes down has the form ((d0 - i_pre) - s * SQUISH) - post along each axis:
  // i_pre in -1..2 is the vertex's lattice offset as it appears inside the expression (d0 + 1 is d0 - (-1)), s the sum of
  // the vertex's three offsets (s * SQUISH: 0, SQ, 2 * SQ, 3 * SQ -- exactly the constants of the source, the products by
  // 0, 1, 2 are exact and 3 * SQ is the same rounded product), and post is non-zero only for the two places where the
  // source subtracts AFTER the squish term (`dy_ext -= 1` in the second tetrahedron, `dx_ext1 -= 2` in the octahedron);
  // subtracting 0.0 changes nothing.  So the three regions of the simplectic honeycomb -- which lanes of one wavefront
  // enter independently -- only decide small integers: which lattice vertices contribute, and (i_pre, post) per axis.
  // The binary64 work (displacements, the three dependent permutation look-ups, the contributions) is one loop over
  // eight vertex slots that every lane runs in the published summation order (the order of the additions matters for
  // the last bits).  Vertex code: 4 bits per axis, (i_pre + 1) | post << 2; kSkip = empty slot.
  static constexpr uint32_t kSkip = 0xFFFFu;
  __device__ __forceinline__ static uint32_t F(int i_pre, int post = 0) { return (uint32_t)(i_pre + 1) | ((uint32_t)post << 2); }
  __device__ __forceinline__ static uint32_t V3(int i, int j, int k) { return F(i) | (F(j) << 4) | (F(k) << 8); }

  __device__ __attribute__((noinline)) double noise3(double x, double y, double z) const {
    W::assume_lds(perm);
    W::assume_lds(pg3);
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
    uint32_t code[8];
    code[6] = kSkip;
    code[7] = kSkip;
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
      code[0] = V3(0, 0, 0);
      code[1] = V3(1, 0, 0);
      code[2] = V3(0, 1, 0);
      code[3] = V3(0, 0, 1);
      code[4] = e0;
      code[5] = e1;
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
      code[0] = V3(1, 1, 0);
      code[1] = V3(1, 0, 1);
      code[2] = V3(0, 1, 1);
      code[3] = V3(1, 1, 1);
      code[4] = e0;
      code[5] = e1;
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
      code[0] = V3(1, 0, 0);
      code[1] = V3(0, 1, 0);
      code[2] = V3(0, 0, 1);
      code[3] = V3(1, 1, 0);
      code[4] = V3(1, 0, 1);
      code[5] = V3(0, 1, 1);
      code[6] = e0;
      code[7] = e1;
    }
    double value = 0.0;
    // slots 0..3 are the region's fixed vertices: offsets 0 / 1, nothing subtracted after the squish term
#pragma unroll
    for (int s = 0; s < 4; s++) {
      uint32_t c = code[s];
      int i = (int)(c & 3) - 1, j = (int)((c >> 4) & 3) - 1, k = (int)((c >> 8) & 3) - 1;
      double sq = (double)(i + j + k) * SQ;
      double dx = (dx0 - (double)i) - sq;
      double dy = (dy0 - (double)j) - sq;
      double dz = (dz0 - (double)k) - sq;
      contrib(value, gradient_of(xsb + i, ysb + j, zsb + k), dx, dy, dz);
    }
#pragma unroll
    for (int s = 4; s < 8; s++) {
      uint32_t c = code[s];
      if (c == kSkip) continue;
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
