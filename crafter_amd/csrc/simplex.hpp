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

  // Lattice vertex (relative to the cell origin) and the displacement of the sample from it.
  struct Vtx {
    int ijk;   // (i + 1) | (j + 1) << 2 | (k + 1) << 4, each offset in -1..2
    double dx, dy, dz;
  };
  __device__ static Vtx V(int i, int j, int k, double dx, double dy, double dz) {
    Vtx v;
    v.ijk = (i + 1) | ((j + 1) << 2) | ((k + 1) << 4);
    v.dx = dx; v.dy = dy; v.dz = dz;
    return v;
  }

  // The three regions of the simplectic honeycomb (two tetrahedra and the octahedron between them)
  // only differ in WHICH lattice vertices contribute; the contribution itself is the same code.  Each
  // region therefore just fills a list of up to 8 vertices, in the published summation order (the
  // order of the additions matters for the last bits), and one shared loop evaluates them: a
  // wavefront whose lanes fall into different regions runs the expensive part once, not three times.
  // Every displacement expression keeps the association order of the published code.
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
    int xe0, ye0, ze0, xe1, ye1, ze1;   // the two "extra" vertices, relative to (xsb, ysb, zsb)
    double dxe0, dye0, dze0, dxe1, dye1, dze1;
    const double FAR = 4.0;             // unused slot: attn = 2 - 48 < 0, contributes nothing
    Vtx v[8];
    v[6] = V(0, 0, 0, FAR, FAR, FAR);
    v[7] = V(0, 0, 0, FAR, FAR, FAR);

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
        if ((c & 1) == 0) {
          xe0 = -1; xe1 = 0; dxe0 = dx0 + 1; dxe1 = dx0;
        } else {
          xe0 = xe1 = 1; dxe0 = dxe1 = dx0 - 1;
        }
        if ((c & 2) == 0) {
          ye0 = ye1 = 0; dye0 = dye1 = dy0;
          if ((c & 1) == 0) { ye1 -= 1; dye1 += 1; } else { ye0 -= 1; dye0 += 1; }
        } else {
          ye0 = ye1 = 1; dye0 = dye1 = dy0 - 1;
        }
        if ((c & 4) == 0) {
          ze0 = 0; ze1 = -1; dze0 = dz0; dze1 = dz0 + 1;
        } else {
          ze0 = ze1 = 1; dze0 = dze1 = dz0 - 1;
        }
      } else {
        int c = ap | bp;
        if ((c & 1) == 0) { xe0 = 0; xe1 = -1; dxe0 = dx0 - 2 * SQ; dxe1 = dx0 + 1 - SQ; }
        else { xe0 = xe1 = 1; dxe0 = dx0 - 1 - 2 * SQ; dxe1 = dx0 - 1 - SQ; }
        if ((c & 2) == 0) { ye0 = 0; ye1 = -1; dye0 = dy0 - 2 * SQ; dye1 = dy0 + 1 - SQ; }
        else { ye0 = ye1 = 1; dye0 = dy0 - 1 - 2 * SQ; dye1 = dy0 - 1 - SQ; }
        if ((c & 4) == 0) { ze0 = 0; ze1 = -1; dze0 = dz0 - 2 * SQ; dze1 = dz0 + 1 - SQ; }
        else { ze0 = ze1 = 1; dze0 = dz0 - 1 - 2 * SQ; dze1 = dz0 - 1 - SQ; }
      }
      double dx1 = dx0 - 1 - SQ, dy1 = dy0 - 0 - SQ, dz1 = dz0 - 0 - SQ;
      double dx2 = dx0 - 0 - SQ, dy2 = dy0 - 1 - SQ, dz2 = dz1;
      double dx3 = dx2, dy3 = dy1, dz3 = dz0 - 1 - SQ;
      v[0] = V(0, 0, 0, dx0, dy0, dz0);
      v[1] = V(1, 0, 0, dx1, dy1, dz1);
      v[2] = V(0, 1, 0, dx2, dy2, dz2);
      v[3] = V(0, 0, 1, dx3, dy3, dz3);
      v[4] = V(xe0, ye0, ze0, dxe0, dye0, dze0);
      v[5] = V(xe1, ye1, ze1, dxe1, dye1, dze1);
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
        if ((c & 1) != 0) { xe0 = 2; xe1 = 1; dxe0 = dx0 - 2 - 3 * SQ; dxe1 = dx0 - 1 - 3 * SQ; }
        else { xe0 = xe1 = 0; dxe0 = dxe1 = dx0 - 3 * SQ; }
        if ((c & 2) != 0) {
          ye0 = ye1 = 1; dye0 = dye1 = dy0 - 1 - 3 * SQ;
          if ((c & 1) != 0) { ye1 += 1; dye1 -= 1; } else { ye0 += 1; dye0 -= 1; }
        } else {
          ye0 = ye1 = 0; dye0 = dye1 = dy0 - 3 * SQ;
        }
        if ((c & 4) != 0) { ze0 = 1; ze1 = 2; dze0 = dz0 - 1 - 3 * SQ; dze1 = dz0 - 2 - 3 * SQ; }
        else { ze0 = ze1 = 0; dze0 = dze1 = dz0 - 3 * SQ; }
      } else {
        int c = ap & bp;
        if ((c & 1) != 0) { xe0 = 1; xe1 = 2; dxe0 = dx0 - 1 - SQ; dxe1 = dx0 - 2 - 2 * SQ; }
        else { xe0 = xe1 = 0; dxe0 = dx0 - SQ; dxe1 = dx0 - 2 * SQ; }
        if ((c & 2) != 0) { ye0 = 1; ye1 = 2; dye0 = dy0 - 1 - SQ; dye1 = dy0 - 2 - 2 * SQ; }
        else { ye0 = ye1 = 0; dye0 = dy0 - SQ; dye1 = dy0 - 2 * SQ; }
        if ((c & 4) != 0) { ze0 = 1; ze1 = 2; dze0 = dz0 - 1 - SQ; dze1 = dz0 - 2 - 2 * SQ; }
        else { ze0 = ze1 = 0; dze0 = dz0 - SQ; dze1 = dz0 - 2 * SQ; }
      }
      double dx3 = dx0 - 1 - 2 * SQ, dy3 = dy0 - 1 - 2 * SQ, dz3 = dz0 - 0 - 2 * SQ;
      double dx2 = dx3, dy2 = dy0 - 0 - 2 * SQ, dz2 = dz0 - 1 - 2 * SQ;
      double dx1 = dx0 - 0 - 2 * SQ, dy1 = dy3, dz1 = dz2;
      v[0] = V(1, 1, 0, dx3, dy3, dz3);
      v[1] = V(1, 0, 1, dx2, dy2, dz2);
      v[2] = V(0, 1, 1, dx1, dy1, dz1);
      v[3] = V(1, 1, 1, dx0 - 1 - 3 * SQ, dy0 - 1 - 3 * SQ, dz0 - 1 - 3 * SQ);
      v[4] = V(xe0, ye0, ze0, dxe0, dye0, dze0);
      v[5] = V(xe1, ye1, ze1, dxe1, dye1, dze1);
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
          dxe0 = dx0 - 1 - 3 * SQ; dye0 = dy0 - 1 - 3 * SQ; dze0 = dz0 - 1 - 3 * SQ;
          xe0 = 1; ye0 = 1; ze0 = 1;
          int c = ap & bp;
          if ((c & 1) != 0) {
            dxe1 = dx0 - 2 - 2 * SQ; dye1 = dy0 - 2 * SQ; dze1 = dz0 - 2 * SQ;
            xe1 = 2; ye1 = 0; ze1 = 0;
          } else if ((c & 2) != 0) {
            dxe1 = dx0 - 2 * SQ; dye1 = dy0 - 2 - 2 * SQ; dze1 = dz0 - 2 * SQ;
            xe1 = 0; ye1 = 2; ze1 = 0;
          } else {
            dxe1 = dx0 - 2 * SQ; dye1 = dy0 - 2 * SQ; dze1 = dz0 - 2 - 2 * SQ;
            xe1 = 0; ye1 = 0; ze1 = 2;
          }
        } else {  // both on the (0,0,0) side
          dxe0 = dx0; dye0 = dy0; dze0 = dz0;
          xe0 = 0; ye0 = 0; ze0 = 0;
          int c = ap | bp;
          if ((c & 1) == 0) {
            dxe1 = dx0 + 1 - SQ; dye1 = dy0 - 1 - SQ; dze1 = dz0 - 1 - SQ;
            xe1 = -1; ye1 = 1; ze1 = 1;
          } else if ((c & 2) == 0) {
            dxe1 = dx0 - 1 - SQ; dye1 = dy0 + 1 - SQ; dze1 = dz0 - 1 - SQ;
            xe1 = 1; ye1 = -1; ze1 = 1;
          } else {
            dxe1 = dx0 - 1 - SQ; dye1 = dy0 - 1 - SQ; dze1 = dz0 + 1 - SQ;
            xe1 = 1; ye1 = 1; ze1 = -1;
          }
        }
      } else {  // one point on each side
        int c1 = af ? ap : bp, c2 = af ? bp : ap;
        if ((c1 & 1) == 0) {
          dxe0 = dx0 + 1 - SQ; dye0 = dy0 - 1 - SQ; dze0 = dz0 - 1 - SQ;
          xe0 = -1; ye0 = 1; ze0 = 1;
        } else if ((c1 & 2) == 0) {
          dxe0 = dx0 - 1 - SQ; dye0 = dy0 + 1 - SQ; dze0 = dz0 - 1 - SQ;
          xe0 = 1; ye0 = -1; ze0 = 1;
        } else {
          dxe0 = dx0 - 1 - SQ; dye0 = dy0 - 1 - SQ; dze0 = dz0 + 1 - SQ;
          xe0 = 1; ye0 = 1; ze0 = -1;
        }
        dxe1 = dx0 - 2 * SQ; dye1 = dy0 - 2 * SQ; dze1 = dz0 - 2 * SQ;
        xe1 = 0; ye1 = 0; ze1 = 0;
        if ((c2 & 1) != 0) { dxe1 -= 2; xe1 += 2; }
        else if ((c2 & 2) != 0) { dye1 -= 2; ye1 += 2; }
        else { dze1 -= 2; ze1 += 2; }
      }
      double dx1 = dx0 - 1 - SQ, dy1 = dy0 - 0 - SQ, dz1 = dz0 - 0 - SQ;
      double dx2 = dx0 - 0 - SQ, dy2 = dy0 - 1 - SQ, dz2 = dz1;
      double dx3 = dx2, dy3 = dy1, dz3 = dz0 - 1 - SQ;
      double dx4 = dx0 - 1 - 2 * SQ, dy4 = dy0 - 1 - 2 * SQ, dz4 = dz0 - 0 - 2 * SQ;
      double dx5 = dx4, dy5 = dy0 - 0 - 2 * SQ, dz5 = dz0 - 1 - 2 * SQ;
      double dx6 = dx0 - 0 - 2 * SQ, dy6 = dy4, dz6 = dz5;
      v[0] = V(1, 0, 0, dx1, dy1, dz1);
      v[1] = V(0, 1, 0, dx2, dy2, dz2);
      v[2] = V(0, 0, 1, dx3, dy3, dz3);
      v[3] = V(1, 1, 0, dx4, dy4, dz4);
      v[4] = V(1, 0, 1, dx5, dy5, dz5);
      v[5] = V(0, 1, 1, dx6, dy6, dz6);
      v[6] = V(xe0, ye0, ze0, dxe0, dye0, dze0);
      v[7] = V(xe1, ye1, ze1, dxe1, dye1, dze1);
    }
    // The permutation look-ups first, for all eight vertices at once: eight independent chains of three dependent
    // LDS reads (inside the attn > 0 branch they would be issued, and waited for, one vertex after the other).
    int g[8];
#pragma unroll
    for (int s = 0; s < 8; s++)
      g[s] = gradient_of(xsb + (v[s].ijk & 3) - 1, ysb + ((v[s].ijk >> 2) & 3) - 1, zsb + ((v[s].ijk >> 4) & 3) - 1);
    double value = 0.0;
#pragma unroll
    for (int s = 0; s < 8; s++) contrib(value, g[s], v[s].dx, v[s].dy, v[s].dz);
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
