/* ORACLE (test infrastructure, never linked into the product library).
 *
 * C restatement of opensimplex.OpenSimplex(seed).noise3 -- the un-vendored 3rd-party
 * dependency the reference calls at /root/reference/crafter/worldgen.py:11,84-87.
 * Algorithm: K. Spencer, "OpenSimplex Noise in Java" (2014, public domain), as specified in
 * SURVEY.md App. B.  Same maths as oracle/opensimplex_ref.py, kept in C only so that oracle
 * resets (26 k evaluations per 64x64 world) take milliseconds instead of 0.25 s.
 *
 * PARITY STATUS: seeding pinned by the 2-D upstream known answers (through the Python twin);
 * the 3-D evaluation is "parity unpinned" -- no output of the real package is available.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  Every expression is
 * written in the association order of the published code; no FMA contraction.
 */
#include <math.h>
#include <stdint.h>

static const double SQ = 1.0 / 3.0;
static const double ST = -1.0 / 6.0;

static const int8_t G3[72] = {
    -11, 4, 4, -4, 11, 4, -4, 4, 11, 11, 4, 4, 4, 11, 4, 4, 4, 11,
    -11, -4, 4, -4, -11, 4, -4, -4, 11, 11, -4, 4, 4, -11, 4, 4, -4, 11,
    -11, 4, -4, -4, 11, -4, -4, 4, -11, 11, 4, -4, 4, 11, -4, 4, 4, -11,
    -11, -4, -4, -4, -11, -4, -4, -4, -11, 11, -4, -4, 4, -11, -4, 4, -4, -11};

/* perm / pg3: 256 int16 each. */
void osn_make_perm(int64_t seed, int16_t* perm, int16_t* pg3) {
  int16_t source[256];
  uint64_t s = (uint64_t)seed;
  for (int i = 0; i < 256; i++) source[i] = (int16_t)i;
  for (int k = 0; k < 3; k++) s = s * 6364136223846793005ULL + 1442695040888963407ULL;
  for (int i = 255; i >= 0; i--) {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    /* (seed + 31) mod (i + 1) as a NON-NEGATIVE remainder of the signed value; the Python
       package adds 31 without wrapping, so widen to 128 bits. */
    __int128 v = (__int128)(int64_t)s + 31;
    int64_t r = (int64_t)(v % (i + 1));
    if (r < 0) r += i + 1;
    perm[i] = source[r];
    pg3[i] = (int16_t)((perm[i] % 24) * 3);
    source[r] = source[i];
  }
}

typedef struct {
  const int16_t* perm;
  const int16_t* pg3;
  double value;
} Ctx;

static inline void contrib(Ctx* c, int64_t xsv, int64_t ysv, int64_t zsv, double dx, double dy, double dz) {
  double attn = 2 - dx * dx - dy * dy - dz * dz;
  if (attn > 0) {
    int g = c->pg3[(c->perm[(c->perm[xsv & 0xFF] + ysv) & 0xFF] + zsv) & 0xFF];
    attn *= attn;
    c->value += attn * attn * (G3[g] * dx + G3[g + 1] * dy + G3[g + 2] * dz);
  }
}

double osn_noise3(const int16_t* perm, const int16_t* pg3, double x, double y, double z) {
  Ctx c = {perm, pg3, 0.0};
  double so = (x + y + z) * ST;
  double xs = x + so, ys = y + so, zs = z + so;
  int64_t xsb = (int64_t)floor(xs), ysb = (int64_t)floor(ys), zsb = (int64_t)floor(zs);
  double qo = (double)(xsb + ysb + zsb) * SQ;
  double xb = xsb + qo, yb = ysb + qo, zb = zsb + qo;
  double xins = xs - xsb, yins = ys - ysb, zins = zs - zsb;
  double in_sum = xins + yins + zins;
  double dx0 = x - xb, dy0 = y - yb, dz0 = z - zb;
  int64_t xe0, ye0, ze0, xe1, ye1, ze1;
  double dxe0, dye0, dze0, dxe1, dye1, dze1;

  if (in_sum <= 1) {
    int ap = 1, bp = 2;
    double as = xins, bs = yins;
    if (as >= bs && zins > bs) { bs = zins; bp = 4; }
    else if (as < bs && zins > as) { as = zins; ap = 4; }
    double wins = 1 - in_sum;
    if (wins > as || wins > bs) {
      int cc = (bs > as) ? bp : ap;
      if ((cc & 1) == 0) { xe0 = xsb - 1; xe1 = xsb; dxe0 = dx0 + 1; dxe1 = dx0; }
      else { xe0 = xe1 = xsb + 1; dxe0 = dxe1 = dx0 - 1; }
      if ((cc & 2) == 0) {
        ye0 = ye1 = ysb; dye0 = dye1 = dy0;
        if ((cc & 1) == 0) { ye1 -= 1; dye1 += 1; } else { ye0 -= 1; dye0 += 1; }
      } else { ye0 = ye1 = ysb + 1; dye0 = dye1 = dy0 - 1; }
      if ((cc & 4) == 0) { ze0 = zsb; ze1 = zsb - 1; dze0 = dz0; dze1 = dz0 + 1; }
      else { ze0 = ze1 = zsb + 1; dze0 = dze1 = dz0 - 1; }
    } else {
      int cc = ap | bp;
      if ((cc & 1) == 0) { xe0 = xsb; xe1 = xsb - 1; dxe0 = dx0 - 2 * SQ; dxe1 = dx0 + 1 - SQ; }
      else { xe0 = xe1 = xsb + 1; dxe0 = dx0 - 1 - 2 * SQ; dxe1 = dx0 - 1 - SQ; }
      if ((cc & 2) == 0) { ye0 = ysb; ye1 = ysb - 1; dye0 = dy0 - 2 * SQ; dye1 = dy0 + 1 - SQ; }
      else { ye0 = ye1 = ysb + 1; dye0 = dy0 - 1 - 2 * SQ; dye1 = dy0 - 1 - SQ; }
      if ((cc & 4) == 0) { ze0 = zsb; ze1 = zsb - 1; dze0 = dz0 - 2 * SQ; dze1 = dz0 + 1 - SQ; }
      else { ze0 = ze1 = zsb + 1; dze0 = dz0 - 1 - 2 * SQ; dze1 = dz0 - 1 - SQ; }
    }
    contrib(&c, xsb, ysb, zsb, dx0, dy0, dz0);
    double dx1 = dx0 - 1 - SQ, dy1 = dy0 - 0 - SQ, dz1 = dz0 - 0 - SQ;
    contrib(&c, xsb + 1, ysb, zsb, dx1, dy1, dz1);
    double dx2 = dx0 - 0 - SQ, dy2 = dy0 - 1 - SQ, dz2 = dz1;
    contrib(&c, xsb, ysb + 1, zsb, dx2, dy2, dz2);
    double dx3 = dx2, dy3 = dy1, dz3 = dz0 - 1 - SQ;
    contrib(&c, xsb, ysb, zsb + 1, dx3, dy3, dz3);
  } else if (in_sum >= 2) {
    int ap = 6, bp = 5;
    double as = xins, bs = yins;
    if (as <= bs && zins < bs) { bs = zins; bp = 3; }
    else if (as > bs && zins < as) { as = zins; ap = 3; }
    double wins = 3 - in_sum;
    if (wins < as || wins < bs) {
      int cc = (bs < as) ? bp : ap;
      if ((cc & 1) != 0) { xe0 = xsb + 2; xe1 = xsb + 1; dxe0 = dx0 - 2 - 3 * SQ; dxe1 = dx0 - 1 - 3 * SQ; }
      else { xe0 = xe1 = xsb; dxe0 = dxe1 = dx0 - 3 * SQ; }
      if ((cc & 2) != 0) {
        ye0 = ye1 = ysb + 1; dye0 = dye1 = dy0 - 1 - 3 * SQ;
        if ((cc & 1) != 0) { ye1 += 1; dye1 -= 1; } else { ye0 += 1; dye0 -= 1; }
      } else { ye0 = ye1 = ysb; dye0 = dye1 = dy0 - 3 * SQ; }
      if ((cc & 4) != 0) { ze0 = zsb + 1; ze1 = zsb + 2; dze0 = dz0 - 1 - 3 * SQ; dze1 = dz0 - 2 - 3 * SQ; }
      else { ze0 = ze1 = zsb; dze0 = dze1 = dz0 - 3 * SQ; }
    } else {
      int cc = ap & bp;
      if ((cc & 1) != 0) { xe0 = xsb + 1; xe1 = xsb + 2; dxe0 = dx0 - 1 - SQ; dxe1 = dx0 - 2 - 2 * SQ; }
      else { xe0 = xe1 = xsb; dxe0 = dx0 - SQ; dxe1 = dx0 - 2 * SQ; }
      if ((cc & 2) != 0) { ye0 = ysb + 1; ye1 = ysb + 2; dye0 = dy0 - 1 - SQ; dye1 = dy0 - 2 - 2 * SQ; }
      else { ye0 = ye1 = ysb; dye0 = dy0 - SQ; dye1 = dy0 - 2 * SQ; }
      if ((cc & 4) != 0) { ze0 = zsb + 1; ze1 = zsb + 2; dze0 = dz0 - 1 - SQ; dze1 = dz0 - 2 - 2 * SQ; }
      else { ze0 = ze1 = zsb; dze0 = dz0 - SQ; dze1 = dz0 - 2 * SQ; }
    }
    double dx3 = dx0 - 1 - 2 * SQ, dy3 = dy0 - 1 - 2 * SQ, dz3 = dz0 - 0 - 2 * SQ;
    contrib(&c, xsb + 1, ysb + 1, zsb, dx3, dy3, dz3);
    double dx2 = dx3, dy2 = dy0 - 0 - 2 * SQ, dz2 = dz0 - 1 - 2 * SQ;
    contrib(&c, xsb + 1, ysb, zsb + 1, dx2, dy2, dz2);
    double dx1 = dx0 - 0 - 2 * SQ, dy1 = dy3, dz1 = dz2;
    contrib(&c, xsb, ysb + 1, zsb + 1, dx1, dy1, dz1);
    dx0 = dx0 - 1 - 3 * SQ; dy0 = dy0 - 1 - 3 * SQ; dz0 = dz0 - 1 - 3 * SQ;
    contrib(&c, xsb + 1, ysb + 1, zsb + 1, dx0, dy0, dz0);
  } else {
    double as, bs; int ap, bp, af, bf;
    double p1 = xins + yins;
    if (p1 > 1) { as = p1 - 1; ap = 3; af = 1; } else { as = 1 - p1; ap = 4; af = 0; }
    double p2 = xins + zins;
    if (p2 > 1) { bs = p2 - 1; bp = 5; bf = 1; } else { bs = 1 - p2; bp = 2; bf = 0; }
    double p3 = yins + zins;
    if (p3 > 1) {
      double sc = p3 - 1;
      if (as <= bs && as < sc) { as = sc; ap = 6; af = 1; }
      else if (as > bs && bs < sc) { bs = sc; bp = 6; bf = 1; }
    } else {
      double sc = 1 - p3;
      if (as <= bs && as < sc) { as = sc; ap = 1; af = 0; }
      else if (as > bs && bs < sc) { bs = sc; bp = 1; bf = 0; }
    }
    if (af == bf) {
      if (af) {
        dxe0 = dx0 - 1 - 3 * SQ; dye0 = dy0 - 1 - 3 * SQ; dze0 = dz0 - 1 - 3 * SQ;
        xe0 = xsb + 1; ye0 = ysb + 1; ze0 = zsb + 1;
        int cc = ap & bp;
        if ((cc & 1) != 0) { dxe1 = dx0 - 2 - 2 * SQ; dye1 = dy0 - 2 * SQ; dze1 = dz0 - 2 * SQ; xe1 = xsb + 2; ye1 = ysb; ze1 = zsb; }
        else if ((cc & 2) != 0) { dxe1 = dx0 - 2 * SQ; dye1 = dy0 - 2 - 2 * SQ; dze1 = dz0 - 2 * SQ; xe1 = xsb; ye1 = ysb + 2; ze1 = zsb; }
        else { dxe1 = dx0 - 2 * SQ; dye1 = dy0 - 2 * SQ; dze1 = dz0 - 2 - 2 * SQ; xe1 = xsb; ye1 = ysb; ze1 = zsb + 2; }
      } else {
        dxe0 = dx0; dye0 = dy0; dze0 = dz0; xe0 = xsb; ye0 = ysb; ze0 = zsb;
        int cc = ap | bp;
        if ((cc & 1) == 0) { dxe1 = dx0 + 1 - SQ; dye1 = dy0 - 1 - SQ; dze1 = dz0 - 1 - SQ; xe1 = xsb - 1; ye1 = ysb + 1; ze1 = zsb + 1; }
        else if ((cc & 2) == 0) { dxe1 = dx0 - 1 - SQ; dye1 = dy0 + 1 - SQ; dze1 = dz0 - 1 - SQ; xe1 = xsb + 1; ye1 = ysb - 1; ze1 = zsb + 1; }
        else { dxe1 = dx0 - 1 - SQ; dye1 = dy0 - 1 - SQ; dze1 = dz0 + 1 - SQ; xe1 = xsb + 1; ye1 = ysb + 1; ze1 = zsb - 1; }
      }
    } else {
      int c1 = af ? ap : bp, c2 = af ? bp : ap;
      if ((c1 & 1) == 0) { dxe0 = dx0 + 1 - SQ; dye0 = dy0 - 1 - SQ; dze0 = dz0 - 1 - SQ; xe0 = xsb - 1; ye0 = ysb + 1; ze0 = zsb + 1; }
      else if ((c1 & 2) == 0) { dxe0 = dx0 - 1 - SQ; dye0 = dy0 + 1 - SQ; dze0 = dz0 - 1 - SQ; xe0 = xsb + 1; ye0 = ysb - 1; ze0 = zsb + 1; }
      else { dxe0 = dx0 - 1 - SQ; dye0 = dy0 - 1 - SQ; dze0 = dz0 + 1 - SQ; xe0 = xsb + 1; ye0 = ysb + 1; ze0 = zsb - 1; }
      dxe1 = dx0 - 2 * SQ; dye1 = dy0 - 2 * SQ; dze1 = dz0 - 2 * SQ;
      xe1 = xsb; ye1 = ysb; ze1 = zsb;
      if ((c2 & 1) != 0) { dxe1 -= 2; xe1 += 2; }
      else if ((c2 & 2) != 0) { dye1 -= 2; ye1 += 2; }
      else { dze1 -= 2; ze1 += 2; }
    }
    double dx1 = dx0 - 1 - SQ, dy1 = dy0 - 0 - SQ, dz1 = dz0 - 0 - SQ;
    contrib(&c, xsb + 1, ysb, zsb, dx1, dy1, dz1);
    double dx2 = dx0 - 0 - SQ, dy2 = dy0 - 1 - SQ, dz2 = dz1;
    contrib(&c, xsb, ysb + 1, zsb, dx2, dy2, dz2);
    double dx3 = dx2, dy3 = dy1, dz3 = dz0 - 1 - SQ;
    contrib(&c, xsb, ysb, zsb + 1, dx3, dy3, dz3);
    double dx4 = dx0 - 1 - 2 * SQ, dy4 = dy0 - 1 - 2 * SQ, dz4 = dz0 - 0 - 2 * SQ;
    contrib(&c, xsb + 1, ysb + 1, zsb, dx4, dy4, dz4);
    double dx5 = dx4, dy5 = dy0 - 0 - 2 * SQ, dz5 = dz0 - 1 - 2 * SQ;
    contrib(&c, xsb + 1, ysb, zsb + 1, dx5, dy5, dz5);
    double dx6 = dx0 - 0 - 2 * SQ, dy6 = dy4, dz6 = dz5;
    contrib(&c, xsb, ysb + 1, zsb + 1, dx6, dy6, dz6);
  }
  contrib(&c, xe0, ye0, ze0, dxe0, dye0, dze0);
  contrib(&c, xe1, ye1, ze1, dxe1, dye1, dze1);
  return c.value / 103.0;
}

/* Batch helper: out[i] = noise3(xs[i], ys[i], zs[i]). */
void osn_noise3_many(const int16_t* perm, const int16_t* pg3, const double* xs, const double* ys,
                     const double* zs, double* out, int64_t n) {
  for (int64_t i = 0; i < n; i++) out[i] = osn_noise3(perm, pg3, xs[i], ys[i], zs[i]);
}
