"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement of the 3rd-party arithmetic the reference calls for terrain:
``opensimplex.OpenSimplex(seed).noise3(x, y, z)`` -- call sites
/root/reference/crafter/worldgen.py:11 (constructor, seed = randint(0, 2**31-1)) and
worldgen.py:84-87 (``noise3d`` for package <=0.3, ``noise3`` for >=0.4; same maths).

The package (PyPI ``opensimplex``, lmas/opensimplex, UN-PINNED in the reference's
setup.py:16) is NOT installed in this image and there is no network, so this file restates
the published public-domain algorithm it ports: Kurt Spencer, "OpenSimplex Noise in
Java" (2014, legacy OpenSimplex, not OpenSimplex2), as specified in SURVEY.md App. B.

PARITY STATUS: seeding + permutation + the 2-D path are pinned by the two upstream-README
known answers (tests/test_noise.py).  The 3-D path is **parity unpinned**: no value
produced by the real package is available here.  It is checked for internal consistency
(continuity, symmetry, the survey's provisional vectors) and against the C restatement.

Every expression is evaluated in float64, left to right, exactly as written (Python never
contracts a*b+c into an FMA).
"""
from math import floor

STRETCH_3D = -1.0 / 6.0
SQUISH_3D = 1.0 / 3.0
NORM_3D = 103.0
STRETCH_2D = -0.211324865405187
SQUISH_2D = 0.366025403784439
NORM_2D = 47.0

GRAD3 = (
    -11, 4, 4, -4, 11, 4, -4, 4, 11,
    11, 4, 4, 4, 11, 4, 4, 4, 11,
    -11, -4, 4, -4, -11, 4, -4, -4, 11,
    11, -4, 4, 4, -11, 4, 4, -4, 11,
    -11, 4, -4, -4, 11, -4, -4, 4, -11,
    11, 4, -4, 4, 11, -4, 4, 4, -11,
    -11, -4, -4, -4, -11, -4, -4, -4, -11,
    11, -4, -4, 4, -11, -4, 4, -4, -11,
)
GRAD2 = (5, 2, 2, 5, -5, 2, -2, 5, 5, -2, 2, -5, -5, -2, -2, -5)

_M64 = (1 << 64) - 1


def _wrap64(v):
  """Signed 64-bit wrap-around (Java long / numpy int64 overflow)."""
  v &= _M64
  return v - (1 << 64) if v >> 63 else v


def make_perm(seed):
  """Permutation tables of OpenSimplex(seed): returns (perm[256], perm_grad3[256])."""
  perm = [0] * 256
  pg3 = [0] * 256
  source = list(range(256))
  for _ in range(3):
    seed = _wrap64(seed * 6364136223846793005 + 1442695040888963407)
  for i in range(255, -1, -1):
    seed = _wrap64(seed * 6364136223846793005 + 1442695040888963407)
    r = (seed + 31) % (i + 1)  # Python %: non-negative remainder
    perm[i] = source[r]
    pg3[i] = (perm[i] % 24) * 3
    source[r] = source[i]
  return perm, pg3


class OpenSimplex:
  """Same surface the reference touches: OpenSimplex(seed).noise3(x, y, z) (+ noise2 for KATs)."""

  def __init__(self, seed=0):
    self.perm, self.pg3 = make_perm(int(seed))

  # -- 2-D, only used to pin seeding against the upstream README known answers ---------
  def noise2(self, x, y):
    perm = self.perm
    so = (x + y) * STRETCH_2D
    xs = x + so
    ys = y + so
    xsb = floor(xs)
    ysb = floor(ys)
    qo = (xsb + ysb) * SQUISH_2D
    xb = xsb + qo
    yb = ysb + qo
    xins = xs - xsb
    yins = ys - ysb
    in_sum = xins + yins
    dx0 = x - xb
    dy0 = y - yb
    value = 0.0

    def contrib(xsv, ysv, dx, dy):
      attn = 2 - dx * dx - dy * dy
      if attn > 0:
        i = perm[(perm[xsv & 0xFF] + ysv) & 0xFF] & 0x0E
        attn *= attn
        return attn * attn * (GRAD2[i] * dx + GRAD2[i + 1] * dy)
      return 0.0

    dx1 = dx0 - 1 - SQUISH_2D
    dy1 = dy0 - 0 - SQUISH_2D
    value += contrib(xsb + 1, ysb + 0, dx1, dy1)
    dx2 = dx0 - 0 - SQUISH_2D
    dy2 = dy0 - 1 - SQUISH_2D
    value += contrib(xsb + 0, ysb + 1, dx2, dy2)
    if in_sum <= 1:
      zins = 1 - in_sum
      if zins > xins or zins > yins:
        if xins > yins:
          xsv_ext, ysv_ext = xsb + 1, ysb - 1
          dx_ext, dy_ext = dx0 - 1, dy0 + 1
        else:
          xsv_ext, ysv_ext = xsb - 1, ysb + 1
          dx_ext, dy_ext = dx0 + 1, dy0 - 1
      else:
        xsv_ext, ysv_ext = xsb + 1, ysb + 1
        dx_ext = dx0 - 1 - 2 * SQUISH_2D
        dy_ext = dy0 - 1 - 2 * SQUISH_2D
    else:
      zins = 2 - in_sum
      if zins < xins or zins < yins:
        if xins > yins:
          xsv_ext, ysv_ext = xsb + 2, ysb + 0
          dx_ext = dx0 - 2 - 2 * SQUISH_2D
          dy_ext = dy0 + 0 - 2 * SQUISH_2D
        else:
          xsv_ext, ysv_ext = xsb + 0, ysb + 2
          dx_ext = dx0 + 0 - 2 * SQUISH_2D
          dy_ext = dy0 - 2 - 2 * SQUISH_2D
      else:
        dx_ext, dy_ext = dx0, dy0
        xsv_ext, ysv_ext = xsb, ysb
      xsb += 1
      ysb += 1
      dx0 = dx0 - 1 - 2 * SQUISH_2D
      dy0 = dy0 - 1 - 2 * SQUISH_2D
    value += contrib(xsb, ysb, dx0, dy0)
    value += contrib(xsv_ext, ysv_ext, dx_ext, dy_ext)
    return value / NORM_2D

  # -- 3-D: the path worldgen uses ------------------------------------------------------
  def noise3(self, x, y, z):
    perm = self.perm
    pg3 = self.pg3
    S = SQUISH_3D
    so = (x + y + z) * STRETCH_3D
    xs = x + so
    ys = y + so
    zs = z + so
    xsb = floor(xs)
    ysb = floor(ys)
    zsb = floor(zs)
    qo = (xsb + ysb + zsb) * S
    xb = xsb + qo
    yb = ysb + qo
    zb = zsb + qo
    xins = xs - xsb
    yins = ys - ysb
    zins = zs - zsb
    in_sum = xins + yins + zins
    dx0 = x - xb
    dy0 = y - yb
    dz0 = z - zb
    acc = [0.0]

    def C(xsv, ysv, zsv, dx, dy, dz):
      attn = 2 - dx * dx - dy * dy - dz * dz
      if attn > 0:
        g = pg3[(perm[(perm[xsv & 0xFF] + ysv) & 0xFF] + zsv) & 0xFF]
        attn *= attn
        acc[0] += attn * attn * (GRAD3[g] * dx + GRAD3[g + 1] * dy + GRAD3[g + 2] * dz)

    if in_sum <= 1:
      a_point, a_score = 0x01, xins
      b_point, b_score = 0x02, yins
      if a_score >= b_score and zins > b_score:
        b_score, b_point = zins, 0x04
      elif a_score < b_score and zins > a_score:
        a_score, a_point = zins, 0x04
      wins = 1 - in_sum
      if wins > a_score or wins > b_score:
        c = b_point if b_score > a_score else a_point
        if (c & 0x01) == 0:
          xe0, xe1 = xsb - 1, xsb
          dxe0, dxe1 = dx0 + 1, dx0
        else:
          xe0 = xe1 = xsb + 1
          dxe0 = dxe1 = dx0 - 1
        if (c & 0x02) == 0:
          ye0 = ye1 = ysb
          dye0 = dye1 = dy0
          if (c & 0x01) == 0:
            ye1 -= 1
            dye1 += 1
          else:
            ye0 -= 1
            dye0 += 1
        else:
          ye0 = ye1 = ysb + 1
          dye0 = dye1 = dy0 - 1
        if (c & 0x04) == 0:
          ze0, ze1 = zsb, zsb - 1
          dze0, dze1 = dz0, dz0 + 1
        else:
          ze0 = ze1 = zsb + 1
          dze0 = dze1 = dz0 - 1
      else:
        c = a_point | b_point
        if (c & 0x01) == 0:
          xe0, xe1 = xsb, xsb - 1
          dxe0 = dx0 - 2 * S
          dxe1 = dx0 + 1 - S
        else:
          xe0 = xe1 = xsb + 1
          dxe0 = dx0 - 1 - 2 * S
          dxe1 = dx0 - 1 - S
        if (c & 0x02) == 0:
          ye0, ye1 = ysb, ysb - 1
          dye0 = dy0 - 2 * S
          dye1 = dy0 + 1 - S
        else:
          ye0 = ye1 = ysb + 1
          dye0 = dy0 - 1 - 2 * S
          dye1 = dy0 - 1 - S
        if (c & 0x04) == 0:
          ze0, ze1 = zsb, zsb - 1
          dze0 = dz0 - 2 * S
          dze1 = dz0 + 1 - S
        else:
          ze0 = ze1 = zsb + 1
          dze0 = dz0 - 1 - 2 * S
          dze1 = dz0 - 1 - S
      C(xsb, ysb, zsb, dx0, dy0, dz0)
      dx1 = dx0 - 1 - S
      dy1 = dy0 - 0 - S
      dz1 = dz0 - 0 - S
      C(xsb + 1, ysb, zsb, dx1, dy1, dz1)
      dx2 = dx0 - 0 - S
      dy2 = dy0 - 1 - S
      dz2 = dz1
      C(xsb, ysb + 1, zsb, dx2, dy2, dz2)
      dx3 = dx2
      dy3 = dy1
      dz3 = dz0 - 1 - S
      C(xsb, ysb, zsb + 1, dx3, dy3, dz3)
    elif in_sum >= 2:
      a_point, a_score = 0x06, xins
      b_point, b_score = 0x05, yins
      if a_score <= b_score and zins < b_score:
        b_score, b_point = zins, 0x03
      elif a_score > b_score and zins < a_score:
        a_score, a_point = zins, 0x03
      wins = 3 - in_sum
      if wins < a_score or wins < b_score:
        c = b_point if b_score < a_score else a_point
        if (c & 0x01) != 0:
          xe0, xe1 = xsb + 2, xsb + 1
          dxe0 = dx0 - 2 - 3 * S
          dxe1 = dx0 - 1 - 3 * S
        else:
          xe0 = xe1 = xsb
          dxe0 = dxe1 = dx0 - 3 * S
        if (c & 0x02) != 0:
          ye0 = ye1 = ysb + 1
          dye0 = dye1 = dy0 - 1 - 3 * S
          if (c & 0x01) != 0:
            ye1 += 1
            dye1 -= 1
          else:
            ye0 += 1
            dye0 -= 1
        else:
          ye0 = ye1 = ysb
          dye0 = dye1 = dy0 - 3 * S
        if (c & 0x04) != 0:
          ze0, ze1 = zsb + 1, zsb + 2
          dze0 = dz0 - 1 - 3 * S
          dze1 = dz0 - 2 - 3 * S
        else:
          ze0 = ze1 = zsb
          dze0 = dze1 = dz0 - 3 * S
      else:
        c = a_point & b_point
        if (c & 0x01) != 0:
          xe0, xe1 = xsb + 1, xsb + 2
          dxe0 = dx0 - 1 - S
          dxe1 = dx0 - 2 - 2 * S
        else:
          xe0 = xe1 = xsb
          dxe0 = dx0 - S
          dxe1 = dx0 - 2 * S
        if (c & 0x02) != 0:
          ye0, ye1 = ysb + 1, ysb + 2
          dye0 = dy0 - 1 - S
          dye1 = dy0 - 2 - 2 * S
        else:
          ye0 = ye1 = ysb
          dye0 = dy0 - S
          dye1 = dy0 - 2 * S
        if (c & 0x04) != 0:
          ze0, ze1 = zsb + 1, zsb + 2
          dze0 = dz0 - 1 - S
          dze1 = dz0 - 2 - 2 * S
        else:
          ze0 = ze1 = zsb
          dze0 = dz0 - S
          dze1 = dz0 - 2 * S
      dx3 = dx0 - 1 - 2 * S
      dy3 = dy0 - 1 - 2 * S
      dz3 = dz0 - 0 - 2 * S
      C(xsb + 1, ysb + 1, zsb, dx3, dy3, dz3)
      dx2 = dx3
      dy2 = dy0 - 0 - 2 * S
      dz2 = dz0 - 1 - 2 * S
      C(xsb + 1, ysb, zsb + 1, dx2, dy2, dz2)
      dx1 = dx0 - 0 - 2 * S
      dy1 = dy3
      dz1 = dz2
      C(xsb, ysb + 1, zsb + 1, dx1, dy1, dz1)
      dx0 = dx0 - 1 - 3 * S
      dy0 = dy0 - 1 - 3 * S
      dz0 = dz0 - 1 - 3 * S
      C(xsb + 1, ysb + 1, zsb + 1, dx0, dy0, dz0)
    else:
      p1 = xins + yins
      if p1 > 1:
        a_score, a_point, a_far = p1 - 1, 0x03, True
      else:
        a_score, a_point, a_far = 1 - p1, 0x04, False
      p2 = xins + zins
      if p2 > 1:
        b_score, b_point, b_far = p2 - 1, 0x05, True
      else:
        b_score, b_point, b_far = 1 - p2, 0x02, False
      p3 = yins + zins
      if p3 > 1:
        score = p3 - 1
        if a_score <= b_score and a_score < score:
          a_score, a_point, a_far = score, 0x06, True
        elif a_score > b_score and b_score < score:
          b_score, b_point, b_far = score, 0x06, True
      else:
        score = 1 - p3
        if a_score <= b_score and a_score < score:
          a_score, a_point, a_far = score, 0x01, False
        elif a_score > b_score and b_score < score:
          b_score, b_point, b_far = score, 0x01, False
      if a_far == b_far:
        if a_far:
          dxe0 = dx0 - 1 - 3 * S
          dye0 = dy0 - 1 - 3 * S
          dze0 = dz0 - 1 - 3 * S
          xe0, ye0, ze0 = xsb + 1, ysb + 1, zsb + 1
          c = a_point & b_point
          if (c & 0x01) != 0:
            dxe1 = dx0 - 2 - 2 * S
            dye1 = dy0 - 2 * S
            dze1 = dz0 - 2 * S
            xe1, ye1, ze1 = xsb + 2, ysb, zsb
          elif (c & 0x02) != 0:
            dxe1 = dx0 - 2 * S
            dye1 = dy0 - 2 - 2 * S
            dze1 = dz0 - 2 * S
            xe1, ye1, ze1 = xsb, ysb + 2, zsb
          else:
            dxe1 = dx0 - 2 * S
            dye1 = dy0 - 2 * S
            dze1 = dz0 - 2 - 2 * S
            xe1, ye1, ze1 = xsb, ysb, zsb + 2
        else:
          dxe0, dye0, dze0 = dx0, dy0, dz0
          xe0, ye0, ze0 = xsb, ysb, zsb
          c = a_point | b_point
          if (c & 0x01) == 0:
            dxe1 = dx0 + 1 - S
            dye1 = dy0 - 1 - S
            dze1 = dz0 - 1 - S
            xe1, ye1, ze1 = xsb - 1, ysb + 1, zsb + 1
          elif (c & 0x02) == 0:
            dxe1 = dx0 - 1 - S
            dye1 = dy0 + 1 - S
            dze1 = dz0 - 1 - S
            xe1, ye1, ze1 = xsb + 1, ysb - 1, zsb + 1
          else:
            dxe1 = dx0 - 1 - S
            dye1 = dy0 - 1 - S
            dze1 = dz0 + 1 - S
            xe1, ye1, ze1 = xsb + 1, ysb + 1, zsb - 1
      else:
        if a_far:
          c1, c2 = a_point, b_point
        else:
          c1, c2 = b_point, a_point
        if (c1 & 0x01) == 0:
          dxe0 = dx0 + 1 - S
          dye0 = dy0 - 1 - S
          dze0 = dz0 - 1 - S
          xe0, ye0, ze0 = xsb - 1, ysb + 1, zsb + 1
        elif (c1 & 0x02) == 0:
          dxe0 = dx0 - 1 - S
          dye0 = dy0 + 1 - S
          dze0 = dz0 - 1 - S
          xe0, ye0, ze0 = xsb + 1, ysb - 1, zsb + 1
        else:
          dxe0 = dx0 - 1 - S
          dye0 = dy0 - 1 - S
          dze0 = dz0 + 1 - S
          xe0, ye0, ze0 = xsb + 1, ysb + 1, zsb - 1
        dxe1 = dx0 - 2 * S
        dye1 = dy0 - 2 * S
        dze1 = dz0 - 2 * S
        xe1, ye1, ze1 = xsb, ysb, zsb
        if (c2 & 0x01) != 0:
          dxe1 -= 2
          xe1 += 2
        elif (c2 & 0x02) != 0:
          dye1 -= 2
          ye1 += 2
        else:
          dze1 -= 2
          ze1 += 2
      dx1 = dx0 - 1 - S
      dy1 = dy0 - 0 - S
      dz1 = dz0 - 0 - S
      C(xsb + 1, ysb, zsb, dx1, dy1, dz1)
      dx2 = dx0 - 0 - S
      dy2 = dy0 - 1 - S
      dz2 = dz1
      C(xsb, ysb + 1, zsb, dx2, dy2, dz2)
      dx3 = dx2
      dy3 = dy1
      dz3 = dz0 - 1 - S
      C(xsb, ysb, zsb + 1, dx3, dy3, dz3)
      dx4 = dx0 - 1 - 2 * S
      dy4 = dy0 - 1 - 2 * S
      dz4 = dz0 - 0 - 2 * S
      C(xsb + 1, ysb + 1, zsb, dx4, dy4, dz4)
      dx5 = dx4
      dy5 = dy0 - 0 - 2 * S
      dz5 = dz0 - 1 - 2 * S
      C(xsb + 1, ysb, zsb + 1, dx5, dy5, dz5)
      dx6 = dx0 - 0 - 2 * S
      dy6 = dy4
      dz6 = dz5
      C(xsb, ysb + 1, zsb + 1, dx6, dy6, dz6)
    C(xe0, ye0, ze0, dxe0, dye0, dze0)
    C(xe1, ye1, ze1, dxe1, dye1, dze1)
    return acc[0] / NORM_3D

  noise3d = noise3
  noise2d = noise2
