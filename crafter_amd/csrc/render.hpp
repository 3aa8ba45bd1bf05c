// Observation render: LocalView (tiles + sprites + daylight / night-noise / sleep filters),
// ItemView and the final paste + transpose -- reference engine.py:155-248,267-284 and
// env.py:120-130; arithmetic spec SURVEY.md A.5 (pixel-exact against the reference there).
//
// Parallelisation: one lane per pixel.  The only sequential ingredient is the night noise
// (engine.py:208-209: uniform(32, 127, (local_w, local_h)) from the env's MT19937): the stream is
// consumed one twist epoch (624 words = 312 pixels) at a time, every lane tempering the two
// words of its own pixel straight out of the LDS-resident state, with a one-word carry for a
// double that straddles two epochs.  Nothing but the 2.5 KB state is buffered.
//
// All float arithmetic keeps the reference's evaluation order and width (f32 for the alpha
// blend and Pillow's colour blend, f64 for the filters); the library is compiled with
// -ffp-contract=off so no multiply-add is fused.
#pragma once
#include "env_core.hpp"

namespace crafter {

struct RenderTarget {
  uint8_t* out;      // [size_h][size_w][3] (already offset to this env), may be null when !pixels
  int size_w, size_h;
  int unit_x, unit_y;
  int border_x, border_y;
  int icon_w, icon_h, digit_w, digit_h;
  const int32_t* item_pos;   // [MAX_ITEMS][4]
  const int32_t* tex_tile;   // texture offsets for this unit
  const int32_t* tex_icon;
  const int32_t* tex_digit;
  const uint8_t* atlas;
  const double* vignette;    // [local_w][local_h]
};

template <class W>
struct Renderer {
  Env<W>& e;
  const RenderTarget& rt;
  int16_t* cell_tex;   // LDS [local_gw * local_gh] texture slot of the material, -1 outside the map
  int16_t* cell_obj;   // LDS [local_gw * local_gh] texture slot of the sprite, -1 if none

  __device__ Renderer(Env<W>& env, const RenderTarget& t, int16_t* ct, int16_t* co)
      : e(env), rt(t), cell_tex(ct), cell_obj(co) {}

  // objects.py:85-93,271,291,323,361-367,395-399
  __device__ int sprite_of(const Obj& o) const {
    int f = (o.fx < 0) ? 0 : (o.fx > 0) ? 1 : (o.fy < 0) ? 2 : 3;
    switch (o.type) {
      case T_PLAYER: return e.rec->sleeping ? TEX_PLAYER_SLEEP : TEX_PLAYER_LEFT + f;
      case T_COW: return TEX_COW;
      case T_ZOMBIE: return TEX_ZOMBIE;
      case T_SKELETON: return TEX_SKELETON;
      case T_ARROW: return TEX_ARROW_LEFT + f;
      case T_PLANT: return o.aux > 300 ? TEX_PLANT_RIPE : TEX_PLANT;
    }
    return TEX_UNKNOWN;
  }

  // engine.py:168-180: which texture each of the 9x7 grid cells shows
  __device__ __forceinline__ void build_cells() {
    const Config& c = e.cfg;
    Obj p = e.objs[1];
    int offx = c.local_gw / 2, offy = c.local_gh / 2;
    e.w.block_for(c.local_gw * c.local_gh, [&](int k) {
      int gx = k / c.local_gh, gy = k - gx * c.local_gh;
      int wx = (int)p.x + gx - offx, wy = (int)p.y + gy - offy;
      int t = -1, s = -1;
      if (e.inside(wx, wy)) {
        int ci = e.cidx(wx, wy);
        t = TEX_MATERIAL0 + e.mat[ci];
        int slot = e.objmap[ci];
        if (slot) s = sprite_of(e.objs[slot]);
      }
      cell_tex[k] = (int16_t)t;
      cell_obj[k] = (int16_t)s;
    });
    e.w.sync();
  }

  // engine.py:276-284 _draw_alpha on one pixel; c holds the current canvas bytes
  __device__ void blend(const uint8_t* texel, bool has_alpha, uint8_t c[3]) const {
    if (!has_alpha) {
      c[0] = texel[0];
      c[1] = texel[1];
      c[2] = texel[2];
      return;
    }
    const float* u = e.tb.unit255;
    float a = u[texel[3]];
    float ia = 1.0f - a;
    for (int ch = 0; ch < 3; ch++) {
      float b = a * u[texel[ch]] + ia * u[c[ch]];
      c[ch] = (uint8_t)(int)(255.0f * b);
    }
  }

  __device__ static int luma(const uint8_t c[3]) {  // Pillow RGB -> L
    return (19595 * (int)c[0] + 38470 * (int)c[1] + 7471 * (int)c[2] + 0x8000) >> 16;
  }

  // one LocalView pixel (x, y): tile, sprite, _light, _sleep  (engine.py:165-202)
  __device__ void local_pixel(int x, int y, bool night, double noise, double amount, double D, bool sleeping,
                              uint8_t out[3]) const {
    const Config& c = e.cfg;
    int gx = x / rt.unit_x, tx = x - gx * rt.unit_x;
    int gy = y / rt.unit_y, ty = y - gy * rt.unit_y;
    int k = gx * c.local_gh + gy;
    int texel = (tx * rt.unit_y + ty) * 4;
    uint8_t v[3] = {127, 127, 127};
    int t = cell_tex[k];
    if (t >= 0) {
      const uint8_t* p = rt.atlas + rt.tex_tile[t] + texel;
      v[0] = p[0];
      v[1] = p[1];
      v[2] = p[2];
    }
    int s = cell_obj[k];
    if (s >= 0) blend(rt.atlas + rt.tex_tile[s] + texel, e.tb.tex_alpha[s] != 0, v);
    // _light: night = noise-blended copy, desaturated 0.4, tinted; out = D*canvas + (1-D)*night
    uint8_t n8[3] = {v[0], v[1], v[2]};
    if (night) {
      int lh = c.local_gh * rt.unit_y;
      double m = amount * rt.vignette[x * lh + y];
      double im = 1 - m;
      double mn = m * noise;
      for (int ch = 0; ch < 3; ch++) n8[ch] = (uint8_t)(int)(im * (double)v[ch] + mn);
    }
    int L = luma(n8);
    const double tint[3] = {0.0, 16.0, 64.0};
    double o[3];
    double iD = 1 - D;
    for (int ch = 0; ch < 3; ch++) {
      float ef = (float)L + 0.4f * (float)((int)n8[ch] - L);  // Pillow ImagingBlend, C float
      int e8 = (uint8_t)(int)ef;
      double nt = 0.5 * (double)e8 + 0.5 * tint[ch];
      o[ch] = D * (double)v[ch] + iD * nt;
    }
    if (sleeping) {  // engine.py:198-202
      uint8_t o8[3] = {(uint8_t)(int)o[0], (uint8_t)(int)o[1], (uint8_t)(int)o[2]};
      double g = (double)luma(o8);
      o[0] = 0.5 * g + 0.5 * 0.0;
      o[1] = 0.5 * g + 0.5 * 0.0;
      o[2] = 0.5 * g + 0.5 * 16.0;
    }
    out[0] = (uint8_t)(int)o[0];
    out[1] = (uint8_t)(int)o[1];
    out[2] = (uint8_t)(int)o[2];
  }

  // one ItemView pixel (x, iy)  (engine.py:227-248)
  __device__ void item_pixel(int x, int iy, uint8_t out[3]) const {
    const Config& c = e.cfg;
    out[0] = out[1] = out[2] = 0;
    int cx = x / rt.unit_x, cy = iy / rt.unit_y;
    int k = cy * c.item_gw + cx;
    if (cx >= c.item_gw || cy >= c.item_gh || k >= e.R.n_items) return;
    int amount = e.rec->inv[k];
    if (amount < 1) return;
    const int32_t* pos = rt.item_pos + k * 4;
    int ix = x - pos[0], iyy = iy - pos[1];
    if (ix >= 0 && iyy >= 0 && ix < rt.icon_w && iyy < rt.icon_h)
      blend(rt.atlas + rt.tex_icon[k] + (ix * rt.icon_h + iyy) * 4, e.tb.tex_alpha[TEX_COUNT + k] != 0, out);
    int dx = x - pos[2], dy = iy - pos[3];
    if (dx >= 0 && dy >= 0 && dx < rt.digit_w && dy < rt.digit_h) {
      int d = (amount >= 1 && amount <= 9) ? amount : 10;  // engine.py:245 ('unknown' otherwise)
      blend(rt.atlas + rt.tex_digit[d] + (dx * rt.digit_h + dy) * 4,
            e.tb.tex_alpha[TEX_COUNT + MAX_ITEMS + d] != 0, out);
    }
  }

  __device__ void store_pixel(int X, int Y, const uint8_t v[3]) const {
    uint8_t* p = rt.out + ((size_t)Y * rt.size_w + X) * 3;
    p[0] = v[0];
    p[1] = v[1];
    p[2] = v[2];
  }

  // Full frame.  pixels == false: only the RNG side effect of a night frame happens.
  __device__ __forceinline__ void render(bool pixels) {
    const Config& c = e.cfg;
    int lw = c.local_gw * rt.unit_x, lh = c.local_gh * rt.unit_y;
    int ih = c.item_gh * rt.unit_y;
    double D = e.tb.daylight[e.rec->step];
    bool night = D < 0.5;
    bool sleeping = e.rec->sleeping != 0;
    if (pixels) {
      build_cells();
      // everything except (at night) the LocalView rectangle: canvas is (size_w, size_h) in
      // [x][y]; output is its transpose (env.py:123-130)
      e.w.block_for(rt.size_w * rt.size_h, [&](int p) {
        int Y = p / rt.size_w, X = p - Y * rt.size_w;
        int vx = X - rt.border_x, vy = Y - rt.border_y;
        uint8_t v[3] = {0, 0, 0};
        if (vx >= 0 && vy >= 0 && vx < lw && vy < lh + ih) {
          if (vy < lh) {
            if (night) return;  // written by the noise epochs below
            local_pixel(vx, vy, false, 0.0, 0.0, D, sleeping, v);
          } else {
            item_pixel(vx, vy - lh, v);
          }
        }
        store_pixel(X, Y, v);
      });
    }
    if (!night) return;
    // night: walk the MT19937 stream, 2 words per LocalView pixel, row-major over [x][y]
    double amount = 2 * (0.5 - D);
    int total = lw * lh;
    int words = 2 * total;
    int pos = e.mt_pos;
    int s_lo = 0;
    uint32_t carry = 0;
    while (s_lo < words) {
      if (pos >= MT_N) {
        carry = e.mt[MT_N - 1];
        e.w.sync();
        if (e.w.wave0()) e.w.mt_twist(e.mt);
        e.w.sync();
        pos = 0;
      }
      int s_hi = s_lo + (MT_N - pos);
      if (s_hi > words) s_hi = words;
      if (pixels) {
        int j_first = s_lo >> 1;          // if s_lo is odd its first word is the carry
        int j_last = (s_hi - 2) >> 1;     // last double whose second word lies in this epoch
        int count = (s_hi >= 2) ? (j_last - j_first + 1) : 0;
        const uint32_t* mt = e.mt;
        e.w.block_for(count, [&](int q) {
          int j = j_first + q;
          int ia = 2 * j - s_lo;
          uint32_t a = (ia >= 0) ? mt[pos + ia] : carry;
          uint32_t b = mt[pos + ia + 1];
          double noise = 32.0 + 95.0 * mt_double(mt_temper(a), mt_temper(b));
          int x = j / lh, y = j - x * lh;
          uint8_t v[3];
          local_pixel(x, y, true, noise, amount, D, sleeping, v);
          store_pixel(x + rt.border_x, y + rt.border_y, v);
        });
      }
      pos += s_hi - s_lo;
      s_lo = s_hi;
    }
    e.w.sync();
    e.mt_pos = pos;
  }
};

}  // namespace crafter
