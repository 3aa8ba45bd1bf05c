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

// Day frames: the lit colour of a sprite-free pixel depends only on (material, texel, daylight), so
// it is computed once per material present in view and looked up per pixel.  Only when the table is
// small (unit 7: 13 x 49 x 4 B = 2.5 KB); big render sizes compute every pixel.
__host__ __device__ inline int lit_cache_bytes(const Config& c) {
  int bytes = (MAX_MATERIALS + 1) * c.unit_x * c.unit_y * 4;
  return bytes <= 4096 ? align16(bytes) : 0;
}

// LDS tables the renderer builds once per frame (bytes, 16-byte aligned total)
__host__ __device__ inline int render_lds_bytes(const Config& c) {
  int ncell = c.local_gw * c.local_gh;
  int lw = c.local_gw * c.unit_x, vh = (c.local_gh + c.item_gh) * c.unit_y;
  return align16(8 * ncell) + align16(2 * lw) + align16(2 * vh) + MAX_ITEMS * 32 + lit_cache_bytes(c);
}

template <class W>
struct Renderer {
  Env<W>& e;
  const RenderTarget& rt;
  int32_t* cell_tile;    // LDS [ncell] atlas byte offset of the cell's material texture, -1 outside the map
  int32_t* cell_sprite;  // LDS [ncell] atlas byte offset of the cell's sprite | ALPHA_BIT, -1 if none
  uint16_t* colmap;      // LDS [local_w]          view x pixel -> cell column | texel x << 8
  uint16_t* rowmap;      // LDS [local_h + item_h] view y pixel -> cell row | texel y << 8 (item rows restart at 0)
  int32_t* item_tab;     // LDS [MAX_ITEMS][8] icon off|ALPHA, digit off|ALPHA, icon x,y, digit x,y, amount, -
  uint32_t* lit;         // LDS [materials + 1][unit_x * unit_y] lit RGB of sprite-free day pixels, or null
  uint32_t* present;     // LDS bitmask of material ids visible in this frame (one word)
  uint64_t* prof = nullptr;  // optional shader-clock stamps (slots 7, 8)

  static constexpr int32_t ALPHA_BIT = 1 << 30;
  static constexpr int32_t OFF_MASK = ALPHA_BIT - 1;

  __device__ Renderer(Env<W>& env, const RenderTarget& t, uint8_t* lds) : e(env), rt(t) {
    const Config& c = e.cfg;
    int ncell = c.local_gw * c.local_gh;
    int lw = c.local_gw * c.unit_x, vh = (c.local_gh + c.item_gh) * c.unit_y;
    cell_tile = (int32_t*)lds;
    cell_sprite = cell_tile + ncell;
    colmap = (uint16_t*)(lds + align16(8 * ncell));
    rowmap = (uint16_t*)(lds + align16(8 * ncell) + align16(2 * lw));
    item_tab = (int32_t*)(lds + align16(8 * ncell) + align16(2 * lw) + align16(2 * vh));
    lit = lit_cache_bytes(c) ? (uint32_t*)((uint8_t*)item_tab + MAX_ITEMS * 32) : nullptr;
    present = (uint32_t*)(item_tab + 7);   // spare word of item slot 0
  }

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

  struct Lit {
    double D, iD, amount;
    bool night, sleeping;
  };

  // Per-frame tables: which texture each of the 9x7 grid cells shows (engine.py:168-180), the
  // pixel -> (cell, texel) maps (so the pixel loops contain no division) and the inventory slots
  // (engine.py:227-248).
  __device__ __forceinline__ void build_tables(const Lit& L) {
    const Config& c = e.cfg;
    if (e.w.leader()) *present = 0;
    e.w.sync();
    Obj p = e.objs[1];
    int offx = c.local_gw / 2, offy = c.local_gh / 2;
    int lw = c.local_gw * rt.unit_x, lh = c.local_gh * rt.unit_y, ih = c.item_gh * rt.unit_y;
    e.w.block_for(c.local_gw * c.local_gh, [&](int k) {
      int gx = k / c.local_gh, gy = k - gx * c.local_gh;
      int wx = (int)p.x + gx - offx, wy = (int)p.y + gy - offy;
      int32_t t = -1, s = -1;
      if (e.inside(wx, wy)) {
        int ci = e.cidx(wx, wy);
        int m = e.mat[ci];
        t = rt.tex_tile[TEX_MATERIAL0 + m] | (m << 24);   // atlas offsets are < 2^24
        e.w.lds_or(present, 1u << m);
        int slot = e.objmap[ci];
        if (slot) {
          int sp = sprite_of(e.objs[slot]);
          s = rt.tex_tile[sp] | (e.tb.tex_alpha[sp] ? ALPHA_BIT : 0);
        }
      }
      cell_tile[k] = t;
      cell_sprite[k] = s;
    });
    e.w.block_for(lw, [&](int x) {
      int g = x / rt.unit_x;
      colmap[x] = (uint16_t)(g | ((x - g * rt.unit_x) << 8));
    });
    e.w.block_for(lh + ih, [&](int y) {
      int yy = y < lh ? y : y - lh;
      int g = yy / rt.unit_y;
      rowmap[y] = (uint16_t)(g | ((yy - g * rt.unit_y) << 8));
    });
    e.w.block_for(e.R.n_items, [&](int k) {
      int amount = e.rec->inv[k];
      int d = (amount >= 1 && amount <= 9) ? amount : 10;  // engine.py:245 ('unknown' otherwise)
      int32_t* t = item_tab + k * 8;
      t[0] = rt.tex_icon[k] | (e.tb.tex_alpha[TEX_COUNT + k] ? ALPHA_BIT : 0);
      t[1] = rt.tex_digit[d] | (e.tb.tex_alpha[TEX_COUNT + MAX_ITEMS + d] ? ALPHA_BIT : 0);
      t[2] = rt.item_pos[k * 4 + 0];
      t[3] = rt.item_pos[k * 4 + 1];
      t[4] = rt.item_pos[k * 4 + 2];
      t[5] = rt.item_pos[k * 4 + 3];
      t[6] = amount;
    });
    e.w.sync();
    if (lit && !L.night) {
      int ntex = rt.unit_x * rt.unit_y;
      uint32_t mask = *present;
      e.w.block_for((MAX_MATERIALS + 1) * ntex, [&](int i) {
        int m = i / ntex, texel = i - m * ntex;
        if (!((mask >> m) & 1u)) return;
        uint32_t tile = *(const uint32_t*)(rt.atlas + rt.tex_tile[TEX_MATERIAL0 + m] + texel * 4);
        int v[3] = {(int)(tile & 0xFF), (int)((tile >> 8) & 0xFF), (int)((tile >> 16) & 0xFF)};
        lit[i] = light(v, L, 0.0, 0.0);
      });
      e.w.sync();
    }
  }

  // engine.py:276-284 _draw_alpha on one pixel: texel = packed RGBA (little endian), c = canvas bytes
  __device__ static void blend(uint32_t texel, bool has_alpha, int c[3]) {
    int t0 = texel & 0xFF, t1 = (texel >> 8) & 0xFF, t2 = (texel >> 16) & 0xFF;
    if (!has_alpha) {
      c[0] = t0;
      c[1] = t1;
      c[2] = t2;
      return;
    }
    float a = W::fdiv((float)(texel >> 24), 255.0f);   // arr.astype(float32) / 255, correctly rounded
    float ia = 1.0f - a;
    float b0 = a * W::fdiv((float)t0, 255.0f) + ia * W::fdiv((float)c[0], 255.0f);
    float b1 = a * W::fdiv((float)t1, 255.0f) + ia * W::fdiv((float)c[1], 255.0f);
    float b2 = a * W::fdiv((float)t2, 255.0f) + ia * W::fdiv((float)c[2], 255.0f);
    c[0] = (int)(255.0f * b0);
    c[1] = (int)(255.0f * b1);
    c[2] = (int)(255.0f * b2);
  }

  __device__ static int luma(int r, int g, int b) {  // Pillow RGB -> L
    return (19595 * r + 38470 * g + 7471 * b + 0x8000) >> 16;
  }

  // tile + sprite of LocalView pixel (vx, vy)  (engine.py:168-180)
  __device__ void local_colour(int vx, int vy, int v[3]) const {
    int cm = colmap[vx], rm = rowmap[vy];
    int k = (cm & 0xFF) * e.cfg.local_gh + (rm & 0xFF);
    int texel = ((cm >> 8) * rt.unit_y + (rm >> 8)) * 4;
    int32_t t = cell_tile[k], s = cell_sprite[k];
    uint32_t tile = 0x7F7F7F7Fu, sprite = 0;
    if (t >= 0) tile = *(const uint32_t*)(rt.atlas + (t & 0xFFFFFF) + texel);
    if (s >= 0) sprite = *(const uint32_t*)(rt.atlas + (s & OFF_MASK) + texel);
    v[0] = tile & 0xFF;
    v[1] = (tile >> 8) & 0xFF;
    v[2] = (tile >> 16) & 0xFF;
    if (s >= 0) blend(sprite, (s & ALPHA_BIT) != 0, v);
  }

  // _light and _sleep on one pixel (engine.py:189-202); returns packed 0x00BBGGRR
  __device__ static uint32_t light(const int v[3], const Lit& L, double m, double noise) {
    int n0 = v[0], n1 = v[1], n2 = v[2];
    if (L.night) {
      double im = 1 - m;
      double mn = m * noise;
      n0 = (int)(im * (double)v[0] + mn);
      n1 = (int)(im * (double)v[1] + mn);
      n2 = (int)(im * (double)v[2] + mn);
    }
    int lum = luma(n0, n1, n2);
    // Pillow ImagingBlend (C float): (u8)(L + 0.4f * (c - L)); then tint (0, 16, 64) by 0.5
    int e0 = (int)((float)lum + 0.4f * (float)(n0 - lum));
    int e1 = (int)((float)lum + 0.4f * (float)(n1 - lum));
    int e2 = (int)((float)lum + 0.4f * (float)(n2 - lum));
    double o0 = L.D * (double)v[0] + L.iD * (0.5 * (double)e0 + 0.5 * 0.0);
    double o1 = L.D * (double)v[1] + L.iD * (0.5 * (double)e1 + 0.5 * 16.0);
    double o2 = L.D * (double)v[2] + L.iD * (0.5 * (double)e2 + 0.5 * 64.0);
    if (L.sleeping) {  // engine.py:198-202
      double g = (double)luma((int)o0, (int)o1, (int)o2);
      o0 = 0.5 * g + 0.5 * 0.0;
      o1 = 0.5 * g + 0.5 * 0.0;
      o2 = 0.5 * g + 0.5 * 16.0;
    }
    return (uint32_t)(int)o0 | ((uint32_t)(int)o1 << 8) | ((uint32_t)(int)o2 << 16);
  }

  // one ItemView pixel (vx, row index vy in the combined view)  (engine.py:227-248)
  __device__ uint32_t item_pixel(int vx, int vy, int iy) const {
    const Config& c = e.cfg;
    int cm = colmap[vx], rm = rowmap[vy];
    int k = (rm & 0xFF) * c.item_gw + (cm & 0xFF);
    if (k >= e.R.n_items) return 0;
    const int32_t* t = item_tab + k * 8;
    if (t[6] < 1) return 0;
    int v[3] = {0, 0, 0};
    int ix = vx - t[2], iyy = iy - t[3];
    if (ix >= 0 && iyy >= 0 && ix < rt.icon_w && iyy < rt.icon_h)
      blend(*(const uint32_t*)(rt.atlas + (t[0] & OFF_MASK) + (ix * rt.icon_h + iyy) * 4), (t[0] & ALPHA_BIT) != 0, v);
    int dx = vx - t[4], dy = iy - t[5];
    if (dx >= 0 && dy >= 0 && dx < rt.digit_w && dy < rt.digit_h)
      blend(*(const uint32_t*)(rt.atlas + (t[1] & OFF_MASK) + (dx * rt.digit_h + dy) * 4), (t[1] & ALPHA_BIT) != 0, v);
    return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16);
  }

  // canvas pixel (X, Y): packed 0x00BBGGRR.  Night LocalView pixels come from `staged` (filled by the
  // noise pass), or 0xFFFFFFFF when there is no staging buffer and the noise pass stores them itself.
  __device__ uint32_t canvas_pixel(int X, int Y, int lw, int lh, int ih, const Lit& L, const uint8_t* staged) const {
    int vx = X - rt.border_x, vy = Y - rt.border_y;
    if (vx < 0 || vy < 0 || vx >= lw || vy >= lh + ih) return 0;   // untouched canvas (env.py:123)
    if (vy >= lh) return item_pixel(vx, vy, vy - lh);
    if (L.night) {
      if (!staged) return 0xFFFFFFFFu;
      const uint8_t* s3 = staged + 3 * (vx * lh + vy);
      return (uint32_t)s3[0] | ((uint32_t)s3[1] << 8) | ((uint32_t)s3[2] << 16);
    }
    if (lit) {   // sprite-free pixel: lit colour of (material, texel) was computed once for this frame
      int cm = colmap[vx], rm = rowmap[vy];
      int k = (cm & 0xFF) * e.cfg.local_gh + (rm & 0xFF);
      int32_t t = cell_tile[k];
      if (t >= 0 && cell_sprite[k] < 0) return lit[(t >> 24) * (rt.unit_x * rt.unit_y) + (cm >> 8) * rt.unit_y + (rm >> 8)];
    }
    int v[3];
    local_colour(vx, vy, v);
    return light(v, L, 0.0, 0.0);
  }

  __device__ void store_rgb(int X, int Y, uint32_t rgb) const {
    uint8_t* p = rt.out + ((size_t)Y * rt.size_w + X) * 3;
    p[0] = (uint8_t)rgb;
    p[1] = (uint8_t)(rgb >> 8);
    p[2] = (uint8_t)(rgb >> 16);
  }

  // Full frame.  pixels == false: only the RNG side effect of a night frame happens.
  //
  // Order matters for speed: on gfx9 a vector load issued after a store cannot be consumed before
  // that store has completed (one in-order vmcnt), so every global LOAD of the frame (texels, the
  // vignette) happens before the first global STORE.  At night the noise pass therefore does not
  // write pixels out; it stages them (3 bytes each) in the LDS that held the env's map copies --
  // dead once the per-frame tables are built -- and one write-out pass stores the whole frame.
  __device__ __forceinline__ void render(bool pixels) {
    const Config& c = e.cfg;
    W& w = e.w;
    int lw = c.local_gw * rt.unit_x, lh = c.local_gh * rt.unit_y;
    int ih = c.item_gh * rt.unit_y;
    Lit L;
    L.D = e.tb.daylight[e.rec->step];
    L.iD = 1 - L.D;
    L.night = L.D < 0.5;
    L.sleeping = e.rec->sleeping != 0;
    L.amount = 2 * (0.5 - L.D);
    int sw = rt.size_w, sh = rt.size_h;
    // staging buffer = the LDS copies of mat + objmap (contiguous, 3 * W * H bytes)
    uint8_t* staged = nullptr;
    if (pixels) {
      build_tables(L);
      if (L.night && 3 * lw * lh <= 3 * c.W * c.H && (uint8_t*)e.objmap == e.mat + align16(c.W * c.H)) staged = e.mat;
      if (prof && w.leader()) prof[7] = w.clock();
    }
    if (L.night) {
      // walk the MT19937 stream, 2 words per LocalView pixel, row-major over [x][y]
      int total = lw * lh;
      int words = 2 * total;
      int pos = e.mt_pos;
      int s_lo = 0;
      uint32_t carry = 0;
      uint32_t inv_lh = (uint32_t)(((1u << 24) + (uint32_t)lh - 1) / (uint32_t)lh);   // j / lh by multiplication, j < 2^16
      bool small = total < 65536;
      while (s_lo < words) {
        if (pos >= MT_N) {
          carry = e.mt[MT_N - 1];
          w.sync();
          if (w.wave0()) w.mt_twist(e.mt);
          w.sync();
          pos = 0;
        }
        int s_hi = s_lo + (MT_N - pos);
        if (s_hi > words) s_hi = words;
        if (pixels) {
          int j_first = s_lo >> 1;          // if s_lo is odd its first word is the carry
          int j_last = (s_hi - 2) >> 1;     // last double whose second word lies in this epoch
          int count = (s_hi >= 2) ? (j_last - j_first + 1) : 0;
          const uint32_t* mt = e.mt;
          w.block_for(count, [&](int q) {
            int j = j_first + q;
            int ia = 2 * j - s_lo;
            uint32_t a = (ia >= 0) ? mt[pos + ia] : carry;
            uint32_t b = mt[pos + ia + 1];
            double noise = 32.0 + 95.0 * mt_double(mt_temper(a), mt_temper(b));
            int x;
            if (small) {
              x = (int)(((uint32_t)j * inv_lh) >> 24);
              if (x * lh > j) x--;
              if ((x + 1) * lh <= j) x++;
            } else {
              x = j / lh;
            }
            int y = j - x * lh;
            int v[3];
            local_colour(x, y, v);
            double m = L.amount * rt.vignette[j];
            uint32_t rgb = light(v, L, m, noise);
            if (staged) {
              uint8_t* s3 = staged + 3 * j;
              s3[0] = (uint8_t)rgb;
              s3[1] = (uint8_t)(rgb >> 8);
              s3[2] = (uint8_t)(rgb >> 16);
            } else {
              store_rgb(x + rt.border_x, y + rt.border_y, rgb);
            }
          });
        }
        pos += s_hi - s_lo;
        s_lo = s_hi;
      }
      w.sync();
      e.mt_pos = pos;
    }
    if (prof && w.leader()) prof[8] = w.clock();
    if (!pixels) return;
    // Write-out.  Canvas is (size_w, size_h) in [x][y]; the output is its transpose (env.py:123-130).
    // Each lane owns runs of 4 consecutive pixels of an output row (one 12-byte store per run) and
    // evaluates up to 4 runs before it stores any of them.
    if ((sw & 3) == 0) {
      int qpr = sw >> 2, nquad = qpr * sh;
      int stride = w.nthreads();
      int q = w.tid();
      int Y = q / qpr, Xq = q - Y * qpr;
      int dY = stride / qpr, dX = stride - dY * qpr;
      while (q < nquad) {
        uint32_t px[4][4];
        int qx[4], qy[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          qx[r] = Xq;
          qy[r] = (q < nquad) ? Y : -1;
          if (q < nquad) {
#pragma unroll
            for (int u = 0; u < 4; u++) px[r][u] = canvas_pixel(4 * Xq + u, Y, lw, lh, ih, L, staged);
          }
          q += stride;
          Xq += dX;
          Y += dY;
          if (Xq >= qpr) {
            Xq -= qpr;
            Y++;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (qy[r] < 0) continue;
          const uint32_t* v = px[r];
          if (v[0] != 0xFFFFFFFFu && v[1] != 0xFFFFFFFFu && v[2] != 0xFFFFFFFFu && v[3] != 0xFFFFFFFFu) {
            uint32_t* p32 = (uint32_t*)(rt.out + ((size_t)qy[r] * sw + 4 * qx[r]) * 3);   // 12-byte aligned run
            p32[0] = (v[0] & 0xFFFFFFu) | (v[1] << 24);
            p32[1] = ((v[1] >> 8) & 0xFFFFu) | (v[2] << 16);
            p32[2] = ((v[2] >> 16) & 0xFFu) | (v[3] << 8);
          } else {
#pragma unroll
            for (int u = 0; u < 4; u++)
              if (v[u] != 0xFFFFFFFFu) store_rgb(4 * qx[r] + u, qy[r], v[u]);
          }
        }
      }
    } else {
      w.block_for(sw * sh, [&](int p) {
        int Y = p / sw, X = p - Y * sw;
        uint32_t v = canvas_pixel(X, Y, lw, lh, ih, L, staged);
        if (v != 0xFFFFFFFFu) store_rgb(X, Y, v);
      });
    }
  }
};

}  // namespace crafter
