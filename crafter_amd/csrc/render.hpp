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

// Row table: every LocalView pixel is one lookup table[row of its cell][texel].  Rows 0 .. kTileRows - 1
// hold the material tiles (raw RGBA texels, part of the static block), row kGrayRow the canvas fill of
// cells outside the map, and rows kSpriteRow0 .. are built per frame: one row per cell that shows a sprite
// (tile and sprite already alpha-blended, which does not depend on light or noise).  By day the rows in
// view are lit in place once per frame (daylight is one value per frame), at night they stay raw and
// every pixel is lit with its own noise.  Only when the table is small (unit 7: 34 x 49 x 4 B = 6.7 KB);
// big render sizes compute every pixel from the atlas.
constexpr int kTileRows = 13;      // material ids 0 (None) .. 12: data.yaml has 12 materials; LDS is too dear (every byte per
                                   // env counts against 4 step workgroups + a generator per CU) to reserve rows for
                                   // MAX_MATERIALS; crafter_upload_tables rejects rule sets with more
constexpr int kGrayRow = kTileRows;
constexpr int kSpriteRow0 = kTileRows + 1;
#ifndef CRAFTER_SPRITE_ROWS
#define CRAFTER_SPRITE_ROWS 8   // tests build the CPU harness with 1 to exercise the overflow path
#endif
constexpr int kSpriteRows = CRAFTER_SPRITE_ROWS;   // more sprite cells than this in one view: the extra ones take the generic path
__host__ __device__ __forceinline__ bool texel_rows_fit(const Config& c) {
  return (kSpriteRow0 + kSpriteRows) * c.unit_x * c.unit_y * 4 <= 8192;
}
__host__ __device__ __forceinline__ int texel_cache_bytes(const Config& c) {   // static rows (materials + gray)
  return texel_rows_fit(c) ? align16(kSpriteRow0 * c.unit_x * c.unit_y * 4) : 0;
}
__host__ __device__ __forceinline__ int sprite_rows_bytes(const Config& c) {   // per-frame rows, right behind them
  return texel_rows_fit(c) ? align16((kSpriteRow0 + kSpriteRows) * c.unit_x * c.unit_y * 4) - texel_cache_bytes(c) : 0;
}

// LDS copies of the small read-only tables (texture offsets, alpha flags, item positions): fetched
// together with the env state at kernel start so that building the per-frame tables never waits on
// a global load.
constexpr int RENDER_STATIC_BYTES = 4 * TEX_COUNT + 4 * MAX_ITEMS + 4 * 12 + 64 + 4 * 4 * MAX_ITEMS;   // 124+64+48+64+256 = 556 -> 560
static_assert(RENDER_STATIC_BYTES % 4 == 0, "alignment");

// The renderer's LDS region = per-frame tables, then one static block that is identical for every env
// and every step: pixel maps, texture offset tables and the raw material texels.
// The static block is built ONCE (Renderer::build_static, run when the tables are uploaded) into
// TablePtrs.render_static and staged in with a single 16-byte copy per step.
__host__ __device__ __forceinline__ int render_static_bytes(const Config& c) {
  int lw = c.local_gw * c.unit_x, vh = (c.local_gh + c.item_gh) * c.unit_y;
  return align16(2 * lw) + align16(2 * vh) + align16(RENDER_STATIC_BYTES) + texel_cache_bytes(c);
}
// Behind the static block in GLOBAL memory (never staged): every inventory slot's finished cell -- icon and count digit
// blended over the black canvas (engine.py:227-248) -- for each item and each digit it can show (0 = slot empty, 1..9,
// 10 = 'unknown'), unit_x * unit_y packed pixels each.  An ItemView pixel is then one coalesced load instead of two
// alpha blends; the cells are computed once, by the same blend code, when the tables are uploaded.
constexpr int kItemDigits = 11;
__host__ __device__ __forceinline__ int render_item_cells_bytes(const Config& c) { return MAX_ITEMS * kItemDigits * c.unit_x * c.unit_y * 4; }
// Behind those: the material rows of the row table LIT for every daytime step of an awake player -- what build_tables
// otherwise computes per frame with ~60 instructions (a dozen of them f64) per texel depends only on the step's daylight
// value and on whether the player sleeps (engine.py:198-202; under a random policy one frame in six is a sleeping one).
// The first kLitSteps steps are covered, awake and asleep (random-policy episodes end long before; beyond, the rows are lit on
// the fly); night steps keep raw rows and have no entry to speak of.
constexpr int kLitSteps = 1024;
__host__ __device__ __forceinline__ int render_lit_steps(const Config& c) {
  return texel_rows_fit(c) ? (c.n_daylight < kLitSteps ? c.n_daylight : kLitSteps) : 0;
}
// (one step's rows padded to whole 16-byte units -- texel_cache_bytes: Renderer::stage_rows copies them in 16-byte units)
__host__ __device__ __forceinline__ int render_lit_row_words(const Config& c) { return texel_cache_bytes(c) / 4; }
__host__ __device__ __forceinline__ int render_lit_bytes(const Config& c) { return 2 * render_lit_steps(c) * render_lit_row_words(c) * 4; }   // [awake, asleep][step]
// (Round 2 also kept the SPRITE rows finished per step -- sprite x material row x step, 79 MB.  Same-box A/B in round 3:
// 55.7 M env-steps/s with the table, 56.0 M without (sprite rows blended and lit per frame): dropped.)
// Last: one record per LocalView pixel in the order of the night noise stream (x-major, engine.py:208-209): the vignette
// value (engine.py:213-218) and the pixel's cell | texel << 8 -- one 16-byte load per night pixel instead of the vignette
// load plus a division, two map look-ups and the index arithmetic between them.
struct alignas(16) NightPx {
  double vignette;
  uint32_t desc, pad;
};
__host__ __device__ __forceinline__ int render_night_px_bytes(const Config& c) {
  return c.local_gw * c.unit_x * c.local_gh * c.unit_y * (int)sizeof(NightPx);
}
// Behind those (round 5): every SPRITE ROW the row table can be asked for, finished.  A cell that shows an object shows its
// sprite alpha-blended over the cell's material tile (engine.py:176-180): 49 texels that depend on (sprite, material) only --
// kSprites x kTileRows rows, 38 KB, built once with the same blend code -- and by day, lit, on (sprite, material, step,
// asleep) only: the first kLitSpriteSteps steps are kept lit (78 MB of 288 GB; the rows of a step sit side by side, a frame
// reads the <= 8 it shows).  build_tables used to blend (atlas fetch, seven /255 look-ups and ~40 f32 instructions per texel)
// and light (~60 instructions, a dozen of them f64) every sprite row of every frame, in two passes of the whole workgroup
// with a barrier each, on a kernel whose launch time follows its vector instruction count.  Round 3 dropped the
// same table at 47 % vector utilisation, where it bought nothing (see above).
constexpr int kSprites = TEX_COUNT - TEX_PLAYER_LEFT + 1;   // the object textures, and TEX_UNKNOWN last
__host__ __device__ __forceinline__ int sprite_index(int tex) { return tex >= TEX_PLAYER_LEFT ? tex - TEX_PLAYER_LEFT : kSprites - 1; }
__host__ __device__ __forceinline__ int sprite_tex_of(int index) { return index < kSprites - 1 ? TEX_PLAYER_LEFT + index : TEX_UNKNOWN; }
#ifndef CRAFTER_LIT_SPRITE_STEPS
#define CRAFTER_LIT_SPRITE_STEPS 1024   // (the CPU harness of tests/hostsim builds with fewer: one host thread lights the table there)
#endif
constexpr int kLitSpriteSteps = CRAFTER_LIT_SPRITE_STEPS;
__host__ __device__ __forceinline__ int render_blended_rows(const Config& c) { return texel_rows_fit(c) ? kSprites * kTileRows : 0; }
__host__ __device__ __forceinline__ int render_blended_bytes(const Config& c) { return render_blended_rows(c) * c.unit_x * c.unit_y * 4; }
__host__ __device__ __forceinline__ int render_lit_sprite_steps(const Config& c) {
  if (!c.render_obs) return 0;   // a handle that never draws observations keeps no 78 MB of lit sprite rows: its rare
                                 // Env.render() lights the rows it loads, like a step beyond the table
  int s = render_lit_steps(c);
  return s < kLitSpriteSteps ? s : kLitSpriteSteps;
}
__host__ __device__ __forceinline__ size_t render_lit_sprite_bytes(const Config& c) {   // [awake, asleep][step][sprite][material][texel]
  return (size_t)2 * render_lit_sprite_steps(c) * render_blended_bytes(c);
}
__host__ __device__ __forceinline__ size_t render_static_total_bytes(const Config& c) {
  return (size_t)render_static_bytes(c) + render_item_cells_bytes(c) + render_lit_bytes(c) + render_night_px_bytes(c) +
         render_blended_bytes(c) + render_lit_sprite_bytes(c);
}
__host__ __device__ __forceinline__ int render_frame_bytes(const Config& c) {   // tables rebuilt every frame
  int ncell = c.local_gw * c.local_gh;
  return 16 + align16(8 * ncell) + MAX_ITEMS * 32 + 2 * align16(ncell) + 16 + 32;
}
__host__ __device__ __forceinline__ int render_lds_bytes(const Config& c) {   // frame tables | static block | sprite rows
  return render_frame_bytes(c) + render_static_bytes(c) + sprite_rows_bytes(c);
}

// n / d for 0 <= n < 2^16 with 24-bit multiplications (full rate; an integer division is ~40
// instructions): floor(n * inv / 2^16) with 2^16 / d <= inv <= 2^16 / d + 1 is the quotient or one more
// (the excess n * (inv - 2^16 / d) / 2^16 is < 1), and n * inv < 2^31 for d >= 4.
template <class W>
struct SmallDiv {
  int d, inv;
  bool ok;
  // any inv in [2^16 / d, 2^16 / d + 1] will do: a float division, not a 40-instruction integer one
  __device__ __forceinline__ SmallDiv(int d_, int max_n) : d(d_), inv((int)(65536.0f / (float)d_) + 1), ok(d_ >= 4 && max_n < 65536) {}
  __device__ __forceinline__ int div(int n) const {
    if (!ok) return n / d;
    int q = W::mul24(n, inv) >> 16;
    return W::mul24(q, d) > n ? q - 1 : q;
  }
  __device__ __forceinline__ int mul(int q) const { return W::mul24(q, d); }
};

// Frame record of a split step (env_kernels.hpp): everything the frame of one env-step depends on.
//   [0, 63) material id per view cell (0xFF: outside the map)   [63] 1: no frame this step (env handed to the regeneration kernel)
//   [64, 127) sprite texture id per view cell (0xFF: none)      [127] player asleep
//   [128, 144) inventory   [144, 152) daylight of the step (f64)   [152, 156) step   [156, 160) MT19937 stream position   [160, 164) env
// Night noise generated ahead of the rules (env_kernels.hpp noise_chain): this many consecutive MT19937 states of the env wait
// in its global scratch.  12: the rules' own draws may have entered the second state (stream position up to 1247), and
// 1247 + 2 x 63 x 49 < 12 x 624.
constexpr int kNoiseStates = 12;
constexpr int kFrameRecordBytes = 192;
constexpr int kFrameSprites = 64;
constexpr int kFrameFlag = 63;
constexpr int kFrameSleeping = 127;
constexpr int kFrameInventory = 128;
constexpr int kFrameDaylight = 144;
constexpr int kFrameStep = 152;
constexpr int kFrameMtPos = 156;
constexpr int kFrameEnv = 160;     // pipelined step kernel: the env the record belongs to (the frame group learns it from the hand-off)

// The texture an object shows (objects.py:85-93,271,291,323,361-367,395-399); sleeping: the player's state.
__device__ __forceinline__ int sprite_texture(const Obj& o, bool sleeping) {
  int f = (o.fx < 0) ? 0 : (o.fx > 0) ? 1 : (o.fy < 0) ? 2 : 3;
  switch (o.type) {
    case T_PLAYER: return sleeping ? TEX_PLAYER_SLEEP : TEX_PLAYER_LEFT + f;
    case T_COW: return TEX_COW;
    case T_ZOMBIE: return TEX_ZOMBIE;
    case T_SKELETON: return TEX_SKELETON;
    case T_ARROW: return TEX_ARROW_LEFT + f;
    case T_PLANT: return o.aux > 300 ? TEX_PLANT_RIPE : TEX_PLANT;
  }
  return TEX_UNKNOWN;
}

template <class W, class SlotT = uint16_t>
struct Renderer {
  Env<W, SlotT>& e;
  const RenderTarget& rt;
  uint32_t* hdr;         // LDS [4]: -, #sprite cells, #non-empty item slots, -
  uint8_t* present;      // LDS [32]: material m shows in the view (plain stores: same-address LDS atomics serialise, ~100 clk each)
  int32_t* cell_tile;    // LDS [ncell] atlas byte offset of the cell's material texture | material << 24, -1 outside the map
  int32_t* cell_sprite;  // LDS [ncell] atlas byte offset of the cell's sprite | ALPHA_BIT, -1 if none
  uint16_t* colmap;      // LDS [local_w]          view x pixel -> cell column | texel x << 8
  uint16_t* rowmap;      // LDS [local_h + item_h] view y pixel -> cell row | texel y << 8 (item rows restart at 0)
  int32_t* item_tab;     // LDS [MAX_ITEMS][8] icon off|ALPHA, digit off|ALPHA, icon x,y, digit x,y, amount, -
  uint8_t* sprite_list;  // LDS [ncell] cells that show a sprite
  uint8_t* cell_row;     // LDS [ncell] row of the cell in the texel table (material, kGrayRow or a sprite row)
  uint8_t* slot_list;    // LDS [MAX_ITEMS] inventory slots with amount >= 1
  uint8_t* sprite_src;   // LDS [kSpriteRows] (the upper half of `present`): sprite row s of the table shows row sprite_src[s] of the blended / lit sprite rows
  int32_t* s_tex_tile;   // LDS copies of TablePtrs.tex_tile / tex_icon / tex_digit / tex_alpha / item_pos
  int32_t* s_tex_icon;
  int32_t* s_tex_digit;
  uint8_t* s_tex_alpha;
  int32_t* s_item_pos;
  const float* div255;   // TablePtrs.unit255 in global memory (the alpha blend's only division).  Until round 6 a 1 KB LDS copy inside the static
                         // block -- read only by frames that blend per pixel (other image sizes, a view with more sprites than the row table holds):
                         // the default instance's frames take their sprite rows blended from tables, and its workgroup needed the LDS (env_kernels.hpp lds_layout)
  uint32_t* cache;       // LDS [kSpriteRow0 + kSpriteRows][unit_x * unit_y]: the row table (lit by day, raw at night), or null
  uint32_t* mtb;         // LDS [624] second MT19937 state buffer (shared with the worldgen scratch), or null
  uint32_t* pix;         // [local_w * local_h]: a night frame's LocalView pixels in noise-stream order, or null.  LDS -- or, for
                         //   the frame kernel of the split step, the env's scratch in global memory (pix_global): a night frame's
                         //   12 KB then do not count against the workgroups per CU of the 86 % of frames that are day frames
  bool pix_global = false;
  const uint32_t* noise_raw = nullptr;   // global [kNoiseStates][624]: the states this frame's noise comes from, generated ahead
                                         // (env_kernels.hpp noise_chain), or null: noise_pass regenerates them in the frame
  int noise_base = 0;                    // index of the noise's first word in them
  uint64_t* prof = nullptr;  // optional shader-clock stamps (slots 7, 8)
  const uint8_t* frame_cells = nullptr;   // LDS: the frame record of a split step (env_kernels.hpp) -- the cell table's input instead of the maps

  uint8_t* static_base;  // LDS: start of the static block (render_static_bytes)
  // What stage_rows put into the material rows of the row table: the rows LIT for (rows_step, rows_sleeping), or (-1) the raw
  // texels of the static block.
  int rows_step = -1;
  bool rows_sleeping = false;

  static constexpr int32_t ALPHA_BIT = 1 << 30;
  static constexpr int32_t OFF_MASK = (1 << 24) - 1;
  static constexpr int SPRITE_SHIFT = 24;   // cell_sprite bits 24..29: the sprite's texture id

  __device__ __forceinline__ Renderer(Env<W, SlotT>& env, const RenderTarget& t, uint8_t* lds, uint32_t* second_mt_state, uint8_t* frame_lds)
      : e(env), rt(t) {
    const Config& c = e.cfg;
    int ncell = c.local_gw * c.local_gh;
    int lw = c.local_gw * c.unit_x, vh = (c.local_gh + c.item_gh) * c.unit_y;
    hdr = (uint32_t*)lds;
    lds += 16;
    cell_tile = (int32_t*)lds;
    cell_sprite = cell_tile + ncell;
    lds += align16(8 * ncell);
    item_tab = (int32_t*)lds;
    lds += MAX_ITEMS * 32;
    sprite_list = lds;
    lds += align16(ncell);
    cell_row = lds;
    lds += align16(ncell);
    slot_list = lds;
    lds += 16;
    present = lds;
    sprite_src = lds + 16;
    static_assert(kTileRows <= 16 && kSpriteRows <= 16, "present[] and sprite_src[] share 32 bytes");
    lds += 32;
    bind_static(lds);
    mtb = second_mt_state;
    pix = (uint32_t*)frame_lds;
  }

  // pointers into a static block at `p` (LDS copy, or the global buffer build_static fills)
  __device__ __forceinline__ void bind_static(uint8_t* p) {
    const Config& c = e.cfg;
    int lw = c.local_gw * c.unit_x, vh = (c.local_gh + c.item_gh) * c.unit_y;
    static_base = p;
    colmap = (uint16_t*)p;
    p += align16(2 * lw);
    rowmap = (uint16_t*)p;
    p += align16(2 * vh);
    s_tex_tile = (int32_t*)p;
    s_tex_icon = s_tex_tile + TEX_COUNT;
    s_tex_digit = s_tex_icon + MAX_ITEMS;
    s_tex_alpha = (uint8_t*)(s_tex_digit + 12);
    s_item_pos = (int32_t*)(s_tex_alpha + 64);
    p += align16(RENDER_STATIC_BYTES);
    div255 = e.tb.unit255;
    cache = texel_cache_bytes(c) ? (uint32_t*)p : nullptr;
  }

  // objects.py:85-93,271,291,323,361-367,395-399
  __device__ __forceinline__ int sprite_of(const Obj& o) const { return sprite_texture(o, e.rec->sleeping != 0); }

  struct Lit {
    double D, iD, hD, amount;   // daylight, 1 - daylight, (1 - daylight) * 0.5, 2 * (0.5 - daylight)
    bool night, sleeping;
  };

  // The lit material rows of a day step (render_lit_bytes) sit in global memory (5.6 MB, L2 / MALL resident).  Round 2
  // fetched them into registers right after stage-in, a rule phase before their use; round 3 measured the plain fetch at
  // the point of use as fast (56.0 vs 55.7 M env-steps/s) -- and the early fetch, moved behind the sprite blend, went wrong
  // under load (day frames next to night frames diverged, never reproduced on the CPU harness): removed.
  __device__ __forceinline__ const uint32_t* lit_rows(int step, bool sleeping) const {
    const Config& c = e.cfg;
    return (const uint32_t*)(e.tb.render_static + render_static_bytes(c) + render_item_cells_bytes(c)) +
           ((size_t)(sleeping ? render_lit_steps(c) : 0) + step) * render_lit_row_words(c);
  }
  // the blended sprite rows, raw, and those of a day step, lit (render_blended_bytes / render_lit_sprite_bytes)
  __device__ __forceinline__ const uint32_t* blended_rows() const {
    const Config& c = e.cfg;
    return (const uint32_t*)(e.tb.render_static + render_static_bytes(c) + render_item_cells_bytes(c) + render_lit_bytes(c) + render_night_px_bytes(c));
  }
  __device__ __forceinline__ const uint32_t* lit_sprite_rows(int step, bool sleeping) const {
    const Config& c = e.cfg;
    return blended_rows() + (size_t)render_blended_rows(c) * c.unit_x * c.unit_y * (1 + (size_t)(sleeping ? render_lit_sprite_steps(c) : 0) + step);
  }
  // Fills the static block at `dst` (global memory; one workgroup, once per table upload).
  __device__ __forceinline__ void build_static(uint8_t* dst) {
    const Config& c = e.cfg;
    W& w = e.w;
    bind_static(dst);
    int lw = c.local_gw * rt.unit_x, lh = c.local_gh * rt.unit_y, ih = c.item_gh * rt.unit_y;
    w.block_for(render_static_bytes(c) / 4, [&](int i) { ((uint32_t*)dst)[i] = 0; });
    w.sync();
    w.block_for(TEX_COUNT, [&](int i) { s_tex_tile[i] = rt.tex_tile[i]; });
    w.block_for(MAX_ITEMS, [&](int i) { s_tex_icon[i] = rt.tex_icon[i]; });
    w.block_for(11, [&](int i) { s_tex_digit[i] = rt.tex_digit[i]; });
    w.block_for(TEX_COUNT + MAX_ITEMS + 11, [&](int i) { s_tex_alpha[i] = e.tb.tex_alpha[i]; });
    w.block_for(4 * MAX_ITEMS, [&](int i) { s_item_pos[i] = rt.item_pos[i]; });
    w.block_for(lw, [&](int x) {
      int g = x / rt.unit_x;
      colmap[x] = (uint16_t)(g | ((x - g * rt.unit_x) << 8));
    });
    w.block_for(lh + ih, [&](int y) {
      int yy = y < lh ? y : y - lh;
      int g = yy / rt.unit_y;
      rowmap[y] = (uint16_t)(g | ((yy - g * rt.unit_y) << 8));
    });
    if (cache) {   // raw texels of every material's tile (which ones are in view is a per-frame matter)
      int ntex = rt.unit_x * rt.unit_y;
      w.block_for((e.R.n_materials + 1) * ntex, [&](int i) {
        int m = i / ntex, texel = i - m * ntex;
        int32_t off = rt.tex_tile[TEX_MATERIAL0 + m];
        cache[i] = off >= 0 ? *(const uint32_t*)(rt.atlas + off + texel * 4) : 0u;
      });
      w.block_for(ntex, [&](int i) { cache[kGrayRow * ntex + i] = 0x7F7F7F7Fu; });   // canvas fill, engine.py:167
    }
    w.sync();
    {   // the inventory cells (render_item_cells_bytes), right behind the block
      uint32_t* cells = (uint32_t*)(dst + render_static_bytes(c));
      int ntex = rt.unit_x * rt.unit_y;
      w.block_for(MAX_ITEMS * kItemDigits * ntex, [&](int i) {
        int kd = i / ntex, tex = i - kd * ntex;
        int k = kd / kItemDigits, d = kd - k * kItemDigits;
        uint32_t px = 0;
        if (d >= 1 && k < e.R.n_items) {
          int tx = tex / rt.unit_y, ty = tex - tx * rt.unit_y;
          int cy = k / c.item_gw, cx = k - cy * c.item_gw;
          int vx = cx * rt.unit_x + tx, iy = cy * rt.unit_y + ty;
          int v[3] = {0, 0, 0};
          int ix = vx - rt.item_pos[k * 4 + 0], iyy = iy - rt.item_pos[k * 4 + 1];
          if (ix >= 0 && iyy >= 0 && ix < rt.icon_w && iyy < rt.icon_h)
            blend(*(const uint32_t*)(rt.atlas + rt.tex_icon[k] + (ix * rt.icon_h + iyy) * 4), e.tb.tex_alpha[TEX_COUNT + k] != 0, v);
          int dx = vx - rt.item_pos[k * 4 + 2], dy = iy - rt.item_pos[k * 4 + 3];
          if (dx >= 0 && dy >= 0 && dx < rt.digit_w && dy < rt.digit_h)
            blend(*(const uint32_t*)(rt.atlas + rt.tex_digit[d] + (dx * rt.digit_h + dy) * 4), e.tb.tex_alpha[TEX_COUNT + MAX_ITEMS + d] != 0, v);
          px = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16);
        }
        cells[i] = px;
      });
      w.sync();
    }
    if (cache) {   // the lit material rows (render_lit_bytes), behind the inventory cells; cache = the raw rows just built
      uint32_t* lit = (uint32_t*)(dst + render_static_bytes(c) + render_item_cells_bytes(c));
      int words = render_lit_row_words(c);
      int steps = render_lit_steps(c);
      w.block_for(2 * steps * words, [&](int i) {
        int hs = i / words, j = i - hs * words;
        int step = hs < steps ? hs : hs - steps;
        Lit L;
        L.D = e.tb.daylight[step];
        L.iD = 1 - L.D;
        L.hD = L.iD * 0.5;
        L.night = L.D < 0.5;
        L.sleeping = hs >= steps;
        L.amount = 2 * (0.5 - L.D);
        uint32_t tile = j < kSpriteRow0 * rt.unit_x * rt.unit_y ? cache[j] : 0u;   // (behind the last row: padding)
        int v[3] = {(int)(tile & 0xFF), (int)((tile >> 8) & 0xFF), (int)((tile >> 16) & 0xFF)};
        lit[i] = (L.night || j >= kSpriteRow0 * rt.unit_x * rt.unit_y) ? tile : light(v, L, 0.0, 0.0);
      });
      w.sync();
    }
    {   // the night pixel records (render_night_px_bytes), last
      NightPx* npx = (NightPx*)(dst + render_static_bytes(c) + render_item_cells_bytes(c) + render_lit_bytes(c));
      w.block_for(lw * lh, [&](int j) {
        int x = j / lh, y = j - x * lh;
        int cm = colmap[x], rm = rowmap[y];
        NightPx p;
        p.vignette = rt.vignette[j];
        p.desc = (uint32_t)((cm & 0xFF) * c.local_gh + (rm & 0xFF)) | (uint32_t)((cm >> 8) * rt.unit_y + (rm >> 8)) << 8;
        p.pad = 0;
        npx[j] = p;
      });
      w.sync();
    }
    if (cache) {   // the blended sprite rows (render_blended_bytes), behind them: sprite over the RAW tile, as build_tables blended them per frame
      uint32_t* bl = (uint32_t*)(dst + render_static_bytes(c) + render_item_cells_bytes(c) + render_lit_bytes(c) + render_night_px_bytes(c));
      int ntex = rt.unit_x * rt.unit_y;
      w.block_for(kSprites * kTileRows * ntex, [&](int i) {
        int sm = i / ntex, tex = i - sm * ntex;
        int sidx = sm / kTileRows, m = sm - sidx * kTileRows;
        int sp = sprite_tex_of(sidx);
        uint32_t tile = cache[m * ntex + tex];
        int v[3] = {(int)(tile & 0xFF), (int)((tile >> 8) & 0xFF), (int)((tile >> 16) & 0xFF)};
        blend(*(const uint32_t*)(rt.atlas + rt.tex_tile[sp] + tex * 4), e.tb.tex_alpha[sp] != 0, v);
        bl[i] = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16);
      });
      w.sync();
    }
  }

  // The lit sprite rows (render_lit_sprite_bytes) from the blended ones build_static has left in `dst`: part `part` of `nparts`
  // (one workgroup each; the device launches 2 x steps of them) lights the rows of every nparts-th (asleep, step).
  __device__ __forceinline__ void build_lit_sprites(uint8_t* dst, int part, int nparts) {
    const Config& c = e.cfg;
    W& w = e.w;
    int steps = render_lit_sprite_steps(c);
    int words = render_blended_rows(c) * rt.unit_x * rt.unit_y;
    if (!steps || !words) return;
    const uint32_t* bl = (const uint32_t*)(dst + render_static_bytes(c) + render_item_cells_bytes(c) + render_lit_bytes(c) + render_night_px_bytes(c));
    uint32_t* lit = (uint32_t*)bl + words;
    for (int hs = part; hs < 2 * steps; hs += nparts) {
      int step = hs < steps ? hs : hs - steps;
      Lit L;
      L.D = e.tb.daylight[step];
      L.iD = 1 - L.D;
      L.hD = L.iD * 0.5;
      L.night = L.D < 0.5;
      L.sleeping = hs >= steps;
      L.amount = 2 * (0.5 - L.D);
      uint32_t* out = lit + (size_t)hs * words;
      w.block_for(words, [&](int j) {
        uint32_t tile = bl[j];
        int v[3] = {(int)(tile & 0xFF), (int)((tile >> 8) & 0xFF), (int)((tile >> 16) & 0xFF)};
        out[j] = L.night ? tile : light(v, L, 0.0, 0.0);   // (a night frame takes the raw rows: every pixel has its own noise)
      });
    }
  }

  // Static block -> LDS in the two phases of stage_issue / stage_commit (env_core.hpp), so that the caller
  // can put the env state's loads in flight in between.  No barrier here: the caller's next workgroup
  // barrier covers it.
  struct Preload {
    vec16 blk[2];   // 2 x 16 B per thread covers the default 5.2 KB block with 256 threads
  };
  // rows false: everything but the raw material rows at the block's end -- stage_rows brings those, raw or lit
  __device__ __forceinline__ int static_chunks(bool rows) const { return (render_static_bytes(e.cfg) - (rows ? 0 : texel_cache_bytes(e.cfg))) / 16; }
  __device__ __forceinline__ void preload_issue(Preload& q, bool rows = true) {
    stage_issue(e.w, q.blk, (const vec16*)e.tb.render_static, static_chunks(rows));
  }
  __device__ __forceinline__ void preload_commit(const Preload& q, bool rows = true) {
    W& w = e.w;
    w.block_for(8, [&](int i) { ((uint32_t*)present)[i] = 0; });
    stage_commit(w, q.blk, (vec16*)static_base, (const vec16*)e.tb.render_static, static_chunks(rows));
    if (rows) rows_step = -1;
  }
  __device__ __forceinline__ void preload() {
    Preload q;
    preload_issue(q);
    preload_commit(q);
  }
  // ... by the waves behind the first one only (resident steps, env_kernels.hpp: the rule wave is already running; nothing
  // reads the block before the frame's own barriers)
  __device__ __forceinline__ void preload_beside(bool rows = true) {
    W& w = e.w;
    const vec16* src = (const vec16*)e.tb.render_static;
    vec16* dst = (vec16*)static_base;
    w.consumer_for(static_chunks(rows), [&](int i) { dst[i] = src[i]; });
    w.consumer_for(8, [&](int i) { ((uint32_t*)present)[i] = 0; });
    if (rows) rows_step = -1;
  }
  // The material rows of the row table for the frame of step `step` (the step about to run: the caller knows its number and
  // its daylight before the rules have run): by day the rows LIT for that step -- render_lit_bytes; for a player who is
  // asleep now if `sleeping`: the rules may wake him or put him to sleep, or end the episode; build_tables checks -- straight
  // from the table into the place of the raw rows, which then need no pass of their own in the frame; at night, and beyond
  // the table, the raw rows.  By the waves behind the first one, while the rules run (W::consumers: the first wave does not
  // come here -- it must not wait for `daylight`, a load the caller issued a moment ago, at the head of its rule phase);
  // nothing reads the rows before the frame's own barriers, and nobody else writes them (preload_* with rows = false).
  // rows_staged: what that was, for every wave, when the frame begins.
  __device__ __forceinline__ void rows_staged(int step, double daylight, bool sleeping) {
    if (!cache) return;
    rows_step = (daylight >= 0.5 && step < render_lit_steps(e.cfg)) ? step : -1;
    rows_sleeping = sleeping;
  }
  __device__ __forceinline__ void stage_rows(int step, double daylight, bool sleeping) {
    const Config& c = e.cfg;
    W& w = e.w;
    if (!cache) return;
    const bool lit = daylight >= 0.5 && step < render_lit_steps(c);
    const vec16* src = lit ? (const vec16*)lit_rows(step, sleeping) : (const vec16*)(e.tb.render_static + render_static_bytes(c) - texel_cache_bytes(c));
    // (whole 16-byte units, then the odd words: the row table's sprite rows begin right behind the last material texel, and
    // they have another writer -- build_tables)
    const int words = kSpriteRow0 * rt.unit_x * rt.unit_y;
    vec16* dst = (vec16*)cache;
    w.consumer_for(words / 4, [&](int i) { dst[i] = src[i]; });
    w.consumer_for(words & 3, [&](int i) { cache[(words & ~3) + i] = ((const uint32_t*)src)[(words & ~3) + i]; });
  }

  // ---- Early frame (round 6): the part of a day frame that does not depend on the objects, drawn WHILE the object loop runs.
  // A step's chain used to be rules (one wave, ~12 k clocks, three waves idle) and then the frame (cell table 4.5 k on one
  // wave, pixels 3.1 k on four).  But once Player.update has run (2.5 k clocks into the rules; env.py:86) the player's
  // position, whether he sleeps and every material of the map are final for this step -- what the objects still change is
  // which SPRITES stand where (and, an arrow that breaks something, a material: the rules then flag the frame, and it is
  // drawn again from scratch).  So the waves behind the first one, which have nothing to do until the rules end, wait for
  // that moment (an LDS word the rule wave sets), build the MATERIAL half of the cell table and draw every LocalView pixel
  // from it.  Behind the rules the frame is then: find the sprite cells (<= 8), fetch their rows, and draw the quads that
  // touch those cells over what is there (finish_frame) -- plus the inventory strip, whose health digit the objects may
  // still have changed.
  // hdr words while the rules run: [0] set by the rule wave when the player has moved, [2] counts the waves whose material
  // rows are staged (+ 1 for the cell table), [3] counts the waves whose share of the pixels is drawn.
  __device__ __forceinline__ bool quad_geometry(int& KRP) const {
    const Config& c = e.cfg;
    constexpr int NT = W::kThreads;
    constexpr int NP = NT > 64 ? NT - 64 : NT;   // the threads that draw early
    int lw = c.local_gw * rt.unit_x, lh = c.local_gh * rt.unit_y, ih = c.item_gh * rt.unit_y;
    int sw = rt.size_w, sh = rt.size_h, gpr = sw >> 2;
    KRP = 5;
    return cache != nullptr && pix != nullptr && !pix_global && rt.border_x == 0 && rt.border_y == 0 && (sw & 3) == 0 && lw <= sw && gpr > 0 &&
           NP % gpr == 0 && NT % gpr == 0 && lh <= 5 * (NP / gpr) && sh - lh - ih >= 0 && c.item_gw == c.local_gw && c.local_gw * c.local_gh <= 64 &&
           rt.unit_y * ((rt.unit_x + 6) / 4 + 1) * kSpriteRows <= NT && gpr * ih <= 2 * NP;
  }
  // by the waves behind the first one (device), or inline at the rule wave's signal (the CPU harness)
  __device__ __forceinline__ void early_frame(int step, double daylight, bool staged_sleeping) {
    const Config& c = e.cfg;
    W& w = e.w;
    int KRP;
    if (!(daylight >= 0.5) || step >= render_lit_steps(c) || !quad_geometry(KRP)) return;   // (decided before any waiting: a night frame waits for nothing)
    if (w.lane() == 0) w.lds_inc(&hdr[2]);                       // this wave's material rows are in the table (stage_rows: DS operations complete in order)
    w.spin_until(&hdr[0], 1u);                                   // Player.update has run
    if ((e.rec->sleeping != 0) != staged_sleeping || e.rec->step != step) return;   // rows of the other state: the frame is drawn the old way
    constexpr int NT = W::kThreads;
    constexpr int NP = NT > 64 ? NT - 64 : NT;
    constexpr int T0 = NT > 64 ? 64 : 0;
    const int ncell = c.local_gw * c.local_gh;
    const int ntex = rt.unit_x * rt.unit_y;
    const int lw = c.local_gw * rt.unit_x, lh = c.local_gh * rt.unit_y;
    const int sw = rt.size_w, gpr = sw >> 2;
    if (w.wave_is(NT > 64 ? 1 : 0)) {   // the material half of the cell table, one lane per cell
      Obj p = e.obj_rd_lane(1);
      int offx = c.local_gw / 2, offy = c.local_gh / 2;
      SmallDiv<W> by_gh(c.local_gh, ncell);
      w.lanes(0, ncell, [&](int k, int) {
        int gx = by_gh.div(k), gy = k - by_gh.mul(gx);
        int wx = (int)p.x + gx - offx, wy = (int)p.y + gy - offy;
        int32_t t = -1;
        if (e.inside(wx, wy)) {
          int m = e.mat_at(wx, wy);
          t = s_tex_tile[TEX_MATERIAL0 + m] | (m << 24);
          present[m] = 1;
        }
        cell_tile[k] = t;
        cell_sprite[k] = -1;
        cell_row[k] = (uint8_t)(t >= 0 ? (t >> 24) : kGrayRow);
      });
      if (w.lane() == 0) w.lds_inc(&hdr[2]);
    }
    w.spin_until(&hdr[2], (uint32_t)(W::kDrawingWaves + 1));     // every drawing wave's rows + the cell table
    const int row_bytes = 3 * sw;
    const int rows_per = NP / gpr;
    SmallDiv<W> by_gpr(gpr, NT);
    struct Px4 { uint32_t a, b, c; };
    w.each_thread([&](int tid) {
      if (tid < T0) return;
      int t = tid - T0;
      int y0 = by_gpr.div(t), g = t - by_gpr.mul(y0);
      int cm[4];
      bool in[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int x = 4 * g + k;
        in[k] = x < lw;
        cm[k] = colmap[in[k] ? x : lw - 1];
      }
      constexpr int KR = 5;
      int yy[KR], rbase[KR], ty[KR], row[KR][4];
      uint32_t px[KR][4];
#pragma unroll
      for (int r = 0; r < KR; r++) {
        yy[r] = y0 + r * rows_per;
        int rm = rowmap[yy[r] < lh ? yy[r] : lh - 1];
        rbase[r] = rm & 0xFF;
        ty[r] = rm >> 8;
      }
#pragma unroll
      for (int r = 0; r < KR; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) row[r][k] = cell_row[W::mul24(cm[k] & 0xFF, c.local_gh) + rbase[r]];
#pragma unroll
      for (int r = 0; r < KR; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) px[r][k] = cache[W::mul24(row[r][k], ntex) + W::mul24(cm[k] >> 8, rt.unit_y) + ty[r]];
#pragma unroll
      for (int r = 0; r < KR; r++) {
#pragma unroll
        for (int k = 0; k < 4; k++) px[r][k] = in[k] ? (px[r][k] & 0xFFFFFFu) : 0u;
        Px4 v = {px[r][0] | (px[r][1] << 24), (px[r][1] >> 8) | (px[r][2] << 16), (px[r][2] >> 16) | (px[r][3] << 8)};
        if (yy[r] < lh) *(Px4*)(rt.out + W::mul24(yy[r], row_bytes) + 12 * g) = v;
      }
    });
    if (w.lane() == 0) w.lds_inc(&hdr[3]);
  }
  // did every drawing wave finish an early frame of this step? (all waves, behind the rules' barrier)
  __device__ __forceinline__ bool early_frame_drawn() const {
    return hdr[3] == (uint32_t)W::kDrawingWaves;
  }

  // The rest of a day frame whose material pixels early_frame drew: the sprite cells and the inventory strip.  All waves.
  // Returns false if the view shows more sprite cells than the row table holds: the caller draws the frame the old way.
  __device__ __forceinline__ bool finish_frame(const Lit& L) {
    const Config& c = e.cfg;
    W& w = e.w;
    constexpr int NT = W::kThreads;
    const int ncell = c.local_gw * c.local_gh;
    const int ntex = rt.unit_x * rt.unit_y;
    const int lw = c.local_gw * rt.unit_x, lh = c.local_gh * rt.unit_y, ih = c.item_gh * rt.unit_y;
    const int sw = rt.size_w, sh = rt.size_h, gpr = sw >> 2;
    const int item_quads = gpr * ih, tail_quads = gpr * (sh - lh - ih);
    const int row_bytes = 3 * sw;
    SmallDiv<W> by_gpr(gpr, NT);
    struct Px4 { uint32_t a, b, c; };
    // the inventory quads' texels: loads issued now, by the waves behind the first one (as in render)
    constexpr int KI = 2;
    constexpr int kItemOwners = NT > 64 ? NT - 64 : NT;
    constexpr int kItemFirst = NT > 64 ? 64 : 0;
    struct ItemQuad {
      uint32_t px[KI][4];
      bool show[KI][4];
    };
    ItemQuad item_quad[W::kThreadSlots];
    const bool items_ok = item_quads <= KI * kItemOwners && kItemOwners % gpr == 0;
    if (items_ok) {
      const uint32_t* item_cells = (const uint32_t*)(e.tb.render_static + render_static_bytes(c));
      w.each_thread([&](int tid) {
        if (!W::uni((int)(tid >= kItemFirst))) return;
        ItemQuad& iq = item_quad[W::thread_slot(tid)];
        int t0 = tid - kItemFirst;
        int y0 = by_gpr.div(t0), g = t0 - by_gpr.mul(y0);
#pragma unroll
        for (int s_ = 0; s_ < KI; s_++) {
          int iy = y0 + s_ * (kItemOwners / gpr);
          bool mine = W::mul24(iy, gpr) + g < item_quads;
          int rm = rowmap[lh + (mine ? iy : 0)];
          int cy = rm & 0xFF, ty = rm >> 8;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            int x = 4 * g + k;
            int cm = colmap[x < lw ? x : lw - 1];
            int slot = W::mul24(cy, c.item_gw) + (cm & 0xFF);
            bool has = mine && x < lw && slot < e.R.n_items;
            int amount = e.rec->inv[has ? slot : 0];
            iq.show[s_][k] = has && amount >= 1;
            int d = amount <= 9 ? amount : 10;
            int at = W::mul24(W::mul24(slot, kItemDigits) + d, ntex) + W::mul24(cm >> 8, rt.unit_y) + ty;
            iq.px[s_][k] = item_cells[iq.show[s_][k] ? at : 0];
          }
        }
      });
    }
    if (w.wave_is(0)) {
      // the sprite cells: one lane per cell looks for an object on it; their rows' loads leave at once
      Obj p = e.obj_rd_lane(1);
      int offx = c.local_gw / 2, offy = c.local_gh / 2;
      SmallDiv<W> by_gh(c.local_gh, ncell);
      w.lane_set(0, 0, ncell, [&](int k, int) -> uint32_t {
        int gx = by_gh.div(k), gy = k - by_gh.mul(gx);
        int wx = (int)p.x + gx - offx, wy = (int)p.y + gy - offy;
        int m = 0xFF, sp = 0xFF;
        if constexpr (!Env<W, SlotT>::kLane) {
          if (e.inside(wx, wy)) {
            m = e.mat_at(wx, wy);
            int slot = e.slot_lane(wx, wy);
            if (slot) sp = sprite_of(e.obj_rd_lane(slot));
          }
        }
        return (uint32_t)(m | (sp << 8));
      });
      const uint64_t smask = W::uni64(w.ballot(0, ncell, [&](int k) {
        uint32_t v = w.lane_get(0, k);
        return (v & 0xFFu) != 0xFFu && (v >> 8) != 0xFFu;
      }));
      const int out = __builtin_popcountll(smask);
      if (w.leader()) hdr[1] = (uint32_t)out;
      if (out <= kSpriteRows) {
        const bool lit_here = e.rec->step >= render_lit_sprite_steps(c);   // beyond the lit sprite table: the raw blended rows, lit as they are placed
        const uint32_t* tab = lit_here ? blended_rows() : lit_sprite_rows(e.rec->step, L.sleeping);
        constexpr int kRowReg[8] = {1, 3, 4, 5, 6, 7, 8, 9};
        uint64_t left = smask;
#pragma unroll
        for (int s_ = 0; s_ < 8; s_++) {
          if (s_ >= out) break;   // (wave-uniform)
          int from = __builtin_ctzll(left);
          left &= left - 1;
          uint32_t v = w.lane_read(0, from);
          int src = W::mul24(sprite_index((int)(v >> 8)), kTileRows) + (int)(v & 0xFFu);
          w.lane_set(kRowReg[s_], 0, ntex, [&](int t, int) -> uint32_t { return tab[W::mul24(src, ntex) + t]; });
        }
        w.lanes(0, ncell, [&](int k, int lane) {
          if (!((smask >> lane) & 1ull)) return;
          int sidx = __builtin_popcountll(smask & ((1ull << lane) - 1ull));
          sprite_list[sidx] = (uint8_t)k;
          cell_row[k] = (uint8_t)(kSpriteRow0 + sidx);
        });
#pragma unroll
        for (int s_ = 0; s_ < 8; s_++) {
          if (s_ >= out) break;
          w.lanes(0, ntex, [&](int t, int lane) {
            uint32_t px = w.lane_get(kRowReg[s_], lane);
            if (lit_here) {
              int c3[3] = {(int)(px & 0xFF), (int)((px >> 8) & 0xFF), (int)((px >> 16) & 0xFF)};
              px = light(c3, L, 0.0, 0.0);
            }
            cache[W::mul24(kSpriteRow0 + s_, ntex) + t] = px;
          });
        }
      }
    }
    if (prof && w.leader()) prof[12] = w.clock();
    w.sync_lds();
    if (prof && w.leader()) prof[13] = w.clock();
    if (prof && w.leader()) prof[7] = w.clock();
    const int nsp = (int)hdr[1];
    if (nsp > kSpriteRows) return false;
    // every quad that touches a sprite cell, over what early_frame drew there: thread -> (sprite cell, pixel row of the cell, quad of the row)
    const int qpc = (rt.unit_x + 6) / 4 + 1;        // quads a cell's pixel row can touch
    const int per_cell = rt.unit_y * qpc;
    SmallDiv<W> by_cell(per_cell, NT), by_qpc(qpc, NT), by_gh2(c.local_gh, ncell);
    w.each_thread([&](int tid) {
      int s_ = by_cell.div(tid);
      if (s_ < nsp) {
        int rest = tid - by_cell.mul(s_);
        int r = by_qpc.div(rest), qi = rest - by_qpc.mul(r);
        int k = sprite_list[s_];
        int gx = by_gh2.div(k), gy = k - by_gh2.mul(gx);
        int x0 = W::mul24(gx, rt.unit_x);
        int g = (x0 >> 2) + qi;
        int y = W::mul24(gy, rt.unit_y) + r;
        if (4 * g <= x0 + rt.unit_x - 1 && g < gpr) {
          int rm = rowmap[y];
          uint32_t q[4];
#pragma unroll
          for (int kk = 0; kk < 4; kk++) {
            int x = 4 * g + kk;
            bool inside = x < lw;
            int cmk = colmap[inside ? x : lw - 1];
            int row = cell_row[W::mul24(cmk & 0xFF, c.local_gh) + (rm & 0xFF)];
            uint32_t v = cache[W::mul24(row, ntex) + W::mul24(cmk >> 8, rt.unit_y) + (rm >> 8)];
            q[kk] = inside ? (v & 0xFFFFFFu) : 0u;
          }
          Px4 v = {q[0] | (q[1] << 24), (q[1] >> 8) | (q[2] << 16), (q[2] >> 16) | (q[3] << 8)};
          *(Px4*)(rt.out + W::mul24(y, row_bytes) + 12 * g) = v;
        }
      }
      if (items_ok && W::uni((int)(tid >= kItemFirst))) {
        const ItemQuad& iq = item_quad[W::thread_slot(tid)];
        int t0 = tid - kItemFirst;
        int iy0 = by_gpr.div(t0), g = t0 - by_gpr.mul(iy0);
#pragma unroll
        for (int si = 0; si < KI; si++) {
          int iy = iy0 + si * (kItemOwners / gpr);
          if (W::mul24(iy, gpr) + g >= item_quads) continue;
          uint32_t ipx[4];
#pragma unroll
          for (int kk = 0; kk < 4; kk++) ipx[kk] = iq.show[si][kk] ? (iq.px[si][kk] & 0xFFFFFFu) : 0u;
          Px4 v = {ipx[0] | (ipx[1] << 24), (ipx[1] >> 8) | (ipx[2] << 16), (ipx[2] >> 16) | (ipx[3] << 8)};
          *(Px4*)(rt.out + W::mul24(lh + iy, row_bytes) + 12 * g) = v;
        }
      }
      for (int gi = tid; gi < tail_quads; gi += NT) {
        Px4 v = {0u, 0u, 0u};
        *(Px4*)(rt.out + W::mul24(lh + ih, row_bytes) + 12 * gi) = v;
      }
    });
    if (prof && w.leader()) prof[8] = w.clock();
    return true;
  }
  __device__ __forceinline__ bool finish_day_frame(double daylight) {
    Lit L;
    L.D = daylight;
    L.iD = 1 - L.D;
    L.hD = L.iD * 0.5;
    L.night = false;
    L.sleeping = e.rec->sleeping != 0;
    L.amount = 2 * (0.5 - L.D);
    return finish_frame(L);
  }

  // the inventory slot table and list (engine.py:227-248) that direct mode's ItemView pixels read; run by one wave
  __device__ __forceinline__ void build_item_slots() {
    W& w = e.w;
    w.lanes(0, e.R.n_items, [&](int k, int) {
      int amount = e.rec->inv[k];
      int d = (amount >= 1 && amount <= 9) ? amount : 10;  // engine.py:245 ('unknown' otherwise)
      int32_t* t = item_tab + k * 8;
      t[0] = s_tex_icon[k] | (s_tex_alpha[TEX_COUNT + k] ? ALPHA_BIT : 0);
      t[1] = s_tex_digit[d] | (s_tex_alpha[TEX_COUNT + MAX_ITEMS + d] ? ALPHA_BIT : 0);
      t[2] = s_item_pos[k * 4 + 0];
      t[3] = s_item_pos[k * 4 + 1];
      t[4] = s_item_pos[k * 4 + 2];
      t[5] = s_item_pos[k * 4 + 3];
      t[6] = amount;
    });
    uint64_t m = w.ballot(0, e.R.n_items, [&](int k) { return e.rec->inv[k] >= 1; });
    w.lanes(0, e.R.n_items, [&](int k, int lane) {
      if ((m >> lane) & 1ull) slot_list[__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint8_t)k;
    });
    if (w.lane() == 0) hdr[2] = (uint32_t)__builtin_popcountll(m);
  }

  // Per-frame tables: which texture each of the 9x7 grid cells shows (engine.py:168-180), the
  // pixel -> (cell, texel) maps (so the pixel loops contain no division), the inventory slots
  // (engine.py:227-248), the work lists of sprite cells / non-empty slots and the texel cache.
  // slots: also the inventory slot table and list (direct mode's ItemView pixels read them; quad mode does not)
  __device__ __forceinline__ void build_tables(const Lit& L, bool slots = true) {
    const Config& c = e.cfg;
    W& w = e.w;
    Obj p = frame_cells ? Obj{} : e.obj_rd_lane(1);
    int offx = c.local_gw / 2, offy = c.local_gh / 2;
    int ncell = c.local_gw * c.local_gh;
    const int ntex = rt.unit_x * rt.unit_y;
    // The material rows stage_rows left in the row table: the rows this frame shows if they are the raw ones at night, or by
    // day the ones lit for this very step and for the player as he sleeps or wakes now.  Lit rows of another step (the env
    // adopted a new world in this step: it is at step 0) or of the other state (the player fell asleep or woke in this step):
    // the raw rows come back first.
    const bool rows_lit = cache && !L.night && rows_step == e.rec->step && rows_sleeping == L.sleeping;
    if (cache && !rows_lit && rows_step >= 0) {
      const vec16* src = (const vec16*)(e.tb.render_static + render_static_bytes(c) - texel_cache_bytes(c));
      vec16* dst = (vec16*)cache;
      const int words = kSpriteRow0 * ntex;   // (not a byte beyond the material rows: the first wave is about to write the sprite rows behind them)
      w.block_for(words / 4, [&](int i) { dst[i] = src[i]; });   // (read behind the tables' barrier below)
      w.block_for(words & 3, [&](int i) { cache[(words & ~3) + i] = ((const uint32_t*)src)[(words & ~3) + i]; });
      rows_step = -1;
    }
    if (w.wave_is(0)) {
      // One wave: the cell table 64 cells at a time and, in the same breath, the work list of sprite cells by
      // ballot + prefix count (order-preserving, no atomics).  A sprite cell inside the row table's capacity
      // points at its own row.
      SmallDiv<W> by_gh(c.local_gh, ncell);
      int out = 0;
      if (w.leader()) hdr[0] = 0;
      w.wsync();
      for (int base = 0; base < ncell; base += 64) {
        w.lanes(base, ncell, [&](int k, int) {
          int gx = by_gh.div(k), gy = k - by_gh.mul(gx);
          int wx = (int)p.x + gx - offx, wy = (int)p.y + gy - offy;
          int32_t t = -1, s = -1;
          int m = -1, sp = -1;   // material id / sprite texture id of the cell, -1: outside the map / no object
          if (frame_cells) {
            int cm = frame_cells[k], cs = frame_cells[kFrameSprites + k];
            m = cm == 0xFF ? -1 : cm;
            sp = cs == 0xFF ? -1 : cs;
          } else {
            if constexpr (!Env<W, SlotT>::kLane) {   // (the LaneSlots layout never draws: its frames come from frame records)
              if (e.inside(wx, wy)) {
                m = e.mat_at(wx, wy);
                int slot = e.slot_lane(wx, wy);
                if (slot) sp = sprite_of(e.obj_rd_lane(slot));
              }
            }
          }
          if (m >= 0) {
            t = s_tex_tile[TEX_MATERIAL0 + m] | (m << 24);   // atlas offsets are < 2^24
            present[m] = 1;
            if (sp >= 0) {
              s = s_tex_tile[sp] | (s_tex_alpha[sp] ? ALPHA_BIT : 0) | (sp << SPRITE_SHIFT);
            }
          }
          cell_tile[k] = t;
          cell_sprite[k] = s;
          cell_row[k] = (uint8_t)(t >= 0 ? (t >> 24) : kGrayRow);
        });
        w.wsync();
        uint64_t m = w.ballot(base, ncell, [&](int k) { return cell_sprite[k] >= 0; });
        w.lanes(base, ncell, [&](int k, int lane) {
          if (!((m >> lane) & 1ull)) return;
          int sidx = out + __builtin_popcountll(m & ((1ull << lane) - 1ull));
          sprite_list[sidx] = (uint8_t)k;
          if (cache && sidx < kSpriteRows) {
            cell_row[k] = (uint8_t)(kSpriteRow0 + sidx);
            sprite_src[sidx] = (uint8_t)(W::mul24(sprite_index((cell_sprite[k] >> SPRITE_SHIFT) & 63), kTileRows) + (cell_tile[k] >> 24));
          }
        });
        out += __builtin_popcountll(m);
      }
      if (w.leader()) hdr[1] = (uint32_t)out;
      if (cache) {
        // ... and the sprite rows, by the same wave in the same breath (the other waves are waiting for the cell table
        // anyway): tile and sprite blended once per texel (engine.py:176-180) -- which depends on (sprite, material) only --
        // and by day lit (engine.py:189-202: on the step and on whether the player sleeps besides); both were done when the
        // tables were uploaded (render_blended_bytes, render_lit_sprite_bytes).  One lane per texel, four rows' loads in
        // flight together; at night the rows arrive raw (every pixel has its own noise); steps beyond the lit table light
        // what they load.
        int nrow = out < kSpriteRows ? out : kSpriteRows;
        int step = e.rec->step;
        const bool lit_here = !L.night && step >= render_lit_sprite_steps(c);
        const uint32_t* tab = (L.night || lit_here) ? blended_rows() : lit_sprite_rows(step, L.sleeping);
        w.wsync();   // sprite_src
        constexpr int kGroup = 4;
        for (int t0 = 0; t0 < ntex; t0 += 64) {
          for (int s0 = 0; s0 < nrow; s0 += kGroup) {
            w.lanes(t0, ntex, [&](int t, int) {
              uint32_t v[kGroup];
#pragma unroll
              for (int s_ = 0; s_ < kGroup; s_++) v[s_] = tab[W::mul24(sprite_src[s0 + s_ < nrow ? s0 + s_ : 0], ntex) + t];   // (clamped, unconditional)
#pragma unroll
              for (int s_ = 0; s_ < kGroup; s_++) {
                if (s0 + s_ >= nrow) continue;
                uint32_t px = v[s_];
                if (lit_here) {
                  int c3[3] = {(int)(px & 0xFF), (int)((px >> 8) & 0xFF), (int)((px >> 16) & 0xFF)};
                  px = light(c3, L, 0.0, 0.0);
                }
                cache[W::mul24(kSpriteRow0 + s0 + s_, ntex) + t] = px;
              }
            });
          }
        }
      }
    }
    if (slots && w.wave_is(1)) build_item_slots();   // meanwhile, another wave: the inventory slots
    if (prof && w.leader()) prof[12] = w.clock();
    w.sync_lds();
    if (prof && w.leader()) prof[13] = w.clock();
    if (cache && !L.night && !rows_lit) {
      // Day, and the material rows in the table are raw (a step beyond the lit table, a kernel that stages the whole static
      // block -- Env.reset, Env.render, the split step's frame kernel -- or a prediction that failed, see above): the rows in
      // view are lit in place, daylight being one value per frame.
      SmallDiv<W> by_ntex(ntex, kSpriteRow0 * ntex);
      int step = e.rec->step;
      const uint32_t* lit = step < render_lit_steps(c) ? lit_rows(step, L.sleeping) : nullptr;
      w.block_for(kSpriteRow0 * ntex, [&](int i) {
        int row = by_ntex.div(i);
        if (row < kGrayRow && !present[row]) return;
        if (lit) {   // material rows of this step were lit at table upload
          cache[i] = lit[i];
          return;
        }
        uint32_t tile = cache[i];
        int v[3] = {(int)(tile & 0xFF), (int)((tile >> 8) & 0xFF), (int)((tile >> 16) & 0xFF)};
        cache[i] = light(v, L, 0.0, 0.0);
      });
      w.sync_lds();
    }
  }

  // engine.py:276-284 _draw_alpha on one pixel: texel = packed RGBA (little endian), c = canvas bytes
  __device__ __forceinline__ void blend(uint32_t texel, bool has_alpha, int c[3]) const {
    int t0 = texel & 0xFF, t1 = (texel >> 8) & 0xFF, t2 = (texel >> 16) & 0xFF;
    if (!has_alpha) {
      c[0] = t0;
      c[1] = t1;
      c[2] = t2;
      return;
    }
    float a = div255[texel >> 24];
    float ia = 1.0f - a;
    float b0 = a * div255[t0] + ia * div255[c[0]];
    float b1 = a * div255[t1] + ia * div255[c[1]];
    float b2 = a * div255[t2] + ia * div255[c[2]];
    c[0] = (int)(255.0f * b0);
    c[1] = (int)(255.0f * b1);
    c[2] = (int)(255.0f * b2);
  }

  __device__ __forceinline__ static int luma(int r, int g, int b) {  // Pillow RGB -> L
    return (W::mul24(19595, r) + W::mul24(38470, g) + W::mul24(7471, b) + 0x8000) >> 16;   // channels are 0..255
  }

  // tile + sprite of LocalView pixel (vx, vy)  (engine.py:168-180); raw = the texel cache holds raw texels
  __device__ __forceinline__ void local_colour(int vx, int vy, int v[3], bool raw) const {
    int cm = colmap[vx], rm = rowmap[vy];
    int k = W::mul24(cm & 0xFF, e.cfg.local_gh) + (rm & 0xFF);
    int tex = W::mul24(cm >> 8, rt.unit_y) + (rm >> 8);
    int32_t t = cell_tile[k], s = cell_sprite[k];
    uint32_t tile = 0x7F7F7F7Fu;
    if (t >= 0) {
      if (raw && cache)
        tile = cache[W::mul24(t >> 24, W::mul24(rt.unit_x, rt.unit_y)) + tex];
      else
        tile = *(const uint32_t*)(rt.atlas + (t & OFF_MASK) + tex * 4);
    }
    v[0] = tile & 0xFF;
    v[1] = (tile >> 8) & 0xFF;
    v[2] = (tile >> 16) & 0xFF;
    if (s >= 0) blend(*(const uint32_t*)(rt.atlas + (s & OFF_MASK) + tex * 4), (s & ALPHA_BIT) != 0, v);
  }

  // _light and _sleep on one pixel (engine.py:189-202); returns packed 0x00BBGGRR
  __device__ __forceinline__ static uint32_t light(const int v[3], const Lit& L, double m, double noise) {
    int n0 = v[0], n1 = v[1], n2 = v[2];
    if (L.night) {
      double im = 1 - m;
      double mn = m * noise;
      n0 = (int)(im * (double)v[0] + mn);
      n1 = (int)(im * (double)v[1] + mn);
      n2 = (int)(im * (double)v[2] + mn);
    }
    int lum = luma(n0, n1, n2);
    // Pillow ImagingBlend (C float): (u8)(L + 0.4f * (c - L)); then tint (0, 16, 64) by 0.5.  For channels and luma in
    // 0..255 that float expression equals floor((3 L + 2 c) / 5) -- 0.4 (c - L) is either an integer (c - L a multiple of
    // 5, where 0.4f * 5 k rounds to 2 k exactly) or at least 0.2 away from one, and the sum is never negative -- checked
    // over all 65,536 (L, c) pairs (tests/test_render_identities.py).  floor(t / 5) = (t * 13108) >> 16 for t <= 1275, evaluated
    // as the high half of (t << 8) * (13108 << 8): two integer instructions per channel instead of five float ones.
    uint32_t l3 = (uint32_t)W::mul24(lum, 3 << 8);
    int e0 = (int)W::mulhi24(((uint32_t)n0 << 9) + l3, 13108u << 8);
    int e1 = (int)W::mulhi24(((uint32_t)n1 << 9) + l3, 13108u << 8);
    int e2 = (int)W::mulhi24(((uint32_t)n2 << 9) + l3, 13108u << 8);
    // _tint: 0.5 * e + 0.5 * tint with tint = (0, 16, 64); then daylight * canvas + (1 - daylight) * night.  Written as
    // hD * (e + tint), hD = (1 - daylight) * 0.5: e + tint is a small integer, both halvings are exact scalings, so
    // iD * (0.5 * e + 0.5 * tint) and (iD * 0.5) * (e + tint) round the same real number once (checked over every step's
    // daylight value x every e x every tint: identical doubles) -- three f64 operations per channel less.
    double o0 = L.D * (double)v[0] + L.hD * (double)e0;
    double o1 = L.D * (double)v[1] + L.hD * (double)(e1 + 16);
    double o2 = L.D * (double)v[2] + L.hD * (double)(e2 + 64);
    if (L.sleeping) {  // engine.py:198-202
      double g = (double)luma((int)o0, (int)o1, (int)o2);
      o0 = 0.5 * g;   // + 0.5 * 0.0, see above
      o1 = 0.5 * g;
      o2 = 0.5 * g + 0.5 * 16.0;
    }
    return (uint32_t)(int)o0 | ((uint32_t)(int)o1 << 8) | ((uint32_t)(int)o2 << 16);
  }

  // one ItemView pixel of slot k at (ix, iy) relative to the item view origin  (engine.py:227-248)
  __device__ __forceinline__ uint32_t slot_pixel(int k, int vx, int iy) const {
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

  __device__ __forceinline__ uint32_t item_pixel(int vx, int vy, int iy) const {
    int cm = colmap[vx], rm = rowmap[vy];
    int k = (rm & 0xFF) * e.cfg.item_gw + (cm & 0xFF);
    if (k >= e.R.n_items) return 0;
    return slot_pixel(k, vx, iy);
  }

  // canvas pixel (X, Y), direct mode: packed RGB, or 0xFFFFFFFF for night LocalView pixels (the noise
  // pass stores those itself)
  __device__ __forceinline__ uint32_t canvas_pixel(int X, int Y, int lw, int lh, int ih, const Lit& L) const {
    int vx = X - rt.border_x, vy = Y - rt.border_y;
    if (vx < 0 || vy < 0 || vx >= lw || vy >= lh + ih) return 0;   // untouched canvas (env.py:123)
    if (vy >= lh) return item_pixel(vx, vy, vy - lh);
    if (L.night) return 0xFFFFFFFFu;
    int v[3];
    local_colour(vx, vy, v, false);
    return light(v, L, 0.0, 0.0);
  }

  // 3 bytes of pixel (X, Y) of the output image ([Y][X][3], the canvas transposed, env.py:130)
  __device__ __forceinline__ static void put_rgb(uint8_t* image, int sw, int X, int Y, uint32_t rgb) {
    uint8_t* p = image + (uint32_t)W::mul24(W::mul24(Y, sw) + X, 3);   // frames are < 2^24 bytes
    p[0] = (uint8_t)rgb;
    p[1] = (uint8_t)(rgb >> 8);
    p[2] = (uint8_t)(rgb >> 16);
  }

  // Night noise (engine.py:208-209): 2 words of the env's MT19937 stream per LocalView pixel, row-major
  // over [x][y].  mode 0: only advance the stream; 1: pixel j goes to pix[j] (LDS, one word each: consecutive lanes,
  // consecutive words); 2: straight to the output image.  While the consumer
  // waves shade the pixels of one 624-word epoch out of the current state, wave 0 regenerates the
  // next state into the other buffer, and every consumer lane already has the records of its
  // next epoch's pixels in flight (the only global loads of the pass).
  __device__ __forceinline__ void noise_pass(const Lit& L, int mode, int lw, int lh) {
    W& w = e.w;
    const Config& c = e.cfg;
    int sw = rt.size_w;
    int total = lw * lh;
    int words = 2 * total;
    int pos = e.mt_pos;
    uint32_t* cur = e.mt;
    uint32_t* nxt = mtb;
    bool overlap = mode != 0 && nxt != nullptr;
    if (pos >= MT_N) {
      w.sync_lds();
      if (w.wave0()) w.mt_twist(cur);
      w.sync_lds();
      pos = 0;
    }
    // this lane's share of an epoch's pixels: q = q0, q0 + qs, ... (at most W::kEpochSlots of them)
    int q0, qs;
    bool shader = w.consumer_slot(overlap, q0, qs);
    constexpr int K = W::kEpochSlots;
    bool tabled = mode == 1 && cache != nullptr && (int)hdr[1] <= kSpriteRows;   // row table + pixel records
    const NightPx* npx = (const NightPx*)(e.tb.render_static + render_static_bytes(c) + render_item_cells_bytes(c) +
                                          render_lit_bytes(c));
    double vcur[K], vnext[K];
    uint32_t dcur[K], dnext[K];
    auto epoch_first = [&](int s_lo_) { return s_lo_ >> 1; };                       // odd s_lo: first word is the carry
    auto epoch_count = [&](int s_lo_, int s_hi_) { return (s_hi_ >= 2) ? (((s_hi_ - 2) >> 1) - (s_lo_ >> 1) + 1) : 0; };
    // unconditional loads (an idle lane fetches record 0 and drops what it shades): a load under a lane predicate makes
    // the compiler wait for it at the end of the predicated region, i.e. before the epoch it is prefetched for
    auto fetch = [&](double* v, uint32_t* d, int first, int count) {
#pragma unroll
      for (int r = 0; r < K; r++) {
        int q = q0 + r * qs;
        NightPx p = npx[(shader && q < count) ? first + q : 0];
        v[r] = p.vignette;
        d[r] = p.desc;
      }
    };
    int s_lo = 0;
    int s_hi = s_lo + (MT_N - pos);
    if (s_hi > words) s_hi = words;
    if (mode != 0) fetch(vcur, dcur, epoch_first(s_lo), epoch_count(s_lo, s_hi));
    uint32_t carry = 0;
    SmallDiv<W> by_lh(lh, total);
    int ntex = rt.unit_x * rt.unit_y;
    while (s_lo < words) {
      bool more = s_hi < words;
      int n_lo = s_hi, n_hi = s_hi + MT_N;   // next epoch starts on a fresh state
      if (n_hi > words) n_hi = words;
      if (more && mode != 0) fetch(vnext, dnext, epoch_first(n_lo), epoch_count(n_lo, n_hi));
      if (more && overlap && w.producer()) w.mt_twist_from(cur, nxt);
      if (tabled && shader) {
        // Fast path, written stage by stage over the lane's pixels so that their dependency chains
        // (stream words -> noise, record -> cell row -> texel) are in flight together: indices are
        // clamped instead of predicated and only the final store is guarded.
        int j_first = epoch_first(s_lo);
        int count = epoch_count(s_lo, s_hi);
        int j_safe = j_first < total ? j_first : total - 1;
        bool ok[K];
        int jj[K], row[K];
        uint32_t wa[K], wb[K], raw[K];
#pragma unroll
        for (int r = 0; r < K; r++) {
          int q = q0 + r * qs;
          ok[r] = q < count;
          jj[r] = ok[r] ? j_first + q : j_safe;
          int ia = 2 * jj[r] - s_lo;          // -1 only for the epoch's first pixel: its first word is the carry
          uint32_t a = cur[pos + (ia >= 0 ? ia : 0)];
          wa[r] = ia >= 0 ? a : carry;
          wb[r] = cur[pos + ia + 1];
          row[r] = cell_row[dcur[r] & 0xFF];
        }
#pragma unroll
        for (int r = 0; r < K; r++) raw[r] = cache[W::mul24(row[r], ntex) + (dcur[r] >> 8)];
#pragma unroll
        for (int r = 0; r < K; r++) {
          double noise = mt_uniform_32_127(mt_temper(wa[r]), mt_temper(wb[r]));
          int v[3] = {(int)(raw[r] & 0xFF), (int)((raw[r] >> 8) & 0xFF), (int)((raw[r] >> 16) & 0xFF)};
          double m = L.amount * vcur[r];
          uint32_t rgb = light(v, L, m, noise);
          if (ok[r]) pix[jj[r]] = rgb;
        }
      } else if (mode != 0 && shader) {   // generic path: no row table (other render sizes) or sprite cells beyond its rows
        int j_first = epoch_first(s_lo);
        int count = epoch_count(s_lo, s_hi);
#pragma unroll
        for (int r = 0; r < K; r++) {
          int q = q0 + r * qs;
          if (q >= count) continue;
          int j = j_first + q;
          int ia = 2 * j - s_lo;
          uint32_t a = (ia >= 0) ? cur[pos + ia] : carry;
          uint32_t b = cur[pos + ia + 1];
          double noise = mt_uniform_32_127(mt_temper(a), mt_temper(b));
          int x = by_lh.div(j);
          int y = j - by_lh.mul(x);
          int v[3];
          local_colour(x, y, v, true);
          double m = L.amount * vcur[r];
          uint32_t rgb = light(v, L, m, noise);
          if (mode == 1) pix[j] = rgb;
          else put_rgb(rt.out, sw, x + rt.border_x, y + rt.border_y, rgb);
        }
      }
      pos += s_hi - s_lo;
      s_lo = s_hi;
      s_hi = n_hi;
      if (more) {   // the epoch ran to the end of the state
        carry = cur[MT_N - 1];
        w.sync_lds();   // (the state buffers are LDS; this epoch's pixel stores and the next one's record loads stay in flight)
        if (overlap) {
          uint32_t* t = cur;
          cur = nxt;
          nxt = t;
        } else {
          if (w.wave0()) w.mt_twist(cur);
          w.sync_lds();
        }
        pos = 0;
#pragma unroll
        for (int r = 0; r < K; r++) {
          vcur[r] = vnext[r];
          dcur[r] = dnext[r];
        }
      }
    }
    if (pix_global && mode == 1) W::drain_stores();   // the quads of other waves read these pixels back from L2
    w.sync();
    if (cur != e.mt) {
      w.block_for(MT_N, [&](int i) { e.mt[i] = cur[i]; });
      w.sync_lds();
    }
    e.mt_pos = pos;
    e.rng_invalidate();
  }

  // Night noise from states generated AHEAD (env_kernels.hpp noise_chain; noise_raw = the env's kNoiseStates consecutive
  // states in global memory, noise_base = the stream position the rules stopped at, in words from the first state's first
  // word): pixel j takes words noise_base + 2 j and + 2 j + 1 wherever they lie -- no state is regenerated here, so there
  // is no order among the pixels, no epoch and no barrier: every thread lights pixels tid, tid + NT, ... (consecutive lanes:
  // consecutive 8-byte word pairs and 16-byte pixel records, the loads of the pixel after next already in flight) and
  // leaves them in `pix` in stream order, where the quads pick them up exactly as after noise_pass.
  __device__ __forceinline__ void noise_pass_ahead(const Lit& L, int lw, int lh) {
    W& w = e.w;
    const Config& c = e.cfg;
    constexpr int NT = W::kThreads;
    const int total = lw * lh, ntex = rt.unit_x * rt.unit_y;
    const NightPx* npx = (const NightPx*)(e.tb.render_static + render_static_bytes(c) + render_item_cells_bytes(c) + render_lit_bytes(c));
    const uint32_t* words = noise_raw + noise_base;
    w.each_thread([&](int tid) {
      constexpr int D = 2;   // pixels whose loads are in flight ahead of the one being lit
      uint32_t wa[D + 1], wb[D + 1], ds[D + 1];
      double vg[D + 1];
      auto fetch = [&](int j, int slot) {   // (clamped, unconditional)
        int jj = j < total ? j : total - 1;
        wa[slot] = words[2 * jj];
        wb[slot] = words[2 * jj + 1];
        NightPx p = npx[jj];
        vg[slot] = p.vignette;
        ds[slot] = p.desc;
      };
#pragma unroll
      for (int d = 0; d < D; d++) fetch(tid + d * NT, d);
#pragma clang loop unroll(disable)
      for (int j = tid; j < total; j += (D + 1) * NT) {
#pragma unroll
        for (int u = 0; u <= D; u++) {   // (unrolled by D + 1 so that every register slot is named by a literal)
          int jn = j + u * NT;
          fetch(jn + D * NT, (u + D) % (D + 1));
          int row = cell_row[ds[u] & 0xFF];
          uint32_t raw = cache[W::mul24(row, ntex) + (ds[u] >> 8)];
          double noise = mt_uniform_32_127(mt_temper(wa[u]), mt_temper(wb[u]));
          int v[3] = {(int)(raw & 0xFF), (int)((raw >> 8) & 0xFF), (int)((raw >> 16) & 0xFF)};
          uint32_t rgb = light(v, L, L.amount * vg[u], noise);
          if (jn < total) pix[jn] = rgb;
        }
      }
    });
    if (pix_global) W::drain_stores();   // the quads of other waves read these pixels back from L2
    w.sync();
  }

  // Full frame.  pixels == false: only the RNG side effect of a night frame happens.
  //
  // Quad mode (row table in LDS, no border, rows a multiple of four pixels -- the default geometry): every quad of four
  // output pixels is owned by one thread, 12 bytes per lane and 768 contiguous bytes per wave, and nothing but a night
  // frame's pixels is staged.  Day: LocalView quads come straight from the row table.  Night: every pixel needs its own
  // two words of the noise stream, which runs down the columns; noise_pass leaves the pixels in LDS in that order (for
  // small worlds in the LDS that held the env's map copies, dead once the per-frame tables exist) and the quads pick
  // them up transposed -- lane (g, y) reads words (4 g + k) * lh + y: 64 distinct banks.  Inventory row quads read the
  // finished cells of their slots (global, L2-resident: loads issued first, stored last); what is left of the canvas is
  // zeros (env.py:123).  No global store is issued before the last global load (on gfx9 a load behind a store waits
  // for it).
  // Direct mode (any other geometry): one lane per pixel straight to the output.
  // (hint_step, hint_D): a daylight value the caller fetched early, used if it is for the current step.
  __device__ __forceinline__ void render(bool pixels, int hint_step = -1, double hint_D = 0.0) {
    const Config& c = e.cfg;
    W& w = e.w;
    int lw = c.local_gw * rt.unit_x, lh = c.local_gh * rt.unit_y;
    int ih = c.item_gh * rt.unit_y;
    Lit L;
    L.D = (e.rec->step == hint_step) ? hint_D : e.tb.daylight[e.rec->step];
    L.iD = 1 - L.D;
    L.hD = L.iD * 0.5;
    L.night = L.D < 0.5;
    L.sleeping = e.rec->sleeping != 0;
    L.amount = 2 * (0.5 - L.D);
    int sw = rt.size_w, sh = rt.size_h;
    if (L.night) e.mark_mt_rewritten();   // a night frame's noise regenerates the stream's state ten times over, drawn or not
    if (!pixels) {
      if (L.night) noise_pass(L, 0, lw, lh);
      return;
    }
    constexpr int NT = W::kThreads;
    constexpr int KR = NT == 512 ? 2 : NT >= 256 ? 4 : NT >= 192 ? 5 : 13;   // LocalView rows of a thread whose look-up chains run side by side (16 quads per row: 49 rows <= KR * NT / 16)
    int gpr = sw >> 2;
    int item_quads = gpr * ih, tail_quads = gpr * (sh - lh - ih);
    int ntex = rt.unit_x * rt.unit_y;
    bool quads = cache != nullptr && pix != nullptr && rt.border_x == 0 && rt.border_y == 0 && (sw & 3) == 0 && lw <= sw &&
                 NT % gpr == 0 && lh <= KR * (NT / gpr) && tail_quads >= 0 && c.item_gw == c.local_gw;
    SmallDiv<W> by_gpr(gpr, NT);
    // Inventory quad of the thread (quad mode): its four finished texels are global loads -- issued now, before the
    // per-frame tables are built, and placed when the frame goes out.  Unconditional loads from clamped addresses (cell 0
    // when there is nothing to show), masked afterwards: a load under a lane predicate is waited for at the end of its
    // predicated region, one load latency after the other.
    // The inventory quads belong to the threads behind the first wave (which is busy with the cell table until the
    // tables' barrier): quad q = tid - 64 + s * (threads - 64), KI of them per thread at most.
    constexpr int KI = 2;
    constexpr int kItemOwners = NT > 64 ? NT - 64 : NT;
    constexpr int kItemFirst = NT > 64 ? 64 : 0;
    struct ItemQuad {
      uint32_t px[KI][4];
      bool show[KI][4];
    };
    ItemQuad item_quad[W::kThreadSlots];
    quads = quads && item_quads <= KI * kItemOwners && kItemOwners % gpr == 0 && kItemFirst % gpr == 0;
    if (quads) {
      const uint32_t* item_cells = (const uint32_t*)(e.tb.render_static + render_static_bytes(c));
      w.each_thread([&](int tid) {
        if (!W::uni((int)(tid >= kItemFirst))) return;   // a whole-wave (scalar) branch: the first wave skips the code, not just its lanes
        ItemQuad& iq = item_quad[W::thread_slot(tid)];
        int t0 = tid - kItemFirst;
        int y0 = by_gpr.div(t0), g = t0 - by_gpr.mul(y0);   // the column group is tid % gpr for every quad of the thread
#pragma unroll
        for (int s_ = 0; s_ < KI; s_++) {
          int iy = y0 + s_ * (kItemOwners / gpr);
          bool mine = W::mul24(iy, gpr) + g < item_quads;
          int rm = rowmap[lh + (mine ? iy : 0)];
          int cy = rm & 0xFF, ty = rm >> 8;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            int x = 4 * g + k;
            int cm = colmap[x < lw ? x : lw - 1];
            int slot = W::mul24(cy, c.item_gw) + (cm & 0xFF);
            bool has = mine && x < lw && slot < e.R.n_items;
            int amount = e.rec->inv[has ? slot : 0];
            iq.show[s_][k] = has && amount >= 1;
            int d = amount <= 9 ? amount : 10;   // engine.py:245: 'unknown' beyond 9
            int at = W::mul24(W::mul24(slot, kItemDigits) + d, ntex) + W::mul24(cm >> 8, rt.unit_y) + ty;
            iq.px[s_][k] = item_cells[iq.show[s_][k] ? at : 0];
          }
        }
      });
    }
    build_tables(L, !quads);
    if (prof && w.leader()) prof[7] = w.clock();
    if (quads && (L.night || (int)hdr[1] <= kSpriteRows)) {   // (a day view with more sprite cells than the table has rows: direct mode)
      // the noise's states are waiting in global memory (noise_chain) and every pixel's row is in the table: one pass over the
      // pixels with no epochs in it (noise_pass_ahead)
      constexpr int kAheadWords = kNoiseStates * MT_N;
      const bool ahead = L.night && noise_raw != nullptr && (int)hdr[1] <= kSpriteRows && noise_base + 2 * lw * lh <= kAheadWords;
      if (ahead) noise_pass_ahead(L, lw, lh);             // ends on a barrier
      else if (L.night) noise_pass(L, 1, lw, lh);         // ends on a barrier
      int row_bytes = 3 * sw;
      int rows_per = NT / gpr;
      struct Px4 { uint32_t a, b, c; };
      w.each_thread([&](int tid) {
        // the thread's quads all sit in column group g = tid % gpr, in rows tid / gpr + n * (threads / gpr)
        int y0 = by_gpr.div(tid), g = tid - by_gpr.mul(y0);
        int cm[4];
        bool in[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          int x = 4 * g + k;
          in[k] = x < lw;   // beyond the view: untouched canvas
          cm[k] = colmap[in[k] ? x : lw - 1];
        }
        int yy[KR];
        uint32_t px[KR][4];
#pragma unroll
        for (int r = 0; r < KR; r++) yy[r] = y0 + r * rows_per;
        if (L.night) {
#pragma unroll
          for (int r = 0; r < KR; r++) {
            int y = yy[r] < lh ? yy[r] : lh - 1;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const uint32_t* at = pix + (W::mul24(in[k] ? 4 * g + k : lw - 1, lh) + y);
              px[r][k] = pix_global ? W::load_fresh(at) : *at;
            }
          }
        } else {
          // stage by stage over the thread's rows (clamped instead of predicated, only the store is guarded): row map,
          // cell rows, texels -- up to 16 independent chains in flight
          int rbase[KR], ty[KR], row[KR][4];
#pragma unroll
          for (int r = 0; r < KR; r++) {
            int rm = rowmap[yy[r] < lh ? yy[r] : lh - 1];
            rbase[r] = rm & 0xFF;
            ty[r] = rm >> 8;
          }
#pragma unroll
          for (int r = 0; r < KR; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) row[r][k] = cell_row[W::mul24(cm[k] & 0xFF, c.local_gh) + rbase[r]];
#pragma unroll
          for (int r = 0; r < KR; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) px[r][k] = cache[W::mul24(row[r][k], ntex) + W::mul24(cm[k] >> 8, rt.unit_y) + ty[r]];
        }
#pragma unroll
        for (int r = 0; r < KR; r++) {
#pragma unroll
          for (int k = 0; k < 4; k++) px[r][k] = in[k] ? (px[r][k] & 0xFFFFFFu) : 0u;
          Px4 v = {px[r][0] | (px[r][1] << 24), (px[r][1] >> 8) | (px[r][2] << 16), (px[r][2] >> 16) | (px[r][3] << 8)};
          if (yy[r] < lh) *(Px4*)(rt.out + W::mul24(yy[r], row_bytes) + 12 * g) = v;
        }
        if (W::uni((int)(tid >= kItemFirst))) {
          const ItemQuad& iq = item_quad[W::thread_slot(tid)];
          int t0 = tid - kItemFirst;
          int iy0 = by_gpr.div(t0);
#pragma unroll
          for (int s_ = 0; s_ < KI; s_++) {
            int iy = iy0 + s_ * (kItemOwners / gpr);
            if (W::mul24(iy, gpr) + g >= item_quads) continue;
            uint32_t ipx[4];
#pragma unroll
            for (int k = 0; k < 4; k++) ipx[k] = iq.show[s_][k] ? (iq.px[s_][k] & 0xFFFFFFu) : 0u;
            Px4 v = {ipx[0] | (ipx[1] << 24), (ipx[1] >> 8) | (ipx[2] << 16), (ipx[2] >> 16) | (ipx[3] << 8)};
            *(Px4*)(rt.out + W::mul24(lh + iy, row_bytes) + 12 * g) = v;
          }
        }
        for (int gi = tid; gi < tail_quads; gi += NT) {
          Px4 v = {0u, 0u, 0u};
          *(Px4*)(rt.out + W::mul24(lh + ih, row_bytes) + 12 * gi) = v;
        }
      });
      if (ahead) {   // the stream moved on by the frame's 2 * lw * lh words: the state it stopped in comes back from the scratch
        int end = noise_base + 2 * lw * lh;              // words consumed since the staged state's first one
        int s_fin = (end - 1) / MT_N;                    // the state that holds the last word drawn
        const vec16* src = (const vec16*)(noise_raw + (size_t)s_fin * MT_N);
        vec16* dst = (vec16*)e.mt;
        w.block_for(MT_N / 4, [&](int i) { dst[i] = src[i]; });
        e.mt_pos = end - s_fin * MT_N;                   // 1 .. 624
        e.rng_invalidate();
      }
      if (prof && w.leader()) prof[8] = w.clock();
      return;
    }
    // ---- direct mode
    if (quads) {   // (quad mode skipped the inventory slot table)
      if (w.wave_is(0)) build_item_slots();
      w.sync();
    }
    if (L.night) noise_pass(L, 2, lw, lh);
    if (prof && w.leader()) prof[8] = w.clock();
    w.block_for(sw * sh, [&](int p) {
      int Y = p / sw, X = p - Y * sw;
      uint32_t v = canvas_pixel(X, Y, lw, lh, ih, L);
      if (v != 0xFFFFFFFFu) put_rgb(rt.out, sw, X, Y, v);
    });
  }
};

}  // namespace crafter
