// Bodies of the two per-environment kernels (one workgroup = one environment):
//   step_body  : Env.step  (env.py:83-118)  -> dynamics, balance, reward/done, obs render
//   reset_body : Env.reset (env.py:70-81)   -> reseed, worldgen, first obs
// plus the LDS carve-up and the HBM <-> LDS staging they share.
//
// HBM layout is struct-of-arrays over envs (types.hpp StatePtrs).  A kernel stages one env's
// working set into LDS, runs the rules there, writes map changes through to HBM as they happen
// and stores the compact tables (slot table, MT key, scalar record, chunk order) on exit.
#pragma once
#include "env_core.hpp"
#include "render.hpp"
#include "worldgen.hpp"

namespace crafter {

// The geometry everybody uses -- crafter.Env() defaults: 64x64 world, 9x9 view, 64x64 image (env.py:27-46) -- as
// compile-time constants: with GEO = 1 the step kernel overwrites those Config fields with literals, and since
// every helper is inlined into it the compiler folds them everywhere (LDS offsets become immediates, divisions
// by the unit / grid sizes become multiplies, loop trip counts are known).  Any other configuration runs the
// generic instance (GEO = 0) of the same code.
__host__ __device__ inline bool is_default_geometry(const Config& c) {
  return c.W == 64 && c.H == 64 && c.view_w == 9 && c.view_h == 9 && c.size_w == 64 && c.size_h == 64 && c.unit_x == 7 &&
         c.unit_y == 7 && c.local_gw == 9 && c.local_gh == 7 && c.item_gw == 9 && c.item_gh == 2 && c.border_x == 0 &&
         c.border_y == 0 && c.icon_w == 5 && c.icon_h == 5 && c.digit_w == 4 && c.digit_h == 4 && c.max_objects == 256 &&
         c.nchunk_x == 6 && c.nchunk_y == 6 && c.update_dist == 18;
}
__device__ __forceinline__ Config with_default_geometry(Config c) {
  c.W = 64; c.H = 64; c.view_w = 9; c.view_h = 9; c.size_w = 64; c.size_h = 64; c.unit_x = 7; c.unit_y = 7;
  c.local_gw = 9; c.local_gh = 7; c.item_gw = 9; c.item_gh = 2; c.border_x = 0; c.border_y = 0; c.icon_w = 5; c.icon_h = 5;
  c.digit_w = 4; c.digit_h = 4; c.max_objects = 256; c.nchunk_x = 6; c.nchunk_y = 6; c.update_dist = 18;
  return c;
}

// ... and the same for worlds of ANY size seen through crafter.Env()'s default view (9x9 view, 64x64 image; GEO = 2: BASELINE
// configs[3], 256x256): everything about the frame and the update distance folds, the world's extent, its chunk grid and the
// slot table's length stay run-time values.
__host__ __device__ inline bool is_default_view(const Config& c) {
  return c.view_w == 9 && c.view_h == 9 && c.size_w == 64 && c.size_h == 64 && c.unit_x == 7 && c.unit_y == 7 && c.local_gw == 9 &&
         c.local_gh == 7 && c.item_gw == 9 && c.item_gh == 2 && c.border_x == 0 && c.border_y == 0 && c.icon_w == 5 && c.icon_h == 5 &&
         c.digit_w == 4 && c.digit_h == 4 && c.update_dist == 18;
}
__device__ __forceinline__ Config with_default_view(Config c) {
  c.view_w = 9; c.view_h = 9; c.size_w = 64; c.size_h = 64; c.unit_x = 7; c.unit_y = 7;
  c.local_gw = 9; c.local_gh = 7; c.item_gw = 9; c.item_gh = 2; c.border_x = 0; c.border_y = 0; c.icon_w = 5; c.icon_h = 5;
  c.digit_w = 4; c.digit_h = 4; c.update_dist = 18;
  return c;
}

struct LdsLayout {
  int maps_in_lds;   // 1: mat + objmap are staged in LDS; 0: large world, the maps stay in HBM (L2)
  int frame_over_objs = 0;   // the LDS frame extends over the slot table: objs must be stored before the frame is composed
  int mat, objmap, frame, frame_bytes, objs, mt, rec, rules, chunk_order, chunk_seen, census, wg, scratch, render, total;
  int mtb = -1;      // the second MT19937 state (night frames; the worldgen's stream window): inside the wg region
  int wmat = -1, wobj = -1;   // big_layout: the windows of the two maps around the player (env_core.hpp FarSlot)
  int far = -1;      // big_layout: the scan's counters, index, record cache and near bits (env_core.hpp FarSlot); the slot table itself stays in global memory
  int total_no_render;   // the renderer's region comes last: kernels that never draw (world-pool generation) launch without it
};

// Worlds whose maps (3 bytes per cell) would push one env's LDS past this stay in HBM.
constexpr int kMaxLdsWithMaps = 96 * 1024;
// The step kernels' compact layout keeps of the worldgen scratch only what a step uses: the second MT19937 state, and in front of
// it the few bytes by which a night frame's pixel buffer (63 x 49 words) outgrows the maps and the slot table it recycles.  Round 6:
// 64 bytes instead of the scratch's first KB -- with the /255 table out of LDS (render.hpp) the default instance's workgroup is
// 24,848 B = 20 LDS granules of 1,280 B instead of 21: six of them leave a CU 8 granules, and a classification workgroup of the
// world pool (6 granules) runs BESIDE them instead of displacing one.
constexpr int kCompactWgHead = 64;
constexpr int kRenderStaticBound = 20 * 1024;   // >= render_static_bytes + sprite_rows_bytes of any accepted frame size

// slot_bytes: sizeof of the cell -> slot map's element in THIS kernel's LDS (the map is derived state, every kernel
// rebuilds its own): 2, or 1 for the step kernel's default-geometry instance.
// lean: the rule kernel of the split step (crafter_rules_kernel): no worldgen scratch / second MT state, no renderer region.
// LaneSlots layout (the rule kernel of the default instance, env_core.hpp): a window of the material map, no slot map,
// no staged rules, no worldgen scratch, no renderer region.
__host__ __device__ inline LdsLayout lane_layout(const Config& c) {
  LdsLayout L;
  int nch = c.nchunk_x * c.nchunk_y;
  int o = 0;
  L.maps_in_lds = 1;
  L.mat = o;          o += align16(kWinX * kWinY);
  L.objmap = -1;
  L.frame = 0;
  L.frame_bytes = 0;
  L.objs = o;         o += 16 * c.max_objects;
  L.wg = -1;
  L.mt = o;           o += align16(4 * MT_N);
  L.rec = o;          o += align16((int)sizeof(EnvRec));
  L.rules = -1;
  L.chunk_order = o;  o += align16(2 * nch);
  L.chunk_seen = o;   o += align16(nch);
  L.census = o;       o += align16(20 * nch);
  L.scratch = o;      o += 16;
  L.total_no_render = o;
  L.render = o;
  L.total = o;
  return L;
}
__host__ __device__ inline bool lane_layout_ok(const Config& c) {   // the window must fit the map, rows must be 8-byte aligned
  return c.W >= kWinX && c.H >= kWinY && c.H % 8 == 0 && c.max_objects <= 256;
}

// with_rules false: the instance runs the compiled-in rules and stages none (the resident rollout of the default instance:
// 280 bytes that decide whether a sixth workgroup fits a CU)
__host__ __device__ inline LdsLayout lds_layout(const Config& c, int slot_bytes = 2, bool lean = false, bool with_rules = true) {
  LdsLayout L;
  int cells = c.W * c.H;
  int nch = c.nchunk_x * c.nchunk_y;
  // Residency must not depend on the frame size: a render(size) handle (another unit / atlas over the SAME state
  // buffers) has to come to the same decision as the handle that steps, because the slot map of an LDS-resident
  // world is derived state that is never stored.  So the decision prices the renderer's region at a fixed bound
  // (crafter_create rejects frame sizes whose tables exceed it) instead of at render_lds_bytes(c).
  int rest = 16 * c.max_objects + align16(4 * MT_N) + align16((int)sizeof(EnvRec)) + CRAFTER_RULES_HEAD_BYTES + align16(2 * nch) + align16(nch) +
             align16(20 * nch) + render_frame_bytes(c) + kRenderStaticBound + align16(WG_LDS_BYTES) + 16;
  int maps = align16(cells) + align16(slot_bytes * cells);
  // a night frame's pixels wait in LDS, one word each in the noise stream's order (render.hpp), when they fit
  int want_frame = align16(4 * c.local_gw * c.unit_x * c.local_gh * c.unit_y);
  L.maps_in_lds = (align16(cells) + align16(2 * cells) + rest <= kMaxLdsWithMaps) ? 1 : 0;   // same answer for every slot_bytes
  int o = 0;
  if (L.maps_in_lds) {
    L.mat = o;        o += align16(cells);
    L.objmap = o;     o += align16(slot_bytes * cells);
    // the map copies are dead once the per-frame render tables exist: the pixel buffer reuses their LDS -- and, when the
    // maps alone are too small, the slot table behind them, which is then stored to HBM before the frame is drawn
    // (frame_over_objs), and in the step kernel's compact layout the first bytes of the worldgen scratch behind that,
    // which no step uses
    L.frame = 0;
    L.frame_bytes = (!lean && want_frame <= maps + 16 * c.max_objects + (slot_bytes == 1 ? kCompactWgHead : 0)) ? want_frame : 0;
    L.frame_over_objs = L.frame_bytes > maps;
  } else {
    L.mat = L.objmap = -1;
    L.frame = o;
    L.frame_bytes = want_frame <= 16 * 1024 ? want_frame : 0;
    o += L.frame_bytes;
  }
  L.objs = o;         o += 16 * c.max_objects;
  // compact layout (the step kernels): right behind the slot table, and without noise3's tables -- a step only uses the
  // second MT19937 state (night frames) and the first KB (the tail of a night frame's pixel buffer)
  if (slot_bytes == 1) { L.wg = o; L.mtb = o + kCompactWgHead; o += lean ? 0 : kCompactWgHead + align16(4 * MT_N); }
  L.mt = o;           o += align16(4 * MT_N);
  L.rec = o;          o += align16((int)sizeof(EnvRec));
  L.rules = o;        o += with_rules ? CRAFTER_RULES_HEAD_BYTES : 0;
  L.chunk_order = o;  o += align16(2 * nch);
  L.chunk_seen = o;   o += align16(nch);
  L.census = o;       o += align16(20 * nch);
  if (slot_bytes != 1) { L.wg = o; L.mtb = o + 1024; o += align16(WG_LDS_BYTES); }
  L.scratch = o;      o += 16;
  L.total_no_render = o;
  L.render = o;       o += lean ? 0 : align16(render_lds_bytes(c));
  L.total = o;
  return L;
}

// The step kernel's layout for worlds whose maps stay in HBM (crafter_step_kernel<0, 0, 0>, the rollout instance of the
// same; 256x256: BASELINE configs[3]).  lds_layout prices such an env at 73 KB -- two step workgroups per CU, the bound of
// configs[3] in round 3 -- of which a step needs neither the night frame's pixel buffer (12.4 KB: the env's
// scratch in global memory, as the frame kernel of the split step keeps it: Renderer::pix_global), nor the census (9.7 KB
// for 484 chunks: Env::census_global), nor noise3's tables in the worldgen scratch (3 KB: a step only uses the second MT
// state).  48.3 KB: three per CU -- until round 6, which took the slot table out as well (32 KB for 2048 slots: it stays in
// global memory, every write goes through to it, and what a step reads of it comes from one scan at stage-in -- env_core.hpp
// FarSlot): 13.1 KB + 2 KB of cache and near bits = 15.1 KB.
__host__ __device__ inline LdsLayout big_layout(const Config& c) {
  LdsLayout L;
  int nch = c.nchunk_x * c.nchunk_y;
  int o = 0;
  L.maps_in_lds = 0;
  L.mat = L.objmap = -1;
  L.frame = 0;
  L.frame_bytes = 0;
  L.frame_over_objs = 0;
  L.objs = -1;
  L.far = o;          o += far_lds_bytes(c.max_objects);
  L.wg = L.mtb = o;   o += align16(4 * MT_N);   // (no pixel buffer in LDS, no worldgen: of the scratch only the second MT state)
#if CRAFTER_FAR_WINDOW
  L.wmat = o;         o += align16(kWinX * kWinY);
  L.wobj = o;         o += align16(2 * kWinX * kWinY);
#else
  L.wmat = L.wobj = o;
#endif
  L.mt = o;           o += align16(4 * MT_N);
  L.rec = o;          o += align16((int)sizeof(EnvRec));
  L.rules = o;        o += CRAFTER_RULES_HEAD_BYTES;
  L.chunk_order = o;  o += align16(2 * nch);
  L.chunk_seen = o;   o += align16(nch);
  L.census = -1;
  L.scratch = o;      o += 16;
  L.total_no_render = o;
  L.render = o;       o += align16(render_lds_bytes(c));
  L.total = o;
  return L;
}

// Env.reset / the regeneration kernel / the fused generation appended to crafter_reset_kernel, for worlds whose maps stay in
// HBM: the slot table is written where it lives (the env's -- or the pool entry's -- table in global memory: worldgen only
// ever appends to it) and no night-frame pixel buffer is kept (a reset frame is step 0: day).  73 -> 29 KB per workgroup:
// the regeneration kernel, launched after EVERY step to look at an all but always empty queue, waited 100-340 us per step for
// a CU with 73 KB of LDS free next to the generation kernels at 8192 x 256x256 (r4f / r4g).
__host__ __device__ inline LdsLayout big_reset_layout(const Config& c) {
  LdsLayout L = lds_layout(c);
  if (L.maps_in_lds) return L;
  int nch = c.nchunk_x * c.nchunk_y;
  int o = 0;
  L.frame = 0;
  L.frame_bytes = 0;
  L.frame_over_objs = 0;
  L.objs = -1;
  L.mt = o;           o += align16(4 * MT_N);
  L.rec = o;          o += align16((int)sizeof(EnvRec));
  L.rules = o;        o += CRAFTER_RULES_HEAD_BYTES;
  L.chunk_order = o;  o += align16(2 * nch);
  L.chunk_seen = o;   o += align16(nch);
  L.census = o;       o += align16(20 * nch);
  L.wg = o; L.mtb = o + 1024; o += align16(WG_LDS_BYTES);
  L.scratch = o;      o += 16;
  L.total_no_render = o;
  L.render = o;       o += align16(render_lds_bytes(c));
  L.total = o;
  return L;
}

// LM: 1 / 0 = the caller knows at compile time that the maps are LDS-resident / stay in HBM, -1 = decided at run
// time.  It matters for the step kernel: with a run-time choice the map pointers are address-space-unknown and
// every map access of the rule code becomes a FLAT instruction instead of a DS one.
template <class W, int LM = -1, class S = uint16_t>
__device__ __forceinline__ void bind_lds(Env<W, S>& e, uint8_t* smem, const LdsLayout& L, const StatePtrs& st, int env) {
  const Config& c = e.cfg;
  size_t cells = (size_t)c.W * c.H;
  e.g_mat = st.mat + (size_t)env * cells;
  // The slot map (cell -> slot) of an LDS-resident world is derived state: rebuilt from the slot table
  // at stage-in, never written back (StatePtrs.objmap is only live for worlds whose maps stay in HBM).
  if constexpr (Env<W, S>::kLane) {
    e.g_objmap = nullptr;
    e.mat = smem + L.mat;   // the window (load_env_issue decides where it sits)
    e.objmap = nullptr;
  } else if (LM == 1) {
    e.g_objmap = nullptr;
    e.mat = smem + L.mat;
    e.objmap = (S*)(smem + L.objmap);
  } else if (LM == 0) {
    e.g_objmap = st.objmap + (size_t)env * cells;
    e.mat = e.g_mat;
    e.objmap = (S*)e.g_objmap;   // LM == 0 implies S == uint16_t
  } else {
    e.g_objmap = L.maps_in_lds ? nullptr : st.objmap + (size_t)env * cells;
    e.mat = L.maps_in_lds ? smem + L.mat : e.g_mat;
    e.objmap = L.maps_in_lds ? (S*)(smem + L.objmap) : (S*)e.g_objmap;
  }
  e.objs = L.objs >= 0 ? (Obj*)(smem + L.objs) : st.objs + (size_t)env * c.max_objects;   // (big_reset_layout, big_layout: in place)
  if constexpr (Env<W, S>::kFar) {
    e.nctr = (uint32_t*)(smem + L.far);
    e.ndm = smem + L.far + 16;
    e.ncache = (Obj*)(smem + L.far + 16 + 256);
    e.near_mask = (uint64_t*)(smem + L.far + 16 + 256 + 16 * kFarCache);
    e.wmat = smem + L.wmat;
    e.wobj = (uint16_t*)(smem + L.wobj);
    e.win_x0 = e.win_y0 = -(1 << 20);   // (no window yet: every cell is read from the maps)
  }
  e.mt = (uint32_t*)(smem + L.mt);
  e.rec = (EnvRec*)(smem + L.rec);
  e.chunk_order = (uint16_t*)(smem + L.chunk_order);
  e.chunk_seen = smem + L.chunk_seen;
  if (L.census >= 0) {
    e.census = (int32_t*)(smem + L.census);
  } else {   // big_layout: the census is read and updated in place
    e.census = st.census + (size_t)env * c.nchunk_x * c.nchunk_y * 5;
    e.census_global = true;
  }
}

// HBM -> LDS in the two phases of stage_issue / stage_commit (env_core.hpp): every load of the env's
// state is in flight before the first one is waited for.
// everything: 1 = the whole state (step, render), 0 = only the scalar record (reset overwrites the rest)
// Registers per thread and array: sized for the default configuration's arrays at W's workgroup width, so that nothing
// falls through to stage_rest's byte loop (256 threads: one each; the split step's 64-thread rule kernel: up to four).
// OM: registers per thread for the slot table's blind prefix (OM x threads slots: 1 -- at most kBlindSlots of them -- for worlds
// of a few dozen objects; 4 for the instance whose maps stay in HBM: a 256x256 world holds ~750, and what the prefix misses
// costs a second memory round trip behind a barrier at the head of every step)
// CK: registers per thread for the two chunk tables (2 for worlds of up to 512 chunks -- 256x256: 484 --, whose tables' second half
// used to come by stage_rest's loop: two more memory round trips in a row at the head of every step, 5 k of its 10 k clocks: round 6)
template <class W, int OM = 1, int CK = 1>
struct EnvStage {
  static constexpr int M = W::kThreads >= 256 ? 1 : (256 + W::kThreads - 1) / W::kThreads;
  static constexpr int kObjRegs = OM > M ? OM : M;
  uint32_t rec[M];
  uint32_t rules[M];
  vec16 mat[M];
  vec16 mt[M];   // 624 words = 156 x 16 B: one vector load per thread of 256
  uint16_t chunk_order[CK];
  uint8_t chunk_seen[CK];
  int32_t census[M];
  vec16 objs[kObjRegs];
  uint64_t win[(kWinX * kWinY / 8 + W::kThreads - 1) / W::kThreads];   // LaneSlots: the material window, 8 bytes per load
};
#ifndef CRAFTER_BLIND_SLOTS
#define CRAFTER_BLIND_SLOTS 64
#endif
constexpr int kBlindSlots = CRAFTER_BLIND_SLOTS;   // the slot table's length is in the record that is still in flight: this
                                                   // many slots are fetched blindly with it

template <int OM>
__host__ __device__ constexpr int blind_slots(int max_objects, int nthreads) {
  return OM > 1 ? (max_objects < OM * nthreads ? max_objects : OM * nthreads) : (max_objects < kBlindSlots ? max_objects : kBlindSlots);
}
template <class W, class S, int OM, int CK>
__device__ __forceinline__ void load_env_issue(Env<W, S>& e, const StatePtrs& st, int env, int everything, EnvStage<W, OM, CK>& q) {
  const Config& c = e.cfg;
  W& w = e.w;
  int cells = c.W * c.H;
  int nch = c.nchunk_x * c.nchunk_y;
  uint32_t ppos = 0;
  if constexpr (Env<W, S>::kLane) {
    // The window is centred on the player, whose position sits in the slot table that is about to be fetched: ONE word
    // of it is asked for first (loads return in order: it is the shortest wait there is) and the window's loads join
    // the others once it is known.
    if (everything) ppos = ((const uint32_t*)(st.objs + (size_t)env * c.max_objects + 1))[1];
  }
  stage_issue(w, q.rec, (const uint32_t*)(st.rec + env), (int)(sizeof(EnvRec) / 4));
  if (e.rules_staged) stage_issue(w, q.rules, (const uint32_t*)e.tb.rules, CRAFTER_RULES_HEAD_BYTES / 4);
  if (!everything) return;
  if constexpr (!Env<W, S>::kLane)
    if (e.mat != e.g_mat && cells % 16 == 0) stage_issue(w, q.mat, (const vec16*)e.g_mat, cells / 16);
  stage_issue(w, q.mt, (const vec16*)(st.mt + (size_t)env * MT_N), MT_N / 4);
  stage_issue(w, q.chunk_order, st.chunk_order + (size_t)env * nch, nch);
  stage_issue(w, q.chunk_seen, st.chunk_seen + (size_t)env * nch, nch);
  if (!e.census_global) stage_issue(w, q.census, st.census + (size_t)env * nch * 5, nch * 5);
  if constexpr (Env<W, S>::kFar)
    e.far_issue(w.wave_index());   // (the slot table is not staged: the scan's first round, blind)
  else
    stage_issue(w, q.objs, (const vec16*)(st.objs + (size_t)env * c.max_objects), blind_slots<OM>(c.max_objects, W::kThreads));
  if constexpr (Env<W, S>::kLane) {
    place_window(e, (int)(ppos & 0xFFFFu), (int)(ppos >> 16));
    window_issue(e, e.g_mat, q.win);
  }
}

// LaneSlots: where the material window sits for a player at (px, py): inside the map, rows 8-byte aligned.
template <class W, class S>
__device__ __forceinline__ void place_window(Env<W, S>& e, int px, int py) {
  const Config& c = e.cfg;
  int x0 = px - kWinR, y0 = (py - kWinR) & ~7;
  x0 = x0 < 0 ? 0 : (x0 > c.W - kWinX ? c.W - kWinX : x0);
  y0 = y0 < 0 ? 0 : (y0 > c.H - kWinY ? c.H - kWinY : y0);   // H and kWinY are multiples of 8: stays aligned
  e.win_x0 = W::uni(x0);
  e.win_y0 = W::uni(y0);
}
// ... and its loads / LDS stores: row r of the window = kWinY bytes of row win_x0 + r of `src` (a whole map, x * H + y)
template <class W, class S, int K>
__device__ __forceinline__ void window_issue(Env<W, S>& e, const uint8_t* src, uint64_t (&r)[K]) {
  constexpr int per_row = kWinY / 8, n = kWinX * per_row;
#pragma unroll
  for (int k = 0; k < K; k++) {
    int j = e.w.tid() + k * e.w.nthreads();
    j = j < n ? j : n - 1;
    int row = j / per_row, col = j - row * per_row;
    r[k] = *(const uint64_t*)(src + (size_t)(e.win_x0 + row) * e.cfg.H + e.win_y0 + 8 * col);
  }
}
template <class W, class S, int K>
__device__ __forceinline__ void window_commit(Env<W, S>& e, const uint8_t* src, const uint64_t (&r)[K]) {
  constexpr int per_row = kWinY / 8, n = kWinX * per_row;
#pragma unroll
  for (int k = 0; k < K; k++) {
    int j = e.w.tid() + k * e.w.nthreads();
    if (j < n) ((uint64_t*)e.mat)[j] = r[k];
  }
  for (int j = K * e.w.nthreads() + e.w.tid(); j < n; j += e.w.nthreads()) {   // (narrower workgroups than the registers cover: the CPU harness)
    int row = j / per_row, col = j - row * per_row;
    ((uint64_t*)e.mat)[j] = *(const uint64_t*)(src + (size_t)(e.win_x0 + row) * e.cfg.H + e.win_y0 + 8 * col);
  }
}

// FarSlot: the windows of BOTH maps (material bytes, two-byte slot ids) around the player at (px, py), 8 bytes per load;
// issued once the player's position is known, committed behind the scan's evaluation (which runs while they are in flight).
template <class W>
struct FarWindow {
  static constexpr int NM = kWinX * (kWinY / 8), NO = kWinX * (kWinY / 4);
  static constexpr int KM = (NM + W::kThreads - 1) / W::kThreads, KO = (NO + W::kThreads - 1) / W::kThreads;
  uint64_t m[KM], o[KO];
};
template <class W, class S>
__device__ __forceinline__ bool far_window_ok(const Env<W, S>& e) { return CRAFTER_FAR_WINDOW != 0 && e.cfg.W >= kWinX && e.cfg.H >= kWinY && e.cfg.H % 8 == 0; }
template <class W, class S>
__device__ __forceinline__ const uint64_t* far_window_src(const Env<W, S>& e, int j, bool slots) {
  const int per_row = slots ? kWinY / 4 : kWinY / 8;
  int row = j / per_row, col = j - row * per_row;
  size_t cell0 = (size_t)(e.win_x0 + row) * e.cfg.H + e.win_y0;
  const uint8_t* base = slots ? (const uint8_t*)e.objmap + 2 * cell0 : (const uint8_t*)e.mat + cell0;
  return (const uint64_t*)(base + 8 * col);
}
template <class W, class S>
__device__ __forceinline__ void far_window_issue(Env<W, S>& e, int px, int py, FarWindow<W>& q) {
  if (!far_window_ok(e)) return;
  place_window(e, px, py);
#pragma unroll
  for (int k = 0; k < FarWindow<W>::KM; k++) {
    int j = e.w.tid() + k * e.w.nthreads();
    q.m[k] = *far_window_src(e, j < FarWindow<W>::NM ? j : FarWindow<W>::NM - 1, false);
  }
#pragma unroll
  for (int k = 0; k < FarWindow<W>::KO; k++) {
    int j = e.w.tid() + k * e.w.nthreads();
    q.o[k] = *far_window_src(e, j < FarWindow<W>::NO ? j : FarWindow<W>::NO - 1, true);
  }
}
template <class W, class S>
__device__ __forceinline__ void far_window_commit(Env<W, S>& e, const FarWindow<W>& q) {
  if (!far_window_ok(e)) return;
  uint64_t* lm = (uint64_t*)e.wmat;
  uint64_t* lo = (uint64_t*)e.wobj;
  const int nt = e.w.nthreads();
#pragma unroll
  for (int k = 0; k < FarWindow<W>::KM; k++) {
    int j = e.w.tid() + k * nt;
    if (j < FarWindow<W>::NM) lm[j] = q.m[k];
  }
#pragma unroll
  for (int k = 0; k < FarWindow<W>::KO; k++) {
    int j = e.w.tid() + k * nt;
    if (j < FarWindow<W>::NO) lo[j] = q.o[k];
  }
  // (narrower workgroups than the registers cover: the CPU harness)
  for (int j = FarWindow<W>::KM * nt + e.w.tid(); j < FarWindow<W>::NM; j += nt) lm[j] = *far_window_src(e, j, false);
  for (int j = FarWindow<W>::KO * nt + e.w.tid(); j < FarWindow<W>::NO; j += nt) lo[j] = *far_window_src(e, j, true);
}
// the scan's second half, behind the barrier that published the record and the player's position: the windows' loads leave,
// the scan's batches are evaluated, the windows land in LDS.  Ends on a barrier.
template <class W, class S>
__device__ __forceinline__ void far_finish_scan(Env<W, S>& e) {
  FarWindow<W> fw;
  const uint32_t pp = e.nctr[2];
  far_window_issue(e, (int)(pp & 0xFFFFu), (int)(pp >> 16), fw);
  e.far_rounds(e.nobj);
  far_window_commit(e, fw);
  // (read BETWEEN the scan's two barriers: behind the second one the rule wave is on its way and may touch a new chunk before a
  // slower wave has looked -- that wave would then believe nothing changed and keep its share of the chunk tables to itself)
  e.far_chunks_staged = e.rec->nchunks_seen;
  e.w.sync();
  e.far_done();
}

// mt_copy: a second LDS home for the MT19937 state as staged (the noise look-ahead twists its own copy: noise_chain)
template <class W, class S, int OM, int CK>
__device__ __forceinline__ void load_env_commit(Env<W, S>& e, const StatePtrs& st, int env, int everything, const EnvStage<W, OM, CK>& q,
                                       uint32_t* mt_copy = nullptr) {
  const Config& c = e.cfg;
  W& w = e.w;
  int cells = c.W * c.H;
  int nch = c.nchunk_x * c.nchunk_y;
  bool lds_maps = e.mat != e.g_mat && !Env<W, S>::kLane;   // small world: maps are LDS-resident, the slot map is derived
  if constexpr (Env<W, S>::kLane)
    if (everything) window_commit(e, e.g_mat, q.win);
  if (everything && lds_maps) {
    uint4 z;
    z.x = z.y = z.z = z.w = 0;
    if ((cells * (int)sizeof(S)) % 16 == 0) {
      uint4* lo = (uint4*)e.objmap;
      w.block_for(cells * (int)sizeof(S) / 16, [&](int i) { lo[i] = z; });
    } else {
      w.block_for(cells, [&](int i) { e.objmap[i] = 0; });
    }
  }
  stage_commit(w, q.rec, (uint32_t*)e.rec, (const uint32_t*)(st.rec + env), (int)(sizeof(EnvRec) / 4));
  if (e.rules_staged) stage_commit(w, q.rules, (uint32_t*)&e.R, (const uint32_t*)e.tb.rules, CRAFTER_RULES_HEAD_BYTES / 4);
  const int blind = blind_slots<OM>(c.max_objects, W::kThreads);
  const uint4* gob = (const uint4*)(st.objs + (size_t)env * c.max_objects);
  uint4* lob = (uint4*)e.objs;
  if (everything) {
    if (lds_maps) {
      if (cells % 16 == 0)
        stage_commit(w, q.mat, (vec16*)e.mat, (const vec16*)e.g_mat, cells / 16);
      else
        w.block_for(cells, [&](int i) { e.mat[i] = e.g_mat[i]; });
    }
    stage_commit(w, q.mt, (vec16*)e.mt, (const vec16*)(st.mt + (size_t)env * MT_N), MT_N / 4);
    if (mt_copy) stage_commit(w, q.mt, (vec16*)mt_copy, (const vec16*)(st.mt + (size_t)env * MT_N), MT_N / 4);
    stage_commit(w, q.chunk_order, e.chunk_order, (const uint16_t*)(st.chunk_order + (size_t)env * nch), nch);
    stage_commit(w, q.chunk_seen, e.chunk_seen, (const uint8_t*)(st.chunk_seen + (size_t)env * nch), nch);
    if (!e.census_global) stage_commit(w, q.census, e.census, (const int32_t*)(st.census + (size_t)env * nch * 5), nch * 5);
    if constexpr (Env<W, S>::kFar) {
      e.far_clear();
      e.far_publish();
    } else {
      stage_commit(w, q.objs, (vec16*)lob, (const vec16*)gob, blind);
    }
  }
  {   // The step counter AS STAGED, in a word nobody writes while the step runs: wave 0 stores the incremented counter into
      // the record itself, with no barrier before the other waves' first look at it (a thread that arrives late
      // read the new value and guessed the frame's step one too far).  By the thread that holds the record's word.
    // Beside it, in the word's top byte: whether the player is asleep as the step begins (Renderer::stage_rows picks the
    // frame's lit rows by it while the rules run).  Two threads, disjoint bytes.
    constexpr int kStepWord = (int)(offsetof(EnvRec, step) / 4);
    constexpr int kSleepWord = (int)(offsetof(EnvRec, sleeping) / 4);
    const int nt = w.nthreads();
    auto staged = [&](int word) -> uint32_t {
      return (word / nt < EnvStage<W, OM, CK>::M) ? q.rec[word / nt < EnvStage<W, OM, CK>::M ? word / nt : 0] : ((const uint32_t*)(st.rec + env))[word];
    };
    if (w.tid() == kStepWord % nt) w.scratch[3] = 0u;   // (Env::mark_mt_rewritten: nothing has rewritten the stream's state yet)
    if (w.tid() == kStepWord % nt) {
      uint32_t v = staged(kStepWord);   // (< 2^24: crafter_create bounds the daylight table)
      ((uint16_t*)&w.scratch[1])[0] = (uint16_t)v;
      ((uint8_t*)&w.scratch[1])[2] = (uint8_t)(v >> 16);
    }
    if (w.tid() == kSleepWord % nt) ((uint8_t*)&w.scratch[1])[3] = staged(kSleepWord) != 0u ? 1 : 0;
  }
  w.sync();
  e.mt_pos = e.rec->mt_pos;
  e.nobj = e.rec->nobj;
  e.dirty_slots = 0;
  if constexpr (Env<W, S>::kFar) {
    if (everything) far_finish_scan(e);
    return;
  }
  if (everything && e.nobj > blind) {
    // the rest of the slot table (large worlds: ~700 records behind the blind prefix): four records' loads in flight per
    // thread and round -- as a plain copy loop each thread paid one memory round trip per record, three in a row at
    // 8192 x 256x256 (the 11 k clocks of that instance's stage-in: round 5)
    stage_rest((vec16*)lob + blind, (const vec16*)gob + blind, w.tid(), w.nthreads(), e.nobj - blind);
    w.sync();
  }
  if constexpr (Env<W, S>::kLane) {
    if (everything) e.occ_rebuild();   // (a single wave: the commits above are visible to it)
  } else if (everything && lds_maps) {   // derive the slot map
    w.block_for(e.nobj, [&](int i) {
      Obj o = e.objs[i];
      if (i >= 1 && o.type != T_NONE) e.objmap[e.cidx(o.x, o.y)] = (S)i;
    });
    w.sync();
  }
}

template <class W, class S>
__device__ __forceinline__ void load_env(Env<W, S>& e, const StatePtrs& st, int env, int everything) {
  EnvStage<W> q;
  load_env_issue(e, st, env, everything, q);
  load_env_commit(e, st, env, everything, q);
}

// LDS -> HBM for the compact tables (maps are written through while the rules run)
// the slot table alone (the step kernel stores it before it draws when the LDS frame overlaps it)
template <class W, class S>
__device__ __forceinline__ void store_objs(Env<W, S>& e, const StatePtrs& st, int env) {
  if constexpr (Env<W, S>::kFar) return;   // (the table in global memory is the one the rules wrote)
  vec16* gob = (vec16*)(st.objs + (size_t)env * e.cfg.max_objects);
  const vec16* lob = (const vec16*)e.objs;
  stage_out<1>(e.w, gob, lob, e.nobj);   // (one record per thread through registers: a 64x64 world has ~30 live objects)
}

// mt_if_rewritten: the stream's state array only goes back if something rewrote it since load_env_commit cleared the mark
// (Env::mark_mt_rewritten); otherwise only its position, in the record, moved
template <class W, class S>
__device__ __forceinline__ void store_env(Env<W, S>& e, const StatePtrs& st, int env, bool with_objs = true, bool mt_if_rewritten = false) {
  const Config& c = e.cfg;
  W& w = e.w;
  int nch = c.nchunk_x * c.nchunk_y;
  if (w.leader()) {
    e.rec->mt_pos = e.mt_pos;
    e.rec->nobj = e.nobj;
  }
  w.sync();
  static_assert(offsetof(EnvRec, mt_pos) == 0, "the record's first word is the stream position");
  uint32_t* grec = (uint32_t*)(st.rec + env);
  const uint32_t* lrec = (const uint32_t*)e.rec;
  constexpr int NT = W::kThreads >= 256 ? 256 : W::kThreads;
  constexpr int KREC = ((int)(sizeof(EnvRec) / 4) + NT - 1) / NT, KMT = (MT_N / 4 + NT - 1) / NT, KCEN = NT >= 256 ? 1 : 3;
  stage_out<KREC>(w, grec, lrec, (int)(sizeof(EnvRec) / 4));
  if (with_objs) store_objs(e, st, env);
  if (!mt_if_rewritten || w.scratch[3] != 0u) stage_out<KMT>(w, (vec16*)(st.mt + (size_t)env * MT_N), (const vec16*)e.mt, MT_N / 4);   // (behind the barrier above)
  bool chunks_changed = true;
  // (behind the barrier above; -1: an adopted world.  A resident stretch stores them whatever its last step did: an earlier one may have touched a chunk)
  if constexpr (Env<W, S>::kFar) chunks_changed = !mt_if_rewritten || e.rec->nchunks_seen != e.far_chunks_staged;
  if (chunks_changed) {
    stage_out<1>(w, st.chunk_order + (size_t)env * nch, (const uint16_t*)e.chunk_order, nch);
    stage_out<1>(w, st.chunk_seen + (size_t)env * nch, (const uint8_t*)e.chunk_seen, nch);
  }
  if (!e.census_global) stage_out<KCEN>(w, st.census + (size_t)env * nch * 5, (const int32_t*)e.census, nch * 5);
}

// wave 0 owns the wave-uniform registers while the rules run; hand them to the other waves
template <class W, class S>
__device__ __forceinline__ void share_registers(Env<W, S>& e) {
  if (e.w.leader()) {
    e.rec->mt_pos = e.mt_pos;
    e.rec->nobj = e.nobj;
    if (e.count_twists) e.w.scratch[2] = (uint32_t)e.rng_twists | ((uint32_t)(e.mat_dirty != 0) << 16);
  }
  e.w.sync();
  e.mt_pos = e.rec->mt_pos;
  e.nobj = e.rec->nobj;
  if (e.count_twists) {
    e.rng_twists = (int)(e.w.scratch[2] & 0xFFFFu);
    e.mat_dirty = (int)(e.w.scratch[2] >> 16);
  }
  e.rng_invalidate();
}

// Night noise, generated AHEAD.  A night frame draws 2 x 63 x 49 = 6174 words of the env's MT19937 stream behind whatever
// the step's rules drew (engine.py:208-209): ten regenerations of the 624-word state, each depending on the one before.
// Drawn inside the frame that is a serial chain in the middle of it -- an epoch's pixels cannot be shaded before its state
// exists, one wave regenerates while the others wait at a barrier per epoch: 28 k clocks per night frame against 2.6 k for
// a day frame's pixels, and the launch lasts as long as its slowest envs.  But the SEQUENCE of states does not
// depend on the rules at all -- they only decide where in it the noise starts -- and while wave 0 runs the rules, the
// workgroup's other waves have nothing to do: wave 1 regenerates a COPY of the staged state kNoiseStates - 1 times and
// leaves every state in the env's scratch in global memory (30 KB, L2 / MALL); the frame then reads its pixels' words
// from there, with no order among pixels, no epochs and no barriers, and the final state goes back into LDS from the
// same scratch.  (kNoiseStates, render.hpp; a step that regenerates twice in its rules falls back to the in-frame pass.)
template <class W>
__device__ __forceinline__ void noise_chain(W& w, uint32_t* state, uint32_t* out) {   // state: LDS copy of the staged state; one wave
  W::set_priority_high();   // the rules do not wait for it, the frame does
  {
    const vec16* src = (const vec16*)state;
    vec16* dst = (vec16*)out;
    w.wave_for(MT_N / 4, [&](int i) { dst[i] = src[i]; });
    w.wsync();
  }
  for (int s_ = 1; s_ < kNoiseStates; s_++) w.mt_twist_tee(state, out + (size_t)s_ * MT_N);   // every new word straight to the scratch as well
  W::set_priority_mid();
}

// Resident steps (rollout_body): the LDS copies of the maps behind a night frame, whose pixel buffer recycled them.  The
// material map comes back from global memory (written through as the rules run), the slot table too if the buffer reached
// into it (it was stored before the frame: frame_over_objs), the slot map is derived again.  Called by every thread behind
// the barrier that opens a resident step; ends on a barrier.
template <class W, class S>
__device__ __forceinline__ void restage_maps(Env<W, S>& e, const StatePtrs& st, int env, bool with_objs) {
  const Config& c = e.cfg;
  W& w = e.w;
  int cells = c.W * c.H;
  bool lds_maps = e.mat != e.g_mat && !Env<W, S>::kLane;
  int nobj = e.rec->nobj;   // (the record is not part of what a frame recycles; the step before left the count there)
  if (!lds_maps) return;
  const uint4* gob = (const uint4*)(st.objs + (size_t)env * c.max_objects);
  uint4* lob = (uint4*)e.objs;
  if (cells % 16 == 0) {
    const uint4* gm = (const uint4*)e.g_mat;
    uint4* lm = (uint4*)e.mat;
    w.block_for(cells / 16, [&](int i) { lm[i] = gm[i]; });
  } else {
    w.block_for(cells, [&](int i) { e.mat[i] = e.g_mat[i]; });
  }
  if (with_objs) w.block_for(nobj, [&](int i) { lob[i] = gob[i]; });
  if ((cells * (int)sizeof(S)) % 16 == 0) {
    uint4 z;
    z.x = z.y = z.z = z.w = 0;
    uint4* lo = (uint4*)e.objmap;
    w.block_for(cells * (int)sizeof(S) / 16, [&](int i) { lo[i] = z; });
  } else {
    w.block_for(cells, [&](int i) { e.objmap[i] = 0; });
  }
  w.sync();
  w.block_for(nobj, [&](int i) {
    Obj o = e.objs[i];
    if (i >= 1 && o.type != T_NONE) e.objmap[e.cidx(o.x, o.y)] = (S)i;
  });
  w.sync();
}

template <class W>
__device__ inline RenderTarget obs_target(const Config& c, const TablePtrs& tb, uint8_t* obs, int env) {
  RenderTarget rt;
  rt.out = obs ? obs + (size_t)env * c.size_w * c.size_h * 3 : nullptr;
  rt.size_w = c.size_w;
  rt.size_h = c.size_h;
  rt.unit_x = c.unit_x;
  rt.unit_y = c.unit_y;
  rt.border_x = c.border_x;
  rt.border_y = c.border_y;
  rt.icon_w = c.icon_w;
  rt.icon_h = c.icon_h;
  rt.digit_w = c.digit_w;
  rt.digit_h = c.digit_h;
  rt.item_pos = tb.item_pos;
  rt.tex_tile = tb.tex_tile;
  rt.tex_icon = tb.tex_icon;
  rt.tex_digit = tb.tex_digit;
  rt.atlas = tb.atlas;
  rt.vignette = tb.vignette;
  return rt;
}

// info['semantic'] (engine.py:251-264): material ids with object cells replaced by class ids
template <class W, class S>
__device__ __forceinline__ void write_semantic(Env<W, S>& e, uint8_t* semantic, int env) {
  const Config& c = e.cfg;
  int cells = c.W * c.H;
  uint8_t* out = semantic + (size_t)env * cells;
  int base = e.R.n_materials;  // len(mat_ids) = n_materials + 1 (None), first class id = that + 0
  if constexpr (Env<W, S>::kLane) {   // no slot map: the materials first, then the objects' cells over them
    e.w.block_for(cells, [&](int i) {
      int x = i / c.H, y = i - x * c.H;
      out[i] = (uint8_t)e.mat_at(x, y);
    });
    W::drain_stores();   // the second pass rewrites bytes other lanes have just stored
    e.w.block_for(e.nobj, [&](int i) {
      Obj o = e.objs[i];
      if (i >= 1 && o.type != T_NONE) out[e.cidx(o.x, o.y)] = (uint8_t)(base + o.type);
    });
    return;
  }
  e.w.block_for(cells, [&](int i) {
    int v = e.mat[i];
    int slot = e.objmap[i];
    if (slot) v = base + e.objs[slot].type;
    out[i] = (uint8_t)v;
  });
}

// ---------------------------------------------------------------------------------------------
// World pool.  A world depends only on (seed, episode) (env.py:74), so the NEXT episode's world of
// every env is generated ahead of time by gen_body on a side stream and adopted by the step kernel
// the moment the env finishes; Env.reset's worldgen then never sits on the step's critical path.
// Whether an env adopts a pooled world or falls back to reset_body is unobservable: both run the
// same generator on the same (seed, episode).
//
// Per-step launch parameters of the protocol (host side: crafter_hip.hip).
struct StepCtl {
  int parity;          // which reset_q half this step appends to
  int gen_parity;      // which gen_q segment collects generation requests right now (-1: pool off)
  uint32_t safe_seq;   // newest generation batch whose completion the launch stream has waited on
  // Dispatch order (crafter_step_kernel only; all null: workgroup b steps env b).  A launch lasts as long as the envs
  // dispatched LAST need, and a night frame or a balance step takes twice a plain day step: with those dispatched first
  // the launch ends 8 us earlier.  Every step leaves the number of the env's next step in next_step[env];
  // one extra workgroup of every launch (block 0: dispatched first, done long before the others) sorts the envs for the
  // launch AFTER this one from what the launch BEFORE this one left there -- one step stale, nothing on the critical path.
  uint32_t* noise_raw = nullptr;    // [N][kNoiseStates * 624] the MT19937 states a night frame's noise comes from, generated ahead of
                                    // the rules (noise_chain); null: the frame regenerates them itself, epoch by epoch
  uint32_t* night_px = nullptr;     // [N][frame_night_px_words] scratch for a night frame's pixels (instances whose layout keeps none in LDS)
  int early_frame = 0;              // 1: the waves behind the first one draw the material half of a day frame while the object loop runs (render.hpp
                                    // early_frame): set by the host for crafter_step_early_kernel (batches of at least 2048 envs, where a shorter day
                                    // step is a shorter launch; where all envs are resident at once the launch ends with its night frames, which gain
                                    // nothing: profiles/r6_early_frame_ab.txt)
  const int32_t* order = nullptr;   // [N] workgroup b + 1 steps env order[b]
  int32_t* order_build = nullptr;   // [N] the order the next launch will use, written by block 0 of this one
  int32_t* next_step = nullptr;     // [N]
};

// The pool runs TWO worlds ahead of every env (its two entries, by episode parity): when the env enters episode k it
// asks for every world up to k + 2 that has not been asked for yet -- in steady state exactly one, world k + 2, which is
// then due a whole episode later.  One world ahead was not enough: an episode shorter than the generation latency
// (a few dozen steps) found its successor unfinished and paid an inline regeneration on the launch stream.
// gen_latest[env] = newest episode requested so far; a request is still wanted while it is one of the newest two.
// (gen_latest, PoolHdr.pending and PoolHdr.ready cross concurrently running kernels: W::agent_load / agent_store -- VERDICT r3 1c)
template <class W>
__device__ inline bool gen_wanted(const StatePtrs& st, int env, int episode) { return episode >= W::agent_load(st.gen_latest + env) - 1; }

// one segment of the request ring: count (+3 pad), then up to gen_q_capacity (env, episode) pairs
__host__ __device__ inline int gen_q_capacity(const Config& c) { return 2 * c.num_envs; }
__host__ __device__ inline size_t gen_q_stride(const Config& c) { return (size_t)4 * c.num_envs + 4; }

// pool entry of (env, episode): two entries per env, by episode parity
__device__ inline size_t pool_slot(const Config& c, int env, int episode) {
  return (size_t)(episode & 1) * c.num_envs + env;
}

template <class W>
__device__ __forceinline__ void request_generation(W& w, const Config& cfg, const StatePtrs& st, int gen_parity, int env,
                                          int upto) {
  if (gen_parity < 0 || !st.gen_q || !w.leader()) return;
  int32_t* q = st.gen_q + (size_t)gen_parity * gen_q_stride(cfg);
  int have = W::agent_load(st.gen_latest + env);
  for (int episode = (have + 1 > upto - 1) ? have + 1 : upto - 1; episode <= upto; episode++) {
    // One writer per pool entry at a time: worlds of equal episode parity share an entry.  Normally the older
    // one has long been adopted when the newer one is asked for; after an inline regeneration (the older world was not
    // ready in time) it may still sit in a batch in flight -- then the request is put off (not recorded as requested: the
    // env's next reset asks again), instead of letting two batches write one entry side by side.
    PoolHdr* hdr = st.pool_hdr + pool_slot(cfg, env, episode);
    if (W::agent_load(&hdr->pending) != 0) break;
    int k = w.global_add(q, 1);
    if (k >= gen_q_capacity(cfg)) break;   // full segment (an env would have to reset several times within one batch
                                           // period): not recorded as requested, asked for again at the next reset
    q[4 + 2 * k] = env;
    q[4 + 2 * k + 1] = episode;
    W::agent_store(&hdr->pending, (int32_t)episode);
    W::agent_store(st.gen_latest + env, (int32_t)episode);
  }
}

// the entry already holds this very world, finished (Env.reset generated it itself, crafter_reset_kernel): a queued
// duplicate must not write it again in place -- the entry may be adopted any moment
template <class W>
__device__ inline bool gen_done_already(const Config& c, const StatePtrs& st, int env, int episode) {
  uint64_t r = W::agent_load(&st.pool_hdr[pool_slot(c, env, episode)].ready);
  return (uint32_t)r == (uint32_t)episode && (r >> 32) != 0;
}
// a request is through its batch (generated, superseded or a duplicate): the entry may take the next one
template <class W>
__device__ inline void gen_retire(const Config& c, const StatePtrs& st, int env, int episode) {
  PoolHdr* h = st.pool_hdr + pool_slot(c, env, episode);
  if (W::agent_load(&h->pending) == episode) W::agent_store(&h->pending, (int32_t)0);
}

// true if the pool holds exactly the world `episode` of this env AND its generation batch is known
// complete on the launch stream (ready is one 8-byte word: batch sequence << 32 | episode)

template <class W>
__device__ inline bool pool_ready(const Config& c, const StatePtrs& st, int env, int episode, uint32_t safe_seq) {
  if (!st.pool_hdr) return false;
  uint64_t r = W::agent_load(&st.pool_hdr[pool_slot(c, env, episode)].ready);
  uint32_t seq = (uint32_t)(r >> 32), ep = (uint32_t)r;
  return ep == (uint32_t)episode && seq != 0 && seq <= safe_seq;
}

// Env.reset with a pre-generated world: copies the pool entry into the live state (LDS + HBM).
template <class W, class S>
__device__ __forceinline__ void adopt_world(Env<W, S>& e, const StatePtrs& st, int env, int episode) {
  const Config& c = e.cfg;
  W& w = e.w;
  int cells = c.W * c.H;
  int nch = c.nchunk_x * c.nchunk_y;
  size_t slot = pool_slot(c, env, episode);
  const PoolHdr hdr = st.pool_hdr[slot];
  const uint8_t* pm = st.pool_mat + slot * cells;
  w.sync();
  bool lds_maps = e.mat != e.g_mat && !Env<W, S>::kLane;
  if constexpr (Env<W, S>::kLane) {   // the map goes HBM -> HBM; the window is cut out of the pool entry (read-only)
    const uint4* src = (const uint4*)pm;
    uint4* gm = (uint4*)e.g_mat;
    w.block_for(cells / 16, [&](int i) { gm[i] = src[i]; });
  } else if (cells % 16 == 0) {
    const uint4* src = (const uint4*)pm;
    uint4* lm = (uint4*)e.mat;
    uint4* gm = (uint4*)e.g_mat;
    w.block_for(cells / 16, [&](int i) {
      uint4 v = src[i];
      if (lds_maps) lm[i] = v;
      gm[i] = v;
    });
    uint4 z;
    z.x = z.y = z.z = z.w = 0;
    uint4* lo = (uint4*)e.objmap;
    uint4* go = (uint4*)e.g_objmap;
    if (lds_maps) w.block_for(cells * (int)sizeof(S) / 16, [&](int i) { lo[i] = z; });
    else w.block_for(cells / 8, [&](int i) { go[i] = z; });
  } else {
    if constexpr (!Env<W, S>::kLane)
      w.block_for(cells, [&](int i) {
        uint8_t v = pm[i];
        e.mat[i] = v;
        e.g_mat[i] = v;
        e.objmap[i] = 0;
        if (!lds_maps) e.g_objmap[i] = 0;
      });
  }
  const uint4* pmt = (const uint4*)(st.pool_mt + slot * MT_N);
  uint4* lmt = (uint4*)e.mt;
  w.block_for(MT_N / 4, [&](int i) { lmt[i] = pmt[i]; });
  const uint16_t* pco = st.pool_chunk_order + slot * nch;
  w.block_for(nch, [&](int i) {
    e.chunk_order[i] = pco[i];
    e.chunk_seen[i] = 0;
  });
  w.sync();
  const uint4* po = (const uint4*)(st.pool_objs + slot * c.max_objects);
  uint4* lob = (uint4*)e.objs;
  w.block_for(hdr.nobj, [&](int i) {
    uint4 rec16 = po[i];
    lob[i] = rec16;
    if constexpr (!Env<W, S>::kLane) {
      if (i >= 1) {
        Obj o;   // from the value just copied: reading e.objs[i] back through another type would race the uint4 store
        __builtin_memcpy(&o, &rec16, sizeof(Obj));
        int ci = e.cidx(o.x, o.y);
        e.objmap[ci] = (S)i;
        if (!lds_maps) e.g_objmap[ci] = (uint16_t)i;
      }
    }
  });
  w.block_for(hdr.nchunks_seen, [&](int i) { e.chunk_seen[e.chunk_order[i]] = 1; });
  if constexpr (Env<W, S>::kLane) {
    w.sync();
    Obj p = e.objs[1];   // the new world's player (worldgen puts it at the centre of the map)
    place_window(e, p.x, p.y);
    uint64_t win[(kWinX * kWinY / 8 + W::kThreads - 1) / W::kThreads];
    window_issue(e, pm, win);
    window_commit(e, pm, win);
  }
  // the grass / path counts per chunk came with the world (the generator counted them)
  const int32_t* pcs = st.pool_census + slot * nch * 5;
  w.block_for(nch * 5, [&](int i) { e.census[i] = pcs[i]; });
  e.begin_episode(episode);
  if (w.leader()) {
    e.rec->nchunks_seen = hdr.nchunks_seen;
    e.rec->status |= (uint32_t)hdr.pad;   // e.g. object-table overflow while generating
    if ((uint32_t)hdr.ready != (uint32_t)episode) e.rec->status |= ST_POOL_MISMATCH;   // scheduler invariant broken
  }
  e.mt_pos = hdr.mt_pos;
  e.mark_mt_rewritten();
  e.nobj = hdr.nobj;
  e.dirty_slots = 0;
  if constexpr (Env<W, S>::kFar) {   // the cache held the old world's records: only the new player's entry stands (the frame reads the rest from the table)
    w.block_for(64, [&](int i) { ((uint32_t*)e.ndm)[i] = 0u; });
    if (w.leader()) {
      uint4 p1 = po[1];
      p1.w = 1u;
      *(uint4*)&e.ncache[0] = p1;
      e.ndm[1] = 1;
    }
    e.far_live = hdr.nobj - 1;
    e.far_chunks_staged = -1;
    e.win_x0 = e.win_y0 = -(1 << 20);   // (the windows showed the old world: the frame reads the new maps)
    e.cur_slot = -1;
    e.cur_idx = -1;
    W::drain_stores();   // (the table's new records are read straight from the coherence point: Env::obj_rd_lane)
  }
  w.sync();
  e.occ_rebuild();
}

// ---------------------------------------------------------------------------------------------
// Split step (default geometry, default rules; crafter_hip.hip): the serial rule half of a step runs one WAVE per env
// (a 64-thread workgroup with 15 KB of LDS, ten to a CU), the frame half four waves per env with the renderer's tables but
// none of the env's maps.  What passes between them, besides the state itself (record, MT19937 state): the frame record,
// kFrameRecordBytes per env in the env's slice of StatePtrs.objmap (unused for LDS-resident worlds, whose slot map is
// derived state): for every cell of the view its material id (0xFF: outside the map) and its sprite's texture id (0xFF: none).
__device__ inline uint8_t* frame_record(const StatePtrs& st, const Config& c, int env) {
  return (uint8_t*)(st.objmap + (size_t)env * c.W * c.H);
}

// staging: kFrameRecordBytes of LDS the caller no longer needs once the view's materials have been read (LaneSlots: the window).
// returns whether the frame is a night frame
template <class W, class S>
__device__ __forceinline__ bool emit_frame_cells(Env<W, S>& e, const StatePtrs& st, int env, uint8_t* staging = nullptr, int hint_step = -1,
                                        double hint_D = 0.0) {
  const Config& c = e.cfg;
  W& w = e.w;
  uint8_t* rec = frame_record(st, c, env);
  Obj p = e.objs[1];
  int offx = c.local_gw / 2, offy = c.local_gh / 2;
  int ncell = c.local_gw * c.local_gh;
  bool sleeping = e.rec->sleeping != 0;
  if constexpr (Env<W, S>::kLane) {
    // (one wave) the record is put together in LDS -- lane registers first, the window is only overwritten when every
    // material has been read out of it -- and leaves in 8-byte stores, one per lane
    int px = p.x, py = p.y;
    w.lane_set(1, 0, ncell, [&](int k, int) -> uint32_t {
      int gx = k / c.local_gh, gy = k - gx * c.local_gh;
      int wx = px + gx - offx, wy = py + gy - offy;
      return (uint32_t)(e.inside(wx, wy) ? e.mat_at(wx, wy) : 0xFF);
    });
    w.lane_set(0, 0, ncell, [&](int, int) -> uint32_t { return 0xFFu; });
    w.occ_groups(
        e.nobj,
        [&](uint32_t pos) {
          int dx = (int)(pos & 0xFFFFu) - px + offx, dy = (int)(pos >> 16) - py + offy;
          return pos != 0xFFFFFFFFu && (unsigned)dx < (unsigned)c.local_gw && (unsigned)dy < (unsigned)c.local_gh;
        },
        [&](int g, uint64_t m) {
          while (m) {
            int slot = 64 * g + __builtin_ctzll(m);
            m &= m - 1;
            Obj o = e.objs[slot];
            w.lane_put(0, ((int)o.x - px + offx) * c.local_gh + ((int)o.y - py + offy), (uint32_t)sprite_texture(o, sleeping));
          }
        });
    int step = e.rec->step;
    double D = step == hint_step ? hint_D : e.tb.daylight[step];
    w.wsync();
    w.lanes(0, ncell, [&](int k, int lane) {
      staging[k] = (uint8_t)w.lane_get(1, lane);
      staging[kFrameSprites + k] = (uint8_t)w.lane_get(0, lane);
    });
    w.lanes(0, MAX_ITEMS, [&](int i, int) { staging[kFrameInventory + i] = (uint8_t)e.rec->inv[i]; });
    if (w.leader()) {
      staging[kFrameFlag] = 0;
      staging[kFrameSleeping] = (uint8_t)sleeping;
      *(double*)(staging + kFrameDaylight) = D;
      *(int32_t*)(staging + kFrameStep) = step;
      *(int32_t*)(staging + kFrameMtPos) = e.mt_pos;
      for (int i = kFrameMtPos + 4; i < kFrameRecordBytes; i += 4) *(int32_t*)(staging + i) = 0;
    }
    w.wsync();
    w.lanes(0, kFrameRecordBytes / 8, [&](int i, int) { ((uint64_t*)rec)[i] = ((const uint64_t*)staging)[i]; });
    return D < 0.5;
  } else {
    w.block_for(ncell, [&](int k) {
      int gx = k / c.local_gh, gy = k - gx * c.local_gh;
      int wx = (int)p.x + gx - offx, wy = (int)p.y + gy - offy;
      int m = 0xFF, sp = 0xFF;
      if (e.inside(wx, wy)) {
        int ci = e.cidx(wx, wy);
        m = e.mat[ci];
        int slot = e.objmap[ci];
        if (slot) sp = sprite_texture(e.objs[slot], sleeping);
      }
      rec[k] = (uint8_t)m;
      rec[kFrameSprites + k] = (uint8_t)sp;
    });
  }
  w.block_for(MAX_ITEMS, [&](int i) { rec[kFrameInventory + i] = (uint8_t)e.rec->inv[i]; });
  if (w.leader()) {
    rec[kFrameFlag] = 0;
    rec[kFrameSleeping] = (uint8_t)sleeping;
    int step = e.rec->step;
    *(double*)(rec + kFrameDaylight) = e.tb.daylight[step];
    *(int32_t*)(rec + kFrameStep) = step;
    *(int32_t*)(rec + kFrameMtPos) = e.mt_pos;
  }
  return e.tb.daylight[e.rec->step] < 0.5;
}

template <class W, class S = uint16_t>
__device__ __forceinline__ int reset_body(W& w, uint8_t* smem, int env, const Config& cfg, const TablePtrs& tb, const StatePtrs& st,
                                 uint8_t* obs, int gen_parity);

// LDS of the frame kernel: record | MT state | second MT state | frame record | night pixel buffer | renderer region
struct FrameLayout {
  int rec, mt, mtb, cells, pix, pix_bytes, scratch, render, total;
};
// (a night frame's pixel buffer is the env's scratch in global memory: frame_night_px_words per env)
__host__ __device__ inline int frame_night_px_words(const Config& c) { return align16(4 * c.local_gw * c.unit_x * c.local_gh * c.unit_y) / 4; }
__host__ __device__ inline FrameLayout frame_layout(const Config& c) {
  FrameLayout F;
  int o = 0;
  F.rec = o;    o += align16((int)sizeof(EnvRec));
  F.mt = o;     o += align16(4 * MT_N);
  F.mtb = o;    o += align16(4 * MT_N);
  F.cells = o;  o += kFrameRecordBytes;
  F.pix = o;    F.pix_bytes = 0;
  F.scratch = o; o += 16;
  F.render = o; o += align16(render_lds_bytes(c));
  F.total = o;
  return F;
}

// The frame half of a split step: env.py:96 obs = self._obs() from the frame record, the env's record (step, sleeping,
// inventory) and -- at night -- its MT19937 stream, which it advances and stores back.
template <class W>
__device__ __forceinline__ void frame_body(W& w, uint8_t* smem, int env, const Config& cfg, const TablePtrs& tb, const StatePtrs& st,
                                  uint8_t* obs, uint32_t* night_px) {
  W::set_priority_mid();
  FrameLayout F = frame_layout(cfg);
  w.scratch = (uint32_t*)(smem + F.scratch);
  Env<W, uint8_t> e(w, cfg, tb, typename Env<W, uint8_t>::DefaultRulesTag{});
  e.mat = nullptr;
  e.objmap = nullptr;
  e.objs = nullptr;
  e.g_mat = nullptr;
  e.g_objmap = nullptr;
  e.rec = (EnvRec*)(smem + F.rec);
  e.mt = (uint32_t*)(smem + F.mt);
  RenderTarget rt = obs_target<W>(cfg, tb, obs, env);
  Renderer<W, uint8_t> r(e, rt, smem + F.render, (uint32_t*)(smem + F.mtb), (uint8_t*)(night_px + (size_t)env * frame_night_px_words(cfg)));
  r.pix_global = true;
  r.frame_cells = smem + F.cells;
  const uint8_t* cells = smem + F.cells;
  uint64_t* prof = st.prof ? st.prof + (size_t)env * 16 : nullptr;   // stamps 14 / 15 / (renderer: 7, 8, 12, 13) / 6: start, staged, ..., done
  r.prof = prof;
  if (prof && w.leader()) prof[14] = w.clock();
  {   // stage-in: the frame record and the static tables
    uint32_t qcells[1];
    typename Renderer<W, uint8_t>::Preload qr;
    const uint32_t* gcells = (const uint32_t*)frame_record(st, cfg, env);
    stage_issue(w, qcells, gcells, kFrameRecordBytes / 4);
    r.preload_issue(qr);
    stage_commit(w, qcells, (uint32_t*)(smem + F.cells), gcells, kFrameRecordBytes / 4);
    r.preload_commit(qr);
    w.sync();
  }
  if (cells[kFrameFlag]) return;   // (uniform: the whole workgroup leaves) the regeneration kernel draws this env's frame
  if (prof && w.leader()) prof[15] = w.clock();
  int step = *(const int32_t*)(cells + kFrameStep);
  double D = *(const double*)(cells + kFrameDaylight);
  bool sleeping = cells[kFrameSleeping] != 0;
  bool night = D < 0.5;
  // the few fields of the env's record a frame reads
  if (w.leader()) {
    e.rec->step = step;
    e.rec->sleeping = sleeping;
    e.rec->mt_pos = *(const int32_t*)(cells + kFrameMtPos);
  }
  w.block_for(MAX_ITEMS, [&](int i) { e.rec->inv[i] = cells[kFrameInventory + i]; });
  if (night) {   // only a night frame needs the stream
    const uint4* gmt = (const uint4*)(st.mt + (size_t)env * MT_N);
    uint4* lmt = (uint4*)e.mt;
    w.block_for(MT_N / 4, [&](int i) { lmt[i] = gmt[i]; });
  }
  w.sync();
  e.mt_pos = e.rec->mt_pos;
  e.nobj = 0;
  r.render(true, step, D);
  if (night) {   // it consumed noise: the stream goes back
    w.sync();
    uint4* gmt = (uint4*)(st.mt + (size_t)env * MT_N);
    const uint4* lmt = (const uint4*)e.mt;
    w.block_for(MT_N / 4, [&](int i) { gmt[i] = lmt[i]; });
    if (w.leader()) st.rec[env].mt_pos = e.mt_pos;
  }
  if (prof && w.leader()) prof[6] = w.clock();
}

// What step_body returns (bits): the env finished its episode and found no world in the pool (it then sits in the
// regeneration queue and this step has not drawn its observation: reset_body will) | the frame it drew recycled the LDS
// copies of the maps (a night frame's pixel buffer: render.hpp) -- only of interest to a caller that keeps the state in LDS.
enum : uint32_t { kStepStopped = 1u, kStepMapsGone = 2u };
// `res` of a RESIDENT step (RES = 1, rollout_body): the env's state stays in LDS from one step of a workgroup to its next.
//   kResLoaded  the state is in LDS already (the step before left it there): nothing but what a frame consumes is staged again
//   kResKeep    leave it there (no write-back to global memory, unless the env stops)
//   kResMapsGone  (with kResLoaded) the step before drew a night frame over the LDS copies of the maps: they come back from
//               global memory (the material map is written through as the rules run; the slot table was stored before the frame)
enum : uint32_t { kResLoaded = 1u, kResKeep = 2u, kResMapsGone = 4u };

template <class W, int LM = -1, int RUL = 0, class S = uint16_t, int SPLIT = 0, int RES = 0>
__device__ __forceinline__ uint32_t step_body(W& w, uint8_t* smem, int env, const Config& cfg, const TablePtrs& tb, const StatePtrs& st,
                                     const int32_t* actions, uint8_t* obs, float* reward, uint8_t* done, const StepCtl& ctl, uint32_t res = 0,
                                     int action_fetched = 0);

// SPLIT 1: the rule half of a split step (crafter_rules_kernel): no frame; the frame's inputs -- what each cell of the view
// shows -- are left in the env's frame record for frame_body (crafter_frame_kernel).
template <class W, int LM, int RUL, class S, int SPLIT, int RES>   // RUL 1: the rules are kDefaultRules (compile-time constants)
__device__ __forceinline__ uint32_t step_body(W& w, uint8_t* smem, int env, const Config& cfg, const TablePtrs& tb,
                                     const StatePtrs& st, const int32_t* actions, uint8_t* obs, float* reward,
                                     uint8_t* done, const StepCtl& ctl, uint32_t res, int action_fetched) {
  static_assert(!RES || !SPLIT, "resident steps are fused steps");
  W::set_priority_mid();   // above background generation; the serial rule phase of wave 0 goes higher still
  static_assert(!Env<W, S>::kLane || (SPLIT != 0 && RUL != 0), "LaneSlots is the rule kernel's layout: split step, compiled-in rules");
  static_assert(Env<W, S>::kFar == (LM == 0 && !SPLIT), "FarSlot is big_layout's slot type: the instance whose maps and slot table stay in global memory");
  LdsLayout L = Env<W, S>::kLane ? lane_layout(cfg) : (LM == 0 && !SPLIT) ? big_layout(cfg) : lds_layout(cfg, (int)sizeof(S), SPLIT != 0, !(RUL && !SPLIT));
  w.scratch = (uint32_t*)(smem + L.scratch);
  uint64_t* prof = st.prof ? st.prof + (size_t)env * 16 : nullptr;
  auto stamp = [&](int k) {
    if (prof && w.leader()) prof[k] = w.clock();
  };
  stamp(0);
  Env<W, S> e_staged(w, cfg, tb, smem + (RUL ? 0 : L.rules));
  Env<W, S> e_const(w, cfg, tb, typename Env<W, S>::DefaultRulesTag{});
  Env<W, S>& e = RUL ? e_const : e_staged;
  bind_lds<W, LM, S>(e, smem, L, st, env);
  // (round 4's batched search for the spawn cells of a balance round -- one lane per hit, six dwords of its chunk in flight -- is the
  // register peak of the instance whose maps stay in global memory: 104 VGPRs with it, 79 without.  Since that instance is bounded
  // to 80 -- six workgroups per CU -- the search would make it spill, and a kernel that spills both vector and scalar registers is
  // where round 6 met a miscompile: off; a balance step is 8 % longer for it, configs[3] 1.1 % slower.)
  e.spawn_batched = CRAFTER_SPAWN_BATCHED != 0 && LM == 0;
#ifdef CRAFTER_BALANCE_PROBE
  e.bal_prof = prof;
#endif
  RenderTarget rt = obs_target<W>(cfg, tb, obs, env);
  // (split: no renderer region in LDS; the object only serves the pixel-less night pass of render-off configurations)
  Renderer<W, S> r(e, rt, SPLIT ? nullptr : smem + L.render, SPLIT ? nullptr : (uint32_t*)(smem + L.mtb),
                   (!SPLIT && L.frame_bytes) ? smem + L.frame : nullptr);
  if (LM == 0 && !SPLIT && ctl.night_px) {   // big_layout: a night frame's pixels wait in the env's global scratch
    r.pix = ctl.night_px + (size_t)env * frame_night_px_words(cfg);
    r.pix_global = true;
  }
  r.prof = prof;
  const bool draw_here = !SPLIT && cfg.render_obs != 0 && obs != nullptr;   // this kernel draws the frame itself
  const bool ahead_possible = draw_here && ctl.noise_raw != nullptr && !Env<W, S>::kLane && W::kThreads >= 128;
  e.count_twists = ahead_possible;
  // the material half of a day frame is drawn by the waves behind the first one while the object loop runs (render.hpp early_frame)
  // (the CPU harness runs the run-time-layout instance, LM -1: there too, so that the path is covered without a GPU)
  const bool early_possible = W::kEarlyFrame && ctl.early_frame != 0 && ahead_possible && !RES && !SPLIT && (LM == 1 || (LM == -1 && !W::kConcurrentWaves && L.maps_in_lds != 0));
  // read before the stage-in: its latency hides under it.  A resident step has no stage-in to hide it under -- the rule wave
  // would wait a whole memory round trip for its action: the caller fetched it a step ago (rollout_body)
  int action_in = (RES && (res & kResLoaded)) ? action_fetched : actions[env];
  if (RES && (res & kResLoaded)) {
    // The state is where the step before left it, the noise look-ahead's copy of the stream state included (made at the end
    // of that step).  ONE barrier -- the frame before is through with the tables and the pixel buffer -- and the rule wave is
    // on its way: it waits for no load.  What a frame consumes and the frame before spoiled is staged again by the waves
    // that would otherwise wait for the rules: the renderer's static block (a day frame lights its rows in place) -- and,
    // behind a night frame, the maps (by everybody: the rules need them).
    w.sync();
    if (res & kResMapsGone) restage_maps(e, st, env, L.frame_over_objs != 0);   // (ends on a barrier)
    if (draw_here) r.preload_beside(false);   // (the material rows: stage_rows below)
    e.mt_pos = e.rec->mt_pos;
    e.nobj = e.rec->nobj;
    e.dirty_slots = 0;
    if constexpr (Env<W, S>::kFar) {   // the player has moved, objects have come and gone: the scan again (the table is current)
      e.far_clear();
      e.far_issue(w.wave_index());
      e.far_publish();
      w.sync();
      far_finish_scan(e);
    }
  } else {   // stage-in: every load of the state and of the renderer's static tables in flight at once
    bool draw = draw_here;
    if (early_possible) w.block_for(4, [&](int i) { r.hdr[i] = 0u; });   // (the early frame's hand-shake words: ahead of the stage-in's barrier)
    EnvStage<W, 1, Env<W, S>::kFar ? 2 : 1> qs;   // (the instance whose maps stay in HBM does not stage its slot table at all: Env::far_issue; its chunk tables are long)
    typename Renderer<W, S>::Preload qr;
    load_env_issue(e, st, env, 1, qs);
    if (draw) r.preload_issue(qr, false);
    load_env_commit(e, st, env, 1, qs, ahead_possible ? r.mtb : nullptr);   // the barrier inside only needs the state ...
    if (draw) r.preload_commit(qr, false);       // ... the tables are not read before the render's own barriers
  }
  stamp(1);
  // daylight of the step about to run, fetched now so the latency hides under the rule code
  const uint32_t staged_word = w.scratch[1];   // (the counter AS STAGED, and whether the player sleeps: see load_env_commit -- NOT e.rec->step, which wave 0 is about to overwrite)
  int step_now = (int)(staged_word & 0xFFFFFFu) + 1;
  if (step_now >= cfg.n_daylight) step_now = cfg.n_daylight - 1;
  double daylight_now = tb.daylight[step_now];
  // the material rows of the frame's row table -- by day lit, for this step, straight from the table -- by the waves that
  // would otherwise wait for the rules
  if (draw_here) w.consumers([&] { r.stage_rows(step_now, daylight_now, (staged_word >> 24) != 0u); });
  // a night step whose frame this workgroup draws: wave 1 generates the frame's noise states while wave 0 runs the rules
  uint32_t* noise_out = ahead_possible ? ctl.noise_raw + (size_t)env * (kNoiseStates * MT_N) : nullptr;
  // (only wave 1 asks, with a load of its own: wave 0 must not wait for a daylight value at the head of its rule phase)
  if (ahead_possible && w.wave_is(1)) {
    if (W::agent_load((const uint64_t*)(tb.daylight + step_now)) < 0x3FE0000000000000ull)   // 0 <= daylight < 0.5, compared as bits
      noise_chain(w, r.mtb, noise_out);
  }
  const bool staged_asleep = (staged_word >> 24) != 0u;
  if (early_possible && W::kConcurrentWaves) w.consumers([&] { r.early_frame(step_now, daylight_now, staged_asleep); });   // (waits for the rule wave's signal below)
  if (w.wave0()) {
    W::set_priority_high();   // the wave-uniform rule code is the critical path of the whole workgroup
    int action = action_in;
    uint32_t bad = 0;
    if (action < 0 || action >= e.R.n_actions) {
      bad |= ST_BAD_ACTION;
      action = 0;
    }
    int step = e.rec->step + 1;  // env.py:84
    if (step >= cfg.n_daylight) {
      bad |= ST_STEP_OVERFLOW;
      step = cfg.n_daylight - 1;
    }
    if (bad) e.st(&e.rec->status, e.rec->status | bad);
    e.st(&e.rec->step, step);
    w.wsync();
    e.update_all(action, prof, [&] {         // env.py:86-89
      if (!early_possible) return;
      if (W::kConcurrentWaves) {
        e.st(&r.hdr[0], 1u);                 // Player.update has run: the other waves start on the frame
      } else {
        r.early_frame(step_now, daylight_now, staged_asleep);
      }
    });
    stamp(2);
    if (step % 10 == 0) e.balance(daylight_now);   // env.py:90-95
    e.compact();
    e.finish_step(reward + env, done + env, cfg.reward);
    if constexpr (Env<W, S>::kFar) W::drain_stores();   // (the frame's waves read table records straight from the coherence point: Env::obj_rd_lane)
    stamp(3);
    W::set_priority_mid();
  }
  share_registers(e);
  if (st.terminal && e.rec->done) {   // the finished episode's totals survive the auto-reset here
    int32_t* t = st.terminal + (size_t)env * (MAX_ACH + 4);
    w.block_for(MAX_ACH, [&](int i) { t[i] = e.rec->ach[i]; });
    if (w.leader()) {
      t[MAX_ACH + 0] = e.rec->step;
      t[MAX_ACH + 1] = e.rec->ep_dhealth;
      t[MAX_ACH + 2] = e.rec->ep_unlock_steps;
      t[MAX_ACH + 3] = e.rec->episode;
    }
    w.sync();
  }
  bool will_reset = e.rec->needs_reset != 0;
  if (will_reset) {
    int next_episode = e.rec->episode + 1;
    if (ctl.gen_parity >= 0 && pool_ready<W>(cfg, st, env, next_episode, ctl.safe_seq)) {
      adopt_world(e, st, env, next_episode);             // Env.reset from the pool
      if (st.pool_stats && w.leader()) w.global_add(st.pool_stats + 0, 1);
      stamp(6);
      request_generation(w, cfg, st, ctl.gen_parity, env, next_episode + 2);
      will_reset = false;                                // falls through to the first-frame render
    } else if (st.reset_q && w.leader()) {               // queue this env for the regeneration kernel
      int32_t* q = st.reset_q + (size_t)ctl.parity * (cfg.num_envs + 4);
      int k = w.global_add(q, 1);
      q[4 + k] = env;
      if (st.pool_stats && ctl.gen_parity >= 0) w.global_add(st.pool_stats + 1, 1);
    }
  }
  bool objs_stored = false;
  uint32_t ret = will_reset ? kStepStopped : 0u;
  if (!will_reset) {
    // env.py:96 obs = self._obs(); an env handed to reset_body gets its obs there
    if (cfg.want_semantic && st.semantic) write_semantic(e, st.semantic, env);
    // A night frame's pixel buffer reaches into the slot table: its final content goes out first.  Only then -- stores
    // issued here would be in the memory pipeline ahead of the frame's own loads, which return in order behind them.
    bool maybe_night = !(daylight_now >= 0.5) || e.rec->step != step_now;   // (an adopted world starts at step 0)
    if (!SPLIT && L.frame_over_objs && maybe_night) {
      store_objs(e, st, env);
      objs_stored = true;
    }
    // (no barrier here: share_registers / adopt_world ended on one and nothing has written LDS since)
    stamp(11);
    if (SPLIT && cfg.render_obs != 0 && obs != nullptr)
      emit_frame_cells(e, st, env, smem + L.mat, step_now, daylight_now);   // the frame kernel draws
    else {
      if (ahead_possible && daylight_now < 0.5 && e.rec->step == step_now) {   // (not a world adopted in this very step: that one is at step 0, by day)
        r.noise_raw = noise_out;
        r.noise_base = e.rng_twists * MT_N + e.mt_pos;   // where the rules left the stream, counted from the staged state's first word
      }
      // (a night frame in quad mode leaves its pixels in L.frame -- the LDS copies of the maps -- whether its noise was
      // generated ahead or not; a frame in direct mode, or one that only advances the stream, touches none of it)
      if (RES && draw_here && L.frame_bytes && !r.pix_global && (e.rec->step == step_now ? daylight_now : e.tb.daylight[e.rec->step]) < 0.5) ret |= kStepMapsGone;
      if (draw_here) r.rows_staged(step_now, daylight_now, (w.scratch[1] >> 24) != 0u);   // (the word is as it was: nothing rewrites it before the step's tail)
      // an early frame stands if every drawing wave finished it, no material changed behind it and the env is still in the
      // step it was drawn for (an adopted world is at step 0)
      bool drawn = false;
      if (early_possible && r.early_frame_drawn() && !e.mat_dirty && e.rec->step == step_now) drawn = r.finish_day_frame(daylight_now);
      if (!drawn) r.render(draw_here, step_now, daylight_now);   // may recycle the LDS map copies: keep it last
    }
  } else if (SPLIT == 1) {
    if (w.leader()) frame_record(st, cfg, env)[kFrameFlag] = 1;   // no frame from this step: the regeneration kernel draws the reset frame
  }
  // (nor here: store_env's own barrier separates the frame's LDS traffic from the write-back)
  stamp(4);
  if (ctl.next_step && w.leader()) ctl.next_step[env] = e.rec->step + 1;   // (0 + 1 in a world just adopted)
  if (RES && (res & kResKeep) && !will_reset) {
    // resident: the wave-uniform registers go back into the LDS record (the next step of this workgroup reads them there,
    // behind its first barrier), and so does the step counter as the other waves will want it (load_env_commit)
    if ((ret & kStepMapsGone) && L.frame_over_objs && !objs_stored) store_objs(e, st, env);   // (never: a night frame stored them above)
    if (w.leader()) {
      e.rec->mt_pos = e.mt_pos;
      e.rec->nobj = e.nobj;
      w.scratch[1] = (uint32_t)e.rec->step | (e.rec->sleeping != 0 ? 1u << 24 : 0u);
    }
    // ... and the stream state as the NEXT step's noise look-ahead will want it (noise_chain twists a copy while the rule
    // wave draws from the original).  Element i by the thread that wrote element i of the state if this frame put it there
    // (Renderer::render, `ahead`); every other writer of the state is a barrier away.
    if (ahead_possible) {
      const vec16* src = (const vec16*)e.mt;
      vec16* dst = (vec16*)r.mtb;
      w.block_for(MT_N / 4, [&](int i) { dst[i] = src[i]; });
    }
  } else {
    store_env(e, st, env, !objs_stored, RES == 0);   // (a resident stretch stores the state array whatever this step did: an earlier one may have rewritten it)
  }
  stamp(5);
  return ret;
}

// Open-loop rollout (crafter_step_n): T consecutive steps of ONE env by one workgroup, actions[t][env] known in advance
// (random / scripted policies, action repeat, planning over fixed action sequences) -- envs are independent, so nothing but
// the API forces all of them through step t before any starts step t + 1, and without that barrier a launch ends with the
// slowest SUM of T steps instead of T times with the slowest step.  Step t's outputs go to obs + t * obs_stride,
// reward + t * N, done + t * N.  An env that finishes an episode and finds no world in the pool (all but never) cannot go
// on here: it is queued for the regeneration kernel as in a single step and records the step it stopped at;
// requeue_rollout_body (below) regenerates it and runs the rest of its steps.
template <class W, int LM, int RUL, class S>
__device__ __forceinline__ void rollout_body(W& w, uint8_t* smem, int env, const Config& cfg, const TablePtrs& tb,
                                    const StatePtrs& st, const int32_t* actions, uint8_t* obs, float* reward, uint8_t* done,
                                    const StepCtl& ctl, int T, size_t obs_stride, int32_t* stalled_at) {
  size_t n = (size_t)cfg.num_envs;
  // The env's working set is staged in ONCE, stays in LDS over the T steps and is written back once (round 5: a store +
  // stage-in per step was 5 k of a step's 32 k clocks and 3.5 x the compulsory reads); per step only the outputs leave --
  // frame, reward, done, map cells written through -- and the renderer's static block comes in again (a day frame lights
  // its rows in place).  A night frame's pixel buffer recycles the LDS copies of the maps: they are staged again behind it.
  uint32_t carry = 0;   // what the step before left: kResLoaded | kResMapsGone
  int action_next = 0;  // the action of the step after the one running, fetched while that one runs
  uint64_t* prof = st.prof ? st.prof + (size_t)env * 16 : nullptr;   // stamps 14 / 15: the workgroup's first / last clock (the steps' own stamps: the last step's survive)
  if (prof && w.leader()) prof[14] = w.clock();
#pragma clang loop unroll(disable)
  for (int t = 0; t < T; t++) {
    // As far as the optimiser can tell every step has a new thread index and a new env: inlined into a plain loop it
    // hoists what a step computes from them out of the loop and keeps it in registers across the steps (251 VGPRs against
    // the step kernel's 71: two workgroups per CU instead of five).
    w.refresh();
    int env_t = W::opaque(env);
    uint32_t res = carry | (t + 1 < T ? (uint32_t)kResKeep : 0u);
    int action_now = action_next;
    action_next = actions[(size_t)(t + 1 < T ? t + 1 : t) * n + env_t];
    uint32_t got = step_body<W, LM, RUL, S, 0, 1>(w, smem, env_t, cfg, tb, st, actions + (size_t)t * n, obs ? obs + (size_t)t * obs_stride : nullptr,
                                                   reward + (size_t)t * n, done + (size_t)t * n, ctl, res, action_now);
    if (got & kStepStopped) {   // (its state went to global memory: the regeneration kernel takes the env from there)
      if (w.leader()) stalled_at[env_t] = t;
      return;
    }
    carry = kResLoaded | ((got & kStepMapsGone) ? (uint32_t)kResMapsGone : 0u);
  }
  if (prof && w.leader()) prof[15] = w.clock();
}

// Returns the episode the env is now in.  S: element type of the slot map in THIS kernel's LDS (2 bytes in general, 1 for
// the default geometry's frame kernel, whose workgroups have the compact layout's room).
template <class W, class S>
__device__ __forceinline__ int reset_body(W& w, uint8_t* smem, int env, const Config& cfg, const TablePtrs& tb,
                                 const StatePtrs& st, uint8_t* obs, int gen_parity) {
  LdsLayout L = sizeof(S) == 2 ? big_reset_layout(cfg) : lds_layout(cfg, (int)sizeof(S));   // (= lds_layout for LDS-resident maps)
  w.scratch = (uint32_t*)(smem + L.scratch);
  Env<W, S> e(w, cfg, tb, smem + L.rules);
  bind_lds<W, -1, S>(e, smem, L, st, env);
  const bool objs_in_place = L.objs < 0;
  uint64_t* prof = st.prof ? st.prof + (size_t)env * 16 : nullptr;
  if (prof && w.leader()) prof[8] = w.clock();
  RenderTarget rt = obs_target<W>(cfg, tb, obs, env);
  Renderer<W, S> r(e, rt, smem + L.render, (uint32_t*)(smem + L.mtb), L.frame_bytes ? smem + L.frame : nullptr);
  if (cfg.render_obs != 0 && obs != nullptr) r.preload();   // completes under the barriers below
  load_env(e, st, env, 0);
  WorldGen<W, S> wg(e, smem + L.wg);
  wg.reset_env(prof);
  e.recount_space();
  share_registers(e);
  request_generation(w, cfg, st, gen_parity, env, e.rec->episode + 2);
  if (cfg.want_semantic && st.semantic) write_semantic(e, st.semantic, env);
  if (L.frame_over_objs) store_objs(e, st, env);   // a night frame's pixel buffer reaches into the slot table
  w.sync();
  r.render(cfg.render_obs != 0 && obs != nullptr);   // may recycle the LDS map copies: keep it last
  w.sync();
  if (prof && w.leader()) prof[14] = w.clock();
  store_env(e, st, env, !L.frame_over_objs && !objs_in_place);
  if (prof && w.leader()) prof[15] = w.clock();
  return e.rec->episode;
}

// The regeneration kernel's half of a rollout: env stopped at step stalled_at[env] for want of a world.  Generate it
// (its first frame is that step's observation), run the env's remaining steps with the generic step instance, and
// should it finish another episode without a pooled world, generate that one here too.
template <class W>
__device__ __forceinline__ void requeue_rollout_body(W& w, uint8_t* smem, int env, const Config& cfg, const TablePtrs& tb,
                                            const StatePtrs& st, const int32_t* actions, uint8_t* obs, float* reward, uint8_t* done,
                                            const StepCtl& ctl, int T, size_t obs_stride, const int32_t* stalled_at) {
  StatePtrs sq = st;
  sq.reset_q = nullptr;   // an env that stops again is not queued: it is regenerated right here
  size_t n = (size_t)cfg.num_envs;
  int t = stalled_at[env];
  for (;;) {
    reset_body<W>(w, smem, env, cfg, tb, st, obs ? obs + (size_t)t * obs_stride : nullptr, ctl.gen_parity);
    w.sync();
    bool stopped = false;
    for (t = t + 1; t < T; t++) {
      stopped = (step_body<W, -1, 0, uint16_t>(w, smem, env, cfg, tb, sq, actions + (size_t)t * n, obs ? obs + (size_t)t * obs_stride : nullptr,
                                               reward + (size_t)t * n, done + (size_t)t * n, ctl) & kStepStopped) != 0;
      w.sync();
      if (stopped) {
        if (st.pool_stats && ctl.gen_parity >= 0 && w.leader()) w.global_add(st.pool_stats + 1, 1);
        break;
      }
    }
    if (!stopped) return;
  }
}

// Generates the world of (env, episode) into the pool.  Touches no live state of the env (which the
// launch stream may be stepping concurrently): reads only the immutable seed lane.
template <class W>
__device__ __forceinline__ void gen_body(W& w, uint8_t* smem, int env, int episode, uint32_t seq, const Config& cfg,
                                const TablePtrs& tb, const StatePtrs& st) {
  LdsLayout L = big_reset_layout(cfg);
  w.scratch = (uint32_t*)(smem + L.scratch);
  if (!gen_wanted<W>(st, env, episode)) return;   // superseded by newer requests of the same env
  Env<W> e(w, cfg, tb, smem + L.rules);
  bind_lds(e, smem, L, st, env);
  if (L.objs < 0) e.objs = st.pool_objs + pool_slot(cfg, env, episode) * cfg.max_objects;   // the pool entry's table, in place
  int cells = cfg.W * cfg.H;
  int nch = cfg.nchunk_x * cfg.nchunk_y;
  bool lds_maps = e.mat != e.g_mat;
  size_t slot = pool_slot(cfg, env, episode);
  e.g_mat = st.pool_mat + slot * cells;   // the generator's write-through target is the pool
  e.g_objmap = nullptr;
  if (!lds_maps) {   // large world: generate straight into the pool entry; no slot map is needed
    e.mat = e.g_mat;
    e.objmap = nullptr;
  }
  w.sync();
  {   // the rules' head (load_env does this for the other kernels)
    const uint32_t* src = (const uint32_t*)tb.rules;
    uint32_t* dst = (uint32_t*)&e.R;
    w.block_for(CRAFTER_RULES_HEAD_BYTES / 4, [&](int i) { dst[i] = src[i]; });
  }
  if (w.leader()) {
    e.rec->seed_lane = st.rec[env].seed_lane;
    e.rec->episode = episode - 1;
    e.rec->status = 0;
  }
  w.sync();
  WorldGen<W> wg(e, smem + L.wg);
  wg.reset_env(nullptr);
  share_registers(e);
  uint4* gob = (uint4*)(st.pool_objs + slot * cfg.max_objects);
  const uint4* lob = (const uint4*)e.objs;
  if (lob != gob) w.block_for(e.nobj, [&](int i) { gob[i] = lob[i]; });
  uint32_t* gmt = st.pool_mt + slot * MT_N;
  w.block_for(MT_N, [&](int i) { gmt[i] = e.mt[i]; });
  uint16_t* gco = st.pool_chunk_order + slot * nch;
  w.block_for(nch, [&](int i) { gco[i] = e.chunk_order[i]; });
  w.sync();
  e.recount_space();
  int32_t* gcs = st.pool_census + slot * nch * 5;
  w.block_for(nch * 5, [&](int i) { gcs[i] = e.census[i]; });
  w.sync();
  if (w.leader()) {
    PoolHdr* h = st.pool_hdr + slot;
    h->mt_pos = e.mt_pos;
    h->nobj = e.nobj;
    h->nchunks_seen = e.rec->nchunks_seen;
    h->pad = (int32_t)e.rec->status;
    W::agent_store(&h->ready, ((uint64_t)seq << 32) | (uint32_t)episode);
  }
}

// ---------------------------------------------------------------------------------------------
// World-pool generation as a pipeline of three kernels (same arithmetic as gen_body, which stays as the fused form
// crafter_reset_kernel appends).  Generating a world is 5 % seeding (serial: init_genrand, the OpenSimplex
// shuffle), 50 % terrain noise (parallel over cells, 128 VGPRs of f64) and 45 % ordered uniform() draws (serial, one
// wave).  In one 1024-thread workgroup the serial half holds a whole CU's registers while 15 waves sleep at a barrier;
// split by kind of parallelism, every piece is sized for what it does and fits NEXT TO the step kernel's workgroups:
//   gen_seed_body      1 wave  / world,  3.6 KB LDS
//   gen_classify_body  4 waves / world, 0.5 KB LDS, no barrier after the table load
//   gen_resolve_body   1 wave  / world, gen_resolve_lds_bytes (14 KB for 64x64)
// Hand-offs go through the pool entry itself: pool_mt (state after the seed draw), pool_perm, pool_mat (cell codes,
// then final materials), PoolHdr.mt_pos.  `ready` is stamped by the last stage only.

// wave 0's stream position -> every wave (single-wave workgroups: a no-op)
template <class W, class S>
__device__ __forceinline__ void share_position(Env<W, S>& e, W& w) {
  e.mt_pos = (int)w.bcast_from_wave0((uint32_t)e.mt_pos);
}

constexpr int kGenSeedLds = ((4 * MT_N + 15) / 16 * 16) + 1024 + 16;   // mt | perm, pg3, source, ridx | scratch

// Probe builds (-DCRAFTER_PROBES): shader clocks per phase of the generation kernels, summed over all worlds
// (crafter_debug_gen_probe reads and clears them).  [0..7] seeding, [8..15] classification, [16..31] ordered draws; the last slot
// of each group counts the worlds.
#if defined(CRAFTER_PROBES)
static __device__ unsigned long long g_gen_probe[32];
#define GEN_T0 uint64_t gen_t0_ = w.clock();
#define GEN_STAMP(k) do { if (w.leader()) { uint64_t t_ = w.clock(); atomicAdd(&g_gen_probe[k], (unsigned long long)(t_ - gen_t0_)); gen_t0_ = t_; } } while (0)
#define GEN_COUNT(k) do { if (w.leader()) atomicAdd(&g_gen_probe[k], 1ull); } while (0)
#else
#define GEN_T0
#define GEN_STAMP(k) do { } while (0)
#define GEN_COUNT(k) do { } while (0)
#endif

template <class W>
__device__ __forceinline__ void gen_seed_body(W& w, uint8_t* smem, int env, int episode, const Config& cfg, const TablePtrs& tb,
                                     const StatePtrs& st) {
  if (!gen_wanted<W>(st, env, episode) || gen_done_already<W>(cfg, st, env, episode)) return;   // superseded by a newer request of the same env / a duplicate
  Env<W> e(w, cfg, tb);
  e.mt = (uint32_t*)smem;
  e.objmap = nullptr;
  e.g_objmap = nullptr;
  w.scratch = (uint32_t*)(smem + kGenSeedLds - 16);
  WorldGen<W> wg(e, smem + align16(4 * MT_N));
  uint32_t wseed = world_seed(st.rec[env].seed_lane, (uint64_t)episode);   // env.py:74
  GEN_T0
  if (w.wave0()) wg.init_mt(wseed);
  e.mt_pos = MT_N;
  e.rng_invalidate();
  w.sync();
  GEN_STAMP(0);
  uint32_t sseed = 0;
  if (w.wave0()) sseed = e.randint(2147483647u);   // worldgen.py:11
  sseed = w.bcast_from_wave0(sseed);
  GEN_STAMP(1);
  wg.seed_simplex((int64_t)sseed);
  GEN_STAMP(2);
  GEN_COUNT(7);
  share_position(e, w);
  size_t slot = pool_slot(cfg, env, episode);
  uint32_t* gmt = st.pool_mt + slot * MT_N;
  w.block_for(MT_N, [&](int i) { gmt[i] = e.mt[i]; });
  uint32_t* gp = (uint32_t*)(st.pool_perm + slot * 512);
  const uint32_t* lp = (const uint32_t*)wg.perm;   // perm[256] and pg3[256] are adjacent
  w.block_for(128, [&](int i) { gp[i] = lp[i]; });
  if (w.leader()) st.pool_hdr[slot].mt_pos = e.mt_pos;
}

// A world's cells are independent: `parts` workgroups share one world (part p classifies cells [p, p + 1) * cells / parts),
// so a batch of few worlds still spreads over the chip and finishes in a fraction of a world's serial time.
constexpr int kGenClassifyTables = 512 + kSimplexLdsBytes;   // perm, gradient numbers | noise3's tables (simplex.hpp)
constexpr int kGenClassifyCells = 1024;   // cells per workgroup (4 per thread): rounds of a few hundred look-ups keep the lanes busy
__host__ __device__ inline int gen_classify_parts(const Config& c) {
  return (c.W * c.H + kGenClassifyCells - 1) / kGenClassifyCells;
}

template <class W>
__device__ __forceinline__ void gen_classify_body(W& w, uint8_t* smem, int env, int episode, int part, int parts, const Config& cfg,
                                         const TablePtrs& tb, const StatePtrs& st) {
  if (!gen_wanted<W>(st, env, episode) || gen_done_already<W>(cfg, st, env, episode)) return;
  size_t slot = pool_slot(cfg, env, episode);
  GEN_T0
  Env<W> e(w, cfg, tb);
  WorldGen<W> wg(e, smem);
  const uint32_t* gp = (const uint32_t*)(st.pool_perm + slot * 512);
  uint32_t* lp = (uint32_t*)smem;
  w.block_for(128, [&](int i) { lp[i] = gp[i]; });
  wg.tab = (SimplexLds*)(smem + 512);   // this kernel keeps no MT state: the tables sit right behind the permutation
  wg.fill_gradients();
  w.sync();
  Simplex<W> sx{wg.perm, wg.pg3, wg.tab};
  const typename WorldGen<W>::ClassIds ids = wg.class_ids();
  int cells = cfg.W * cfg.H;
  int px = cfg.W / 2, py = cfg.H / 2;
  uint8_t* codes = st.pool_mat + slot * cells;
  int first = part * kGenClassifyCells;
  int per = cells - first < kGenClassifyCells ? cells - first : kGenClassifyCells;
  (void)parts;
  // LDS behind the tables (perm, gradient numbers, gradients): per-cell state bytes, the round's work list, its counter
  uint8_t* state = smem + kGenClassifyTables;
  uint16_t* items = (uint16_t*)(smem + kGenClassifyTables + kGenClassifyCells);
  uint32_t* count = (uint32_t*)(smem + kGenClassifyTables + 3 * kGenClassifyCells);
  w.block_for(per, [&](int k) {
    int i = first + k;
    int x = i / cfg.H, y = i - x * cfg.H;
    int stt;
    uint8_t code = WorldGen<W>::classify_head(sx, ids, x, y, px, py, stt);
    state[k] = (uint8_t)stt;
    if ((stt & 15) == WorldGen<W>::NK_DONE) codes[i] = code;
  });
  GEN_STAMP(8);
  for (;;) {
    if (w.leader()) *count = 0;
    w.sync();
    for (int base = 64 * w.wave_index(); base < per; base += 64 * W::num_waves()) {   // dense list of the cells that need a look-up
      uint64_t m = w.ballot(base, per, [&](int k) { return (state[k] & 15) != WorldGen<W>::NK_DONE; });
      if (!m) continue;
      uint32_t at = 0;
      if (w.lane() == 0) at = w.lds_fetch_add(count, (uint32_t)__builtin_popcountll(m));
      at = (uint32_t)W::uni((int)at);
      w.lanes(base, per, [&](int k, int lane) {
        if ((m >> lane) & 1ull) items[at + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint16_t)k;
      });
    }
    w.sync();
    int n = (int)*count;
    if (n == 0) break;
    GEN_COUNT(14);
    w.block_for(n, [&](int j) {
      int k = items[j];
      int i = first + k;
      int x = i / cfg.H, y = i - x * cfg.H;
      int stt = state[k];
      double v = WorldGen<W>::classify_lookup(sx, stt, x, y);
      uint8_t code = 0;
      int nxt = WorldGen<W>::classify_advance(ids, stt, v, code);
      state[k] = (uint8_t)nxt;
      if ((nxt & 15) == WorldGen<W>::NK_DONE) codes[i] = code;
    });
    w.sync();
  }
  GEN_STAMP(9);
  GEN_COUNT(15);
}

// LDS of the classification kernel: tables | state bytes | work list | counter (for `per` cells per workgroup)
__host__ __device__ inline int gen_classify_lds_bytes(const Config&) { return kGenClassifyTables + 3 * kGenClassifyCells + 16; }

struct GenResolveLayout {
  int mat, objs, mt, wg, rec, rules, chunk_order, chunk_seen, census, scratch, total;
};
__host__ __device__ inline GenResolveLayout gen_resolve_layout(const Config& c) {
  GenResolveLayout G;
  int cells = c.W * c.H, nch = c.nchunk_x * c.nchunk_y;
  int o = 0;
  G.mat = lds_layout(c).maps_in_lds ? o : -1;
  if (G.mat >= 0) o += align16(cells);
  // large worlds: the slot table is written straight into the pool entry (worldgen only appends; 32 KB for 2048 slots would
  // make a ONE-wave workgroup hold 54 KB of LDS: three of them took a CU's LDS away from the step kernel, r4g)
  G.objs = G.mat >= 0 ? o : -1;
  if (G.objs >= 0) o += 16 * c.max_objects;
  G.mt = o;           o += align16(4 * MT_N);
  G.wg = o;           o += align16(WG_LDS_BYTES);
  G.rec = o;          o += align16((int)sizeof(EnvRec));
  G.rules = o;        o += CRAFTER_RULES_HEAD_BYTES;
  G.chunk_order = o;  o += align16(2 * nch);
  G.chunk_seen = o;   o += align16(nch);
  G.census = o;       o += align16(20 * nch);
  G.scratch = o;      o += 16;
  G.total = o;
  return G;
}

template <class W>
__device__ __forceinline__ void gen_resolve_body(W& w, uint8_t* smem, int env, int episode, uint32_t seq, const Config& cfg,
                                        const TablePtrs& tb, const StatePtrs& st) {
  if (!gen_wanted<W>(st, env, episode) || gen_done_already<W>(cfg, st, env, episode)) {
    if (w.leader()) gen_retire<W>(cfg, st, env, episode);
    return;
  }
  GenResolveLayout G = gen_resolve_layout(cfg);
  w.scratch = (uint32_t*)(smem + G.scratch);
  GEN_T0
  int cells = cfg.W * cfg.H;
  int nch = cfg.nchunk_x * cfg.nchunk_y;
  size_t slot = pool_slot(cfg, env, episode);
  Env<W> e(w, cfg, tb, smem + G.rules);
  e.g_mat = st.pool_mat + slot * cells;
  e.g_objmap = nullptr;
  e.objmap = nullptr;                                   // worldgen places every creature on its own cell: no slot map
  e.mat = G.mat >= 0 ? smem + G.mat : e.g_mat;          // large world: resolve in place in the pool entry
  e.objs = G.objs >= 0 ? (Obj*)(smem + G.objs) : st.pool_objs + slot * cfg.max_objects;
  e.mt = (uint32_t*)(smem + G.mt);
  e.rec = (EnvRec*)(smem + G.rec);
  e.chunk_order = (uint16_t*)(smem + G.chunk_order);
  e.chunk_seen = smem + G.chunk_seen;
  e.census = (int32_t*)(smem + G.census);
  {
    const uint32_t* src = (const uint32_t*)tb.rules;
    uint32_t* dst = (uint32_t*)&e.R;
    w.block_for(CRAFTER_RULES_HEAD_BYTES / 4, [&](int i) { dst[i] = src[i]; });
  }
  if (e.mat != e.g_mat) {
    if (cells % 16 == 0) {
      const uint4* src = (const uint4*)e.g_mat;
      uint4* dst = (uint4*)e.mat;
      w.block_for(cells / 16, [&](int i) { dst[i] = src[i]; });
    } else {
      w.block_for(cells, [&](int i) { e.mat[i] = e.g_mat[i]; });
    }
  }
  const uint32_t* gmt = st.pool_mt + slot * MT_N;
  w.block_for(MT_N, [&](int i) { e.mt[i] = gmt[i]; });
  w.block_for(nch, [&](int i) { e.chunk_seen[i] = 0; e.chunk_order[i] = 0; });
  e.clear_creature_counts();
  if (w.leader()) {
    e.rec->status = 0;
    e.rec->nchunks_seen = 0;
    Obj z;
    z.type = T_NONE; z.health = 0; z.fx = 0; z.fy = 0; z.x = 0; z.y = 0; z.aux = 0; z.pad = 0;
    e.objs[0] = z;
  }
  e.mt_pos = st.pool_hdr[slot].mt_pos;
  e.nobj = 1;
  e.dirty_slots = 0;
  e.rng_invalidate();
  w.sync();
  WorldGen<W> wg(e, smem + G.wg);
  int px = cfg.W / 2, py = cfg.H / 2;
  GEN_STAMP(16);
  if (w.wave0()) {
    e.obj_add(T_PLAYER, px, py, 0, 0, 1, 0);   // facing (0, 1) objects.py:72; slot 1
    wg.window_open();
    GEN_STAMP(17);
    wg.resolve_materials(cells);
    GEN_STAMP(18);
    wg.place_creatures(cells, px, py);
    GEN_STAMP(19);
  }
  w.sync();
  share_registers(e);
  w.block_for(cells, [&](int i) {   // strip the generation flags
    uint8_t m = (uint8_t)(e.mat[i] & WG_MAT_MASK);
    e.mat[i] = m;
    e.g_mat[i] = m;
  });
  w.sync();
  GEN_STAMP(20);
  e.recount_space();   // the world's grass / path counts per chunk travel with it (adopt_world copies them)
  GEN_STAMP(21);
  int32_t* gcs = st.pool_census + slot * nch * 5;
  w.block_for(nch * 5, [&](int i) { gcs[i] = e.census[i]; });
  uint4* gob = (uint4*)(st.pool_objs + slot * cfg.max_objects);
  const uint4* lob = (const uint4*)e.objs;
  if (lob != gob) w.block_for(e.nobj, [&](int i) { gob[i] = lob[i]; });
  uint32_t* gmt_out = st.pool_mt + slot * MT_N;
  w.block_for(MT_N, [&](int i) { gmt_out[i] = e.mt[i]; });
  uint16_t* gco = st.pool_chunk_order + slot * nch;
  w.block_for(nch, [&](int i) { gco[i] = e.chunk_order[i]; });
  w.sync();
  if (w.leader()) {
    PoolHdr* h = st.pool_hdr + slot;
    h->mt_pos = e.mt_pos;
    h->nobj = e.nobj;
    h->nchunks_seen = e.rec->nchunks_seen;
    h->pad = (int32_t)e.rec->status;
    W::agent_store(&h->ready, ((uint64_t)seq << 32) | (uint32_t)episode);
    gen_retire<W>(cfg, st, env, episode);
  }
  GEN_STAMP(22);
  GEN_COUNT(31);
}

// Env.render() on the current state (env.py:120-130): re-draws the frame and, like the reference,
// consumes the night noise from the env's RNG again (engine.py:208-209).
template <class W>
__device__ __forceinline__ void render_body(W& w, uint8_t* smem, int env, const Config& cfg, const TablePtrs& tb,
                                   const StatePtrs& st, uint8_t* out) {
  LdsLayout L = lds_layout(cfg);
  w.scratch = (uint32_t*)(smem + L.scratch);
  Env<W> e(w, cfg, tb, smem + L.rules);
  bind_lds(e, smem, L, st, env);
  RenderTarget rt = obs_target<W>(cfg, tb, out, env);
  Renderer<W> r(e, rt, smem + L.render, (uint32_t*)(smem + L.mtb), L.frame_bytes ? smem + L.frame : nullptr);
  if (out != nullptr) r.preload();
  load_env(e, st, env, 1);
  r.render(out != nullptr);
  w.sync();
  store_env(e, st, env, !L.frame_over_objs);   // rendering changes no object; the pixel buffer may have reached into the LDS copy
}

}  // namespace crafter
