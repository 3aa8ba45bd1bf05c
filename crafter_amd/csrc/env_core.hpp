// Per-environment world state + game rules, executed by ONE wavefront per environment.
//
// Execution model.  All mutable state of one env is staged in LDS for the duration of a kernel
// (Env<W>::mat/objmap/objs/mt/rec/...).  The game rules are inherently serial inside an env
// (every object update reads what earlier updates wrote and advances the same MT19937 stream,
// reference env.py:87-89), so they run as *wave-uniform* code: all 64 lanes execute the same
// instruction stream on the same values, LDS reads are broadcasts, and every store goes through
// st() (lane 0 only).  Wherever the reference does something data-parallel -- the distance
// filter over the object list, chunk censuses, k-th masked cell selection, slot compaction, the
// MT19937 twist -- the lanes fan out through the wave policy W (ballot / lanes / wave_for) and
// the result is folded back into uniform values with ballots and popcounts, so a branch is taken
// by the whole wave or not at all.
//
// W is the wave policy: crafter_amd/csrc/wave_gfx950.hpp on the device.  tests/hostsim provides
// a serial stand-in so the rule logic can be debugged on a machine without a GPU; that build is
// test infrastructure and is never loaded by the package.
//
// Reference semantics are cited per function (file:line into /root/reference/crafter).
#pragma once
#include "mt19937.hpp"
#include "types.hpp"

namespace crafter {

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int isign(int v) { return (v > 0) - (v < 0); }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// index of the k-th (0-based) set bit of m; m must have more than k bits set
__device__ __forceinline__ int kth_set_bit(uint64_t m, int k) {
  for (int i = 0; i < k; i++) m &= m - 1;
  return __builtin_ctzll(m);
}

// LDS footprint helpers (bytes, every section 16-byte aligned)
__device__ __host__ inline int align16(int v) { return (v + 15) & ~15; }

// Two-phase HBM -> LDS staging.  A plain copy loop compiles to load, wait, LDS store per array, i.e.
// one full memory round trip after the other (17 of them for one env's stage-in).  Instead every
// thread first ISSUES its share of all arrays into registers (stage_issue), and only then commits
// them to LDS in issue order (stage_commit; loads return in order, so the waits overlap).  K elements per
// thread go through registers; whatever an unusually large array has beyond K * nthreads is copied
// by the trailing plain loop.
// 16-byte element for staged copies (a native vector: HIP's uint4 class does not always stay in registers)
typedef uint32_t vec16 __attribute__((vector_size(16)));

template <int K, class W, class T>
__device__ __forceinline__ void stage_issue(const W& w, T (&r)[K], const T* src, int n) {   // n >= 1
#pragma unroll
  for (int k = 0; k < K; k++) {
    int i = w.tid() + k * w.nthreads();
    r[k] = src[i < n ? i : n - 1];   // clamped, not predicated: no control flow between the loads
  }
}
// what an unusually large array has beyond the staged K * nthreads elements (never the default configuration's:
// 256x256 worlds -- 484 chunks, 2420 census words): four elements' loads in flight per round
template <class T>
__device__ __forceinline__ void stage_rest(T* dst, const T* src, int first, int stride, int n) {
#pragma clang loop unroll(disable)
  for (int i = first; i < n; i += 4 * stride) {
    T v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int j = i + k * stride;
      v[k] = src[j < n ? j : n - 1];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int j = i + k * stride;
      if (j < n) dst[j] = v[k];
    }
  }
}
template <int K, class W, class T>
__device__ __forceinline__ void stage_commit(const W& w, const T (&r)[K], T* dst, const T* src, int n) {
#pragma unroll
  for (int k = 0; k < K; k++) {
    int i = w.tid() + k * w.nthreads();
    if (i < n) dst[i] = r[k];
  }
  if (n > K * w.nthreads()) stage_rest(dst, src, K * w.nthreads() + w.tid(), w.nthreads(), n);
}

// ... and the way back, LDS -> HBM: every LDS read of a thread's share is issued before the first store (a plain copy
// loop waits for each LDS read before it stores: one LDS round trip per element and thread -- for the 64-thread rule wave
// of the split / pipelined step that is a dozen round trips per write-back).  K elements per thread through registers.
template <int K, class W, class T>
__device__ __forceinline__ void stage_out(const W& w, T* dst, const T* src, int n) {
  if (n <= 0) return;
#ifndef CRAFTER_STORE_BATCH
#define CRAFTER_STORE_BATCH 0
#endif
  if constexpr (W::kThreads >= 256 && !CRAFTER_STORE_BATCH) {   // wide workgroups: an element or two per thread anyway -- the plain loop (and its registers)
    for (int i = w.tid(); i < n; i += w.nthreads()) dst[i] = src[i];
    return;
  }
  T r[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    int i = w.tid() + k * w.nthreads();
    r[k] = src[i < n ? i : n - 1];
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    int i = w.tid() + k * w.nthreads();
    if (i < n) dst[i] = r[k];
  }
  for (int i = K * w.nthreads() + w.tid(); i < n; i += w.nthreads()) dst[i] = src[i];
}

// SlotT: element type of the cell -> slot map.  uint16_t in general; the step kernel's default-geometry instance
// (max_objects == 256, LDS-resident maps) uses uint8_t: 4 KB less LDS per env, i.e. room on the CU for the background
// world generation next to five step workgroups.
//
// SlotT = LaneSlots (the rule kernel of the default instance, one wave per env): there is NO cell -> slot map.  Which
// object stands on a cell is answered by a compare + ballot over the objects' packed positions, held in lane registers
// (W::occ, one register per 64 slots): a few instructions and no memory access where the map costs an LDS round trip --
// and 4 KB of LDS per env.  And the material map is not staged whole either: `mat` is a WINDOW of it around the player
// (kWinX x kWinY cells, everything the rules of one step can touch: objects update within distance 18, env.py:87-89),
// cells outside are read from HBM (balance passes over far chunks; world adoption).  Together 16.3 -> 9.7 KB per env:
// sixteen rule waves per CU, i.e. all of a 4096-env batch resident at once.
struct LaneSlots {
  uint8_t unused;
  LaneSlots() = default;
  // (so that map-indexing code shared with the other layouts still compiles; with LaneSlots `objmap` is null and none of it runs)
  __host__ __device__ LaneSlots(int) : unused(0) {}
  __host__ __device__ operator int() const { return 0; }
};
constexpr int kWinX = 41;   // rows of the window: x in [px - 20, px + 20]
constexpr int kWinY = 48;   // bytes per row: y in [py - 20, py + 20] from an 8-byte aligned origin (<= 7 bytes of slack)
constexpr int kWinR = 20;

template <class T> struct IsLaneSlots { static constexpr bool value = false; };
template <> struct IsLaneSlots<LaneSlots> { static constexpr bool value = true; };

// SlotT = FarSlot (the step kernel of worlds whose maps stay in HBM, big_layout: 256x256 = BASELINE configs[3]): a two-byte
// slot id like uint16_t -- `objmap` is the env's slot map in global memory -- and the SLOT TABLE STAYS IN GLOBAL MEMORY too:
// `objs` points at the env's table there (2048 slots = 32 KB of LDS per workgroup until round 6: three workgroups per CU,
// 12 KB per env-step staged in and 12 KB stored back).  Every write goes straight through to it (fire-and-forget stores of the
// leader lane; the table is always current, nothing is stored back), and what a step READS of it is kept in LDS by one scan of
// all waves at stage-in (Env::far_issue / far_eval):
//   near_mask  one bit per slot: the object may be within update_dist of the player in this step (env.py:87-89: the only
//              objects whose update() runs) -- the object loop visits the set bits instead of the table;
//   ncache     the records of those slots (kFarCache entries; entry 0 = the player), found through `ndm`, a direct-mapped
//              index by the slot's low byte (a collision or an overflow only costs speed: a miss reads the table).
// Holes (T_NONE records) are squeezed out lazily (compact): only the ORDER of the live slots is observable.
struct FarSlot {
  uint16_t v;
  FarSlot() = default;
  __host__ __device__ FarSlot(int s) : v((uint16_t)s) {}
  __host__ __device__ operator int() const { return (int)v; }
};
static_assert(sizeof(FarSlot) == 2, "FarSlot is the global slot map's element");
template <class T> struct IsFarSlots { static constexpr bool value = false; };
template <> struct IsFarSlots<FarSlot> { static constexpr bool value = true; };
// the slot type of the step / rollout kernels' generic-geometry instances: LM 0 (maps in HBM) = FarSlot
template <int LM> struct StepSlot { typedef uint16_t type; };
template <> struct StepSlot<0> { typedef FarSlot type; };
#ifndef CRAFTER_SPAWN_BATCHED
#define CRAFTER_SPAWN_BATCHED 0   // 1: Env::apply_hits looks for all spawn cells of a round at once (worlds whose maps stay in global memory; see step_body)
#endif
#ifndef CRAFTER_FAR_WINDOW
#define CRAFTER_FAR_WINDOW 0   // 1: LDS windows of the two maps around the player (measured, round 6: what they save the rules and the cell table they cost the stage-in)
#endif
#ifndef CRAFTER_FAR_CACHE
#define CRAFTER_FAR_CACHE 96
#endif
#ifndef CRAFTER_FAR_HOLES
#define CRAFTER_FAR_HOLES 128
#endif
constexpr int kFarCache = CRAFTER_FAR_CACHE;    // cached records (16 B each); the object loop of a 64x64 world visits 5.4 objects per step on average.
                                                // (the CPU harness also runs a build with 3 entries: misses and overflow are then the rule)
constexpr int kFarHoles = CRAFTER_FAR_HOLES;    // holes that make a step squeeze the table (Env::compact)
static_assert(kFarCache >= 1 && kFarCache <= 255, "an index byte names a cache entry");
#ifndef CRAFTER_FAR_ROUND
#define CRAFTER_FAR_ROUND 4
#endif
constexpr int kFarRound = CRAFTER_FAR_ROUND;     // 64-record batches whose loads one wave has in flight during the scan (one dwordx4 each: four lane registers)
static_assert(kFarRound >= 1 && kFarRound <= 5, "four lane registers per batch, twenty in all");
__host__ __device__ inline int far_lds_bytes(int max_objects) {   // counters | ndm | ncache | near_mask
  return 16 + 256 + 16 * kFarCache + ((max_objects + 63) / 64 * 8 + 15) / 16 * 16;
}

template <class W, class SlotT = uint16_t>
struct Env {
  typedef SlotT Slot;
  static constexpr bool kLane = IsLaneSlots<SlotT>::value;
  static constexpr bool kFar = IsFarSlots<SlotT>::value;
  W& w;
  const Config& cfg;
  const TablePtrs& tb;
  const Rules& R;       // head of the rules (everything in front of `collect`): the LDS copy when there is one
  const Rules& RG;      // the full rules in global memory (collect / place / make tables: player actions only)
  // LDS working set
  uint8_t* mat;
  SlotT* objmap;
  Obj* objs;
  uint32_t* mt;
  EnvRec* rec;
  uint16_t* chunk_order;
  uint8_t* chunk_seen;
  int32_t* census;      // [nchunks][5]: grass cells, path cells (always current), zombies, skeletons, cows (per balance)
  // HBM mirrors written through on every map change
  uint8_t* g_mat;
  uint16_t* g_objmap;
  // wave-uniform registers
  int mt_pos;
  int rng_base = -4096;   // see next_u32()
  bool count_twists = false;   // (only the kernels that generate noise ahead ask: the counter costs the rule kernel five registers)
  int rng_twists = 0;     // regenerations of the state by next_u32() since stage-in (the night frame's noise, generated ahead
                          // of the rules from a copy of the staged state, has to know which state the rules stopped in)
  // Whether the 624-word state itself was rewritten since stage-in (a regeneration, a night frame's noise, an adopted world)
  // is kept in LDS, w.scratch[3] (a flag in a register of this struct cost every step kernel fifteen VGPRs): only then does the
  // state go back to global memory -- most steps only move the position (2.5 KB of writes per env-step otherwise: round 5).
  __device__ __forceinline__ void mark_mt_rewritten() {
    if (w.leader()) w.scratch[3] = 1u;
  }
  int nobj;
  int mat_dirty = 0;    // a material changed since Player.update ran (an arrow broke something): a frame whose material half was drawn meanwhile is drawn again (render.hpp early_frame)
  int dirty_slots;      // a slot was freed this step -> compact before the next one
  int win_x0 = 0, win_y0 = 0;   // LaneSlots: map coordinates of the material window's first cell (mat = the window)
  // FarSlot (see there): LDS of the scan -- counters ([0] cache entries in use, [1] live records counted, [2] the player's packed
  // position as staged), index, cache, near bits -- and the wave-uniform registers that go with them
  uint8_t* wmat = nullptr;      // FarSlot: LDS copies of the two maps' windows around the player (kWinX x kWinY cells at win_x0, win_y0 -- every cell
  uint16_t* wobj = nullptr;     // the rules of one step and its frame can touch); the maps themselves (`mat`, `objmap`) are the env's in global memory
  uint32_t* nctr = nullptr;
  uint8_t* ndm = nullptr;
  Obj* ncache = nullptr;        // (an entry's `pad` word holds its slot)
  uint64_t* near_mask = nullptr;
  int far_chunks_staged = -1;   // rec->nchunks_seen as staged: the chunk tables (3 bytes per chunk, 484 chunks) only go back to global memory if a step touched a new chunk
  int far_live = 0;             // live records in slots 1 .. nobj - 1 (the scan counts, World.add / remove keep it): nobj - 1 - far_live holes
  int cur_slot = -1, cur_idx = -1;   // the object whose update() is running and its cache entry (-1: none): no look-up for its own writes
  // The census lives in HBM and is not staged (the step kernel of large worlds, big_layout: 484 chunks x 20 B would be 9.7 KB of
  // LDS per env): counts change by atomic adds that return nothing -- nothing on the rule wave's chain waits for them -- and
  // the one reader, the balance pass, fetches its pairs' entries lane-parallel, 64 pairs per round trip (cen()).
  bool census_global = false;
  // apply_hits looks for all spawn cells of a round at once (maps in HBM: a memory round trip per hit otherwise); the LDS-map
  // instances keep the wave-wide scan per hit -- a handful of hits per pass there, and the batched search costs registers
  bool spawn_batched = false;

  __device__ __forceinline__ Env(W& w_, const Config& c, const TablePtrs& t) : w(w_), cfg(c), tb(t), R(*t.rules), RG(*t.rules) {}
  // lds_rules: CRAFTER_RULES_HEAD_BYTES of LDS that load_env stages the rules' head into
  __device__ __forceinline__ Env(W& w_, const Config& c, const TablePtrs& t, const uint8_t* lds_rules)
      : w(w_), cfg(c), tb(t), R(*(const Rules*)lds_rules), RG(*t.rules) {}
  // the rules are the compiled-in defaults (kDefaultRules): nothing is staged, everything folds
  struct DefaultRulesTag {};
  __device__ __forceinline__ Env(W& w_, const Config& c, const TablePtrs& t, DefaultRulesTag)
      : w(w_), cfg(c), tb(t), R(*(const Rules*)&kDefaultRules), RG(*(const Rules*)&kDefaultRules), rules_staged(false) {}
  bool rules_staged = true;   // load_env copies the rules' head into the LDS behind R

  // ------------------------------------------------------------------ leader-only stores
  template <class T, class V>
  __device__ __forceinline__ void st(T* p, V v) {
    if (w.leader()) *p = (T)v;
  }

  // ------------------------------------------------------------------ RNG
  // Serial draws are the latency chain of the rule code, so the tempering is taken off it: the
  // 64 lanes temper the next 64 words of the stream at once into a W lane register and each draw
  // is a wave broadcast (v_readlane) of one of them instead of an LDS round trip + 10 dependent ALU
  // ops.  rng_base = stream index held by lane 0, or far away when the look-ahead is stale.
  __device__ __forceinline__ uint32_t next_u32() {
    int pos = W::uni(mt_pos), base = W::uni(rng_base);   // wave-uniform by construction
    if (pos >= MT_N) {
      w.mt_twist(mt);
      mark_mt_rewritten();
      if (count_twists) rng_twists++;
      pos = 0;
      base = -4096;
    }
    int k = pos - base;
    if (k < 0 || k >= 64) {
      w.lane_set(2, pos, MT_N, [&](int i, int) -> uint32_t { return mt_temper(mt[i]); });
      base = pos;
      k = 0;
    }
    mt_pos = pos + 1;
    rng_base = base;
    return w.lane_read(2, k);
  }
  // call after anything else moved mt_pos or rewrote mt[] (stream window of worldgen, night render)
  __device__ __forceinline__ void rng_invalidate() { rng_base = -4096; }
  __device__ __forceinline__ double uniform() {
    uint32_t a = next_u32();
    uint32_t b = next_u32();
    return mt_double(a, b);
  }
  // uniform() < p for a compile-time p: `below` = mt_prob53(p) (mt19937.hpp) -- the same two words, no f64 on the chain
  __device__ __forceinline__ bool uniform_below(uint64_t below) {
    uint32_t a = next_u32();
    uint32_t b = next_u32();
    // two 32-bit scalar compares (a 64-bit one is a vector instruction on a register pair): X < below  <=>  hi < bhi || (hi == bhi && lo < blo)
    uint32_t hi = a >> 5, lo = b >> 6;   // X = hi * 2^26 + lo
    uint32_t bhi = (uint32_t)(below >> 26), blo = (uint32_t)(below & 0x3FFFFFFu);
    return hi < bhi || (hi == bhi && lo < blo);
  }
  // RandomState.randint(0, n), n >= 1 (legacy masked rejection; no draw when n == 1)
  __device__ __forceinline__ uint32_t randint(uint32_t n) {
    uint32_t rng = n - 1;
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    do {
      v = next_u32() & mask;
    } while (v > rng);
    return v;
  }

  // ------------------------------------------------------------------ World (engine.py:24-117)
  __device__ __forceinline__ bool inside(int x, int y) const { return x >= 0 && y >= 0 && x < cfg.W && y < cfg.H; }
  __device__ __forceinline__ int cidx(int x, int y) const { return x * cfg.H + y; }

  // ------------------------------------------------------------------ the slot table (FarSlot: it lives in global memory, see there)
  // cache entry of a slot, -1 if it has none (wave-uniform slot)
  __device__ __forceinline__ int far_find(int slot) const {
    if (slot == 1) return 0;
    if (slot == cur_slot) return cur_idx;
    int k = W::uni((int)ndm[slot & 255]);
    if (k == 0) return -1;
    return W::uni((int)ncache[k - 1].pad) == slot ? k - 1 : -1;
  }
  // ... of a slot that differs from lane to lane (the frame's sprite cells)
  __device__ __forceinline__ int far_find_lane(int slot) const {
    int k = ndm[slot & 255];
    if (k == 0) return -1;
    return (int)ncache[k - 1].pad == slot ? k - 1 : -1;
  }
  // the record of a slot as of now (wave-uniform slot; every lane gets the same words)
  __device__ __forceinline__ Obj obj_rd(int slot) const {
    if constexpr (kFar) {
      int k = far_find(slot);
      uint32_t words[4] = {0u, 0u, 0u, 0u};
      if (k >= 0) {
        const uint32_t* c = (const uint32_t*)&ncache[k];
        words[0] = c[0];
        words[1] = c[1];
        words[2] = c[2];
      } else {   // (straight from the coherence point, where the leader lane's stores of this very step have arrived)
        W::drain_stores();
        const uint32_t* g = (const uint32_t*)&objs[slot];
        words[0] = W::load_fresh(g + 0);
        words[1] = W::load_fresh(g + 1);
        words[2] = W::load_fresh(g + 2);
      }
      words[0] = (uint32_t)W::uni((int)words[0]);
      words[1] = (uint32_t)W::uni((int)words[1]);
      words[2] = (uint32_t)W::uni((int)words[2]);
      Obj o;
      __builtin_memcpy(&o, words, sizeof(Obj));
      return o;
    } else {
      return objs[slot];
    }
  }
  // ... of a slot that differs from lane to lane (another wave than the rule wave may ask, behind a barrier)
  __device__ __forceinline__ Obj obj_rd_lane(int slot) const {
    if constexpr (kFar) {
      int k = far_find_lane(slot);
      uint32_t words[4] = {0u, 0u, 0u, 0u};
      if (k >= 0) {
        const uint32_t* c = (const uint32_t*)&ncache[k];
        words[0] = c[0];
        words[1] = c[1];
        words[2] = c[2];
      } else {
        const uint32_t* g = (const uint32_t*)&objs[slot];
        words[0] = W::load_fresh(g + 0);
        words[1] = W::load_fresh(g + 1);
        words[2] = W::load_fresh(g + 2);
      }
      Obj o;
      __builtin_memcpy(&o, words, sizeof(Obj));
      return o;
    } else {
      return objs[slot];
    }
  }
  // FarSlot: f(record) on every copy of a slot's record -- the table, and the cache entry if the slot has one
  template <class F>
  __device__ __forceinline__ void obj_mod(int slot, F f) {
    int k = far_find(slot);
    if (w.leader()) {
      f(objs[slot]);
      if (k >= 0) f(ncache[k]);
    }
  }
  __device__ __forceinline__ void set_aux(int slot, int v) {
    if constexpr (kFar)
      obj_mod(slot, [&](Obj& o) { o.aux = v; });
    else
      st(&objs[slot].aux, v);
  }
  __device__ __forceinline__ void set_health(int slot, int v) {
    if constexpr (kFar)
      obj_mod(slot, [&](Obj& o) { o.health = (int8_t)v; });
    else
      st(&objs[slot].health, v);
  }

  // World.__getitem__ (engine.py:88-93): material id / slot, (0, 0) outside the map
  // material id of a cell of the map.  LaneSlots: from the window when the cell is in it, else from HBM
  __device__ __forceinline__ bool in_window(int x, int y) const {
    return (unsigned)(x - win_x0) < (unsigned)kWinX && (unsigned)(y - win_y0) < (unsigned)kWinY;
  }
  __device__ __forceinline__ int widx(int x, int y) const { return (x - win_x0) * kWinY + (y - win_y0); }
  __device__ __forceinline__ int mat_at(int x, int y) const {
    if constexpr (kLane) {
      if (in_window(x, y)) return mat[widx(x, y)];
      return g_mat[cidx(x, y)];
    } else if constexpr (kFar) {
      if (in_window(x, y)) return wmat[widx(x, y)];
      return mat[cidx(x, y)];
    } else {
      return mat[cidx(x, y)];
    }
  }
  // slot of the object on a cell of the map, 0 if none (wave-uniform cell)
  __device__ __forceinline__ int slot_at(int x, int y) const {
    if constexpr (kLane) {
      int s = w.occ_find((uint32_t)x | ((uint32_t)y << 16), nobj);
      return s < 0 ? 0 : s;
    } else if constexpr (kFar) {
      if (W::uni((int)in_window(x, y))) return wobj[widx(x, y)];
      int far = objmap[cidx(x, y)];
      W::keep_apart();
      return far;
    } else {
      return objmap[cidx(x, y)];
    }
  }
  // ... of a cell that differs from lane to lane (the frame's cell table)
  __device__ __forceinline__ int slot_lane(int x, int y) const {
    if constexpr (kLane) {
      return 0;
    } else if constexpr (kFar) {
      if (in_window(x, y)) return wobj[widx(x, y)];
      return objmap[cidx(x, y)];
    } else {
      return objmap[cidx(x, y)];
    }
  }
  // FarSlot: a slot-map entry changes -- the map in global memory and, for a cell inside it, the window (leader lane / own lane)
  __device__ __forceinline__ void far_put_objmap(int x, int y, int slot) {
    objmap[cidx(x, y)] = (SlotT)slot;
    if (in_window(x, y)) wobj[widx(x, y)] = (uint16_t)slot;
  }
  // mat_at for a WAVE-UNIFORM cell (the serial rule code: every lane asks for the same cell).  LaneSlots: written as
  // mat_at the compiler selects between the window's address and the map's and loads through a generic pointer -- a FLAT
  // instruction on every is_free() of the object loop, waited for on both memory counters; here the all-but-always case
  // is a plain LDS read behind a scalar branch.
  __device__ __forceinline__ int mat_at_uniform(int x, int y) const {
    if constexpr (kLane) {
      if (W::uni((int)in_window(x, y))) return mat[widx(x, y)];
      int far = g_mat[cidx(x, y)];
      W::keep_apart();   // (keeps the two loads in their own blocks: merged they become one flat load)
      return far;
    } else if constexpr (kFar) {
      if (W::uni((int)in_window(x, y))) return wmat[widx(x, y)];
      int far = mat[cidx(x, y)];
      W::keep_apart();
      return far;
    } else {
      return mat[cidx(x, y)];
    }
  }
  __device__ __forceinline__ void cell(int x, int y, int& m, int& o) const {
    if (!inside(x, y)) {
      m = 0;
      o = 0;
      return;
    }
    // (every lane read the same two bytes: said to the compiler, what is decided from them is scalar compares and scalar
    // branches instead of vector compares under saved exec masks -- the object loop's code was the latter throughout: round 5)
    m = W::uni(mat_at_uniform(x, y));
    o = W::uni(slot_at(x, y));
  }
  __device__ __forceinline__ void set_mat(int x, int y, int m) {
    mat_dirty = 1;
    int i = cidx(x, y);
    int old = mat_at(x, y);
    int32_t* cs = census + chunk_of(x, y) * 5;   // keep the per-chunk grass / path counts current
    if (census_global) {
      if (w.leader()) {
        if (old == R.mat_grass) w.global_add(cs + 0, -1);
        if (old == R.mat_path) w.global_add(cs + 1, -1);
        if (m == R.mat_grass) w.global_add(cs + 0, 1);
        if (m == R.mat_path) w.global_add(cs + 1, 1);
      }
    } else {
      if (old == R.mat_grass) st(cs + 0, cs[0] - 1);
      if (old == R.mat_path) st(cs + 1, cs[1] - 1);
      w.wsync();
      if (m == R.mat_grass) st(cs + 0, cs[0] + 1);
      if (m == R.mat_path) st(cs + 1, cs[1] + 1);
    }
    if constexpr (kLane) {
      if (in_window(x, y)) st(mat + widx(x, y), m);
      st(g_mat + i, m);
    } else if constexpr (kFar) {
      if (in_window(x, y)) st(wmat + widx(x, y), m);
      st(mat + i, m);
    } else {
      st(mat + i, m);
      if (g_mat != mat) st(g_mat + i, m);
    }
    w.wsync();
  }

  // Creature counts per chunk (census columns 2..4: zombies, skeletons, cows -- the n of env.py:160-163), kept current by
  // World.add / remove / move themselves since round 4: a balance pass reads them instead of counting the slot table (a
  // pass over 2048 slots and 1452 counters in a 256x256 world, one step in ten).  col: creature_col(type), -1 = not counted.
  __device__ __forceinline__ static int creature_col(int type) {
    return type == T_ZOMBIE ? 2 : type == T_SKELETON ? 3 : type == T_COW ? 4 : -1;
  }
  __device__ __forceinline__ void count_creature(int x, int y, int col, int delta) {
    if (col < 0) return;
    int32_t* p = census + chunk_of(x, y) * 5 + col;
    if (census_global) {
      if (w.leader()) w.global_add(p, delta);
      return;
    }
    st(p, *p + delta);
    w.wsync();
  }
  // a census entry as the balance pass reads it (per lane; the global copy straight from the device's coherence point: the
  // atomic adds of this very step went there)
  __device__ __forceinline__ int cen(int i) const { return census_global ? W::agent_load(census + i) : census[i]; }
  // a world is about to be generated into this state: no creatures yet (all waves; the caller's next barrier covers it)
  __device__ __forceinline__ void clear_creature_counts() {
    int nch_total = cfg.nchunk_x * cfg.nchunk_y;
    w.block_for(nch_total * 3, [&](int i) { census[(i / 3) * 5 + 2 + i % 3] = 0; });
  }

  // grass / path cells per chunk from scratch (after worldgen); all waves.  The creature counts are left alone.
  // src: the whole map, index x * H + y (LaneSlots holds only a window of it: the caller names the full copy)
  __device__ __forceinline__ void recount_space(const uint8_t* src = nullptr) {
    if (!src) src = mat;
    int nch_total = cfg.nchunk_x * cfg.nchunk_y;
    w.block_for(nch_total * 2, [&](int i) { census[(i >> 1) * 5 + (i & 1)] = 0; });
    w.sync();
    int grass = R.mat_grass, path = R.mat_path;
    w.block_for(cfg.W * cfg.H, [&](int i) {
      int m = src[i];
      if (m == grass || m == path) {
        int x = i / cfg.H, y = i - x * cfg.H;
        w.lds_add(&census[chunk_of(x, y) * 5 + (m == grass ? 0 : 1)], 1);
      }
    });
    w.sync();
  }
  // leader-only body of set_objmap (callers batch several stores under ONE lane-0 branch)
  __device__ __forceinline__ void put_objmap(int i, int slot) {
    if constexpr (!kLane) {
      if (objmap) objmap[i] = (SlotT)slot;                                     // null: pool generation
      if (g_objmap && (const void*)g_objmap != (const void*)objmap) g_objmap[i] = (uint16_t)slot;   // null while generating into the pool
    }
  }
  __device__ __forceinline__ void set_objmap(int x, int y, int slot) {
    int i = cidx(x, y);
    if (w.leader()) put_objmap(i, slot);
  }
  // LaneSlots: the occupancy register of `slot` (all lanes call; wave-uniform arguments)
  __device__ __forceinline__ void occ_set(int slot, int x, int y) {
    if constexpr (kLane) w.occ_put(slot, (uint32_t)x | ((uint32_t)y << 16));
  }
  __device__ __forceinline__ void occ_clear(int slot) {
    if constexpr (kLane) w.occ_put(slot, 0xFFFFFFFFu);
  }
  // all registers from the slot table (after stage-in, adoption, compaction); one wave
  __device__ __forceinline__ void occ_rebuild() {
    if constexpr (kLane) {
      w.occ_fill(nobj, [&](int i) -> uint32_t {
        Obj o = objs[i];
        return (i >= 1 && o.type != T_NONE) ? ((uint32_t)o.x | ((uint32_t)o.y << 16)) : 0xFFFFFFFFu;
      });
    }
  }
  // (x, y) is always a cell of the map: unsigned division is a multiply + shift
  __device__ __forceinline__ int chunk_of(int x, int y) const {
    return (int)((uint32_t)x / (uint32_t)CHUNK) * cfg.nchunk_y + (int)((uint32_t)y / (uint32_t)CHUNK);
  }

  // first time a chunk key receives an object it is appended to the dict (engine.py:36,57,79)
  __device__ __forceinline__ void touch_chunk(int x, int y) {
    int c = chunk_of(x, y);
    if (!W::uni((int)chunk_seen[c])) {
      int n = rec->nchunks_seen;
      if (w.leader()) {
        chunk_seen[c] = 1;
        chunk_order[n] = (uint16_t)c;
        rec->nchunks_seen = n + 1;
      }
      w.wsync();
    }
  }

  // World.add (engine.py:50-57); returns the slot or 0 when the table is full
  __device__ __forceinline__ int obj_add(int type, int x, int y, int health, int fx, int fy, int aux) {
    if (nobj >= cfg.max_objects) {
      st(&rec->status, rec->status | ST_OBJ_OVERFLOW);
      w.wsync();
      return 0;
    }
    int slot = nobj++;
    Obj o;
    o.type = (uint8_t)type;
    o.health = (int8_t)health;
    o.fx = (int8_t)fx;
    o.fy = (int8_t)fy;
    o.x = (uint16_t)x;
    o.y = (uint16_t)y;
    o.aux = aux;
    o.pad = 0;
    if constexpr (kFar) {
      if (w.leader()) {
        objs[slot] = o;
        far_put_objmap(x, y, slot);
      }
      far_live++;
    } else if (w.leader()) {
      objs[slot] = o;
      put_objmap(cidx(x, y), slot);
    }
    occ_set(slot, x, y);
    touch_chunk(x, y);
    count_creature(x, y, creature_col(type), 1);
    w.wsync();
    return slot;
  }
  // World.remove (engine.py:59-65)
  __device__ __forceinline__ void obj_remove(int slot) {
    Obj o;
    if constexpr (kFar) {
      o = obj_rd(slot);
    } else {
      const uint32_t* rw = (const uint32_t*)&objs[slot];
      uint32_t words[4] = {(uint32_t)W::uni((int)rw[0]), (uint32_t)W::uni((int)rw[1]), 0u, 0u};   // type, position: all it needs
      __builtin_memcpy(&o, words, sizeof(Obj));
    }
    obj_remove(slot, o);
  }
  // ... of an object whose record the caller holds (the object loop: no LDS round trip for what is in registers)
  __device__ __forceinline__ void obj_remove(int slot, const Obj& o) {
    if (o.type == T_NONE) return;
    if constexpr (kFar) {
      int k = far_find(slot);
      if (w.leader()) {
        far_put_objmap(o.x, o.y, 0);
        objs[slot].type = T_NONE;
        if (k >= 0) ncache[k].type = T_NONE;
      }
      far_live--;
    } else if (w.leader()) {
      put_objmap(cidx(o.x, o.y), 0);
      objs[slot].type = T_NONE;
    }
    occ_clear(slot);
    count_creature(o.x, o.y, creature_col(o.type), -1);
    dirty_slots = 1;
    w.wsync();
  }
  // World.move (engine.py:67-80) from (ox, oy), the object's position field; a no-op for an object that has removed itself
  // in this very update (alive == false: its record says T_NONE).  Written against what the caller already holds -- the
  // record is not read again: every LDS round trip here sits on the serial chain of the rule phase.
  // col: the mover's census column (creature_col of its type; -1 for the player and arrows)
  __device__ __forceinline__ void obj_move(int slot, int ox, int oy, int x, int y, bool alive, int col = -1) {
    if (!alive) return;
    if constexpr (kFar) {
      int k = far_find(slot);
      if (w.leader()) {
        far_put_objmap(x, y, slot);
        far_put_objmap(ox, oy, 0);
        ((uint32_t*)&objs[slot])[1] = (uint32_t)x | ((uint32_t)y << 16);   // (x, y: one word)
        if (k >= 0) ((uint32_t*)&ncache[k])[1] = (uint32_t)x | ((uint32_t)y << 16);
      }
    } else if (w.leader()) {
      put_objmap(cidx(x, y), slot);
      put_objmap(cidx(ox, oy), 0);
      objs[slot].x = (uint16_t)x;
      objs[slot].y = (uint16_t)y;
    }
    occ_set(slot, x, y);
    // the chunk the object leaves has been seen (the object was added or moved into it): only a new chunk key needs the look
    if (chunk_of(x, y) != chunk_of(ox, oy)) {
      touch_chunk(x, y);
      count_creature(ox, oy, col, -1);
      count_creature(x, y, col, 1);
    }
    w.wsync();
  }
  // health setter (objects.py:28-30); the player's health is inventory['health']
  __device__ __forceinline__ void damage(int slot, int amount) {
    bool is_player;
    if constexpr (kFar)
      is_player = slot == 1;   // (slot 1 is the player's, and only his)
    else
      is_player = W::uni((int)objs[slot].type) == T_PLAYER;
    if (is_player) {
      st(&rec->inv[R.item_health], imax(0, rec->inv[R.item_health] - amount));
    } else {
      if constexpr (kFar) {
        Obj t = obj_rd(slot);
        set_health(slot, imax(0, (int)t.health - amount));
      } else {
        st(&objs[slot].health, imax(0, (int)objs[slot].health - amount));
      }
      int b = slot - lane_objs_base;   // its copy in the lane registers of update_all, if it has one, is out of date
      if (b >= 0 && b < 64) lane_objs_stale |= 1ull << b;
    }
    w.wsync();
  }

  // ------------------------------------------------------------------ Object helpers (objects.py:36-65)
  __device__ __forceinline__ bool is_free(int x, int y, uint32_t walk_mask) const {
    int m, o;
    cell(x, y, m, o);
    return o == 0 && ((walk_mask >> m) & 1u);
  }
  // Object.move; (px, py) is the object's own position field (stale once it removed itself)
  __device__ __forceinline__ bool try_move(int slot, int px, int py, int dx, int dy, uint32_t walk_mask, bool alive = true, int col = -1) {
    int tx = px + dx, ty = py + dy;
    if (is_free(tx, ty, walk_mask)) {
      obj_move(slot, px, py, tx, ty, alive, col);
      return true;
    }
    return false;
  }
  __device__ __forceinline__ static void toward(int px, int py, int tx, int ty, bool long_axis, int& dx, int& dy) {
    int ox = tx - px, oy = ty - py;
    int d0 = iabs(ox), d1 = iabs(oy);
    bool horiz = long_axis ? (d0 > d1) : (d0 <= d1);
    dx = horiz ? isign(ox) : 0;
    dy = horiz ? 0 : isign(oy);
  }
  __device__ __forceinline__ void random_dir(int& dx, int& dy) {
    uint32_t k = randint(4);  // all_dirs = ((-1,0),(1,0),(0,-1),(0,1))  objects.py:33-34
    dx = (k == 0) ? -1 : (k == 1) ? 1 : 0;
    dy = (k == 2) ? -1 : (k == 3) ? 1 : 0;
  }

  // ------------------------------------------------------------------ Player (objects.py:99-261)
  __device__ __forceinline__ bool pay(const ItemList& uses) {
    for (int i = 0; i < uses.n; i++)
      if (rec->inv[uses.item[i]] < uses.amount[i]) return false;
    for (int i = 0; i < uses.n; i++) st(&rec->inv[uses.item[i]], rec->inv[uses.item[i]] - uses.amount[i]);
    w.wsync();
    return true;
  }
  __device__ __forceinline__ void bump_ach(int a) {
    st(&rec->ach[a], rec->ach[a] + 1);
    w.wsync();
  }
  __device__ __forceinline__ void add_item(int item, int amount) {
    st(&rec->inv[item], rec->inv[item] + amount);
    w.wsync();
  }

  // objects.py:181-212
  __device__ __forceinline__ void do_object(int slot) {
    int dmg = 1;
    if (rec->inv[R.item_wood_sword]) dmg = imax(dmg, 2);
    if (rec->inv[R.item_stone_sword]) dmg = imax(dmg, 3);
    if (rec->inv[R.item_iron_sword]) dmg = imax(dmg, 5);
    Obj o = obj_rd(slot);
    if (o.type == T_PLANT) {
      if (o.aux > 300) {
        set_aux(slot, 0);
        add_item(R.item_food, 4);
        bump_ach(R.ach_eat_plant);
      }
    } else if (o.type == T_ZOMBIE || o.type == T_SKELETON || o.type == T_COW) {
      int h = imax(0, (int)o.health - dmg);
      set_health(slot, h);
      w.wsync();
      if (h <= 0) {
        if (o.type == T_ZOMBIE) bump_ach(R.ach_defeat_zombie);
        if (o.type == T_SKELETON) bump_ach(R.ach_defeat_skeleton);
        if (o.type == T_COW) {
          add_item(R.item_food, 6);
          bump_ach(R.ach_eat_cow);
          st(&rec->hunger2, 0);
        }
      }
    }
  }

  // objects.py:214-229
  __device__ __forceinline__ void do_material(int tx, int ty, int material) {
    if (material == R.mat_water) st(&rec->thirst2, 0);
    const CollectRule& cr = RG.collect[material];
    if (!cr.valid) return;
    for (int i = 0; i < cr.require.n; i++)
      if (rec->inv[cr.require.item[i]] < cr.require.amount[i]) return;
    set_mat(tx, ty, cr.leaves);
    double u = uniform();  // drawn even when probability is 1
    if (u <= cr.probability) {
      for (int i = 0; i < cr.receive.n; i++) {
        add_item(cr.receive.item[i], cr.receive.amount[i]);
        bump_ach(cr.receive.ach[i]);
      }
    }
  }

  // objects.py:231-249
  __device__ __forceinline__ void place(int k, int tx, int ty, int material, int obj) {
    if (obj) return;
    const PlaceRule& pr = RG.place[k];
    if (!((pr.where_mask >> material) & 1u)) return;
    if (!pay(pr.uses)) return;
    if (pr.is_object)
      obj_add(T_PLANT, tx, ty, 1, 0, 0, 0);
    else
      set_mat(tx, ty, pr.material);
    bump_ach(pr.ach);
  }

  // objects.py:251-261; World.nearby slices mat[x-1:x+2, y-1:y+2] with numpy semantics, so the
  // window is EMPTY when x == 0 or y == 0 (negative start wraps; engine.py:95-98)
  __device__ __forceinline__ void make(int k, int px, int py) {
    const MakeRule& mk = RG.make[k];
    uint32_t near = 0;
    if (px > 0 && py > 0) {
      int x1 = imin(px + 1, cfg.W - 1), y1 = imin(py + 1, cfg.H - 1);
      for (int x = px - 1; x <= x1; x++)
        for (int y = py - 1; y <= y1; y++) near |= 1u << mat_at(x, y);
    }
    if ((near & mk.nearby_mask) != mk.nearby_mask) return;
    if (!pay(mk.uses)) return;
    add_item(mk.item, mk.gives);
    bump_ach(mk.ach);
  }

  __device__ __forceinline__ void player_update(int action) {
    Obj p = obj_rd(1);
    int px = p.x, py = p.y;
    int tx = px + p.fx, ty = py + p.fy;
    int material, obj;
    cell(tx, ty, material, obj);
    int kind, arg;
    if (!rules_staged) {   // the compiled-in default rules: packed literals (types.hpp), no memory access
      int sh = 3 * (action < kPackedActions ? action : kPackedActions - 1);   // (actions are validated: < n_actions)
      kind = (int)((kDefaultActionKinds >> sh) & 7u);
      arg = (int)((kDefaultActionArgs >> sh) & 7u);
    } else {
      kind = R.action_kind[action];
      arg = R.action_arg[action];
    }
    int energy_max = R.item_max[R.item_energy];
    if (rec->sleeping) {  // objects.py:103-108
      if (rec->inv[R.item_energy] < energy_max) {
        kind = A_SLEEP;
      } else {
        st(&rec->sleeping, 0);
        bump_ach(R.ach_wake_up);
      }
    }
    if (kind == A_MOVE) {  // objects.py:174-179
      int fx = (arg == 0) ? -1 : (arg == 1) ? 1 : 0;
      int fy = (arg == 2) ? -1 : (arg == 3) ? 1 : 0;
      if constexpr (kFar) {
        obj_mod(1, [&](Obj& o) {
          o.fx = (int8_t)fx;
          o.fy = (int8_t)fy;
        });
      } else {
        st(&objs[1].fx, fx);
        st(&objs[1].fy, fy);
      }
      w.wsync();
      try_move(1, px, py, fx, fy, R.player_walkable_mask);
      Obj q = obj_rd(1);
      if (mat_at_uniform(q.x, q.y) == R.mat_lava) {
        st(&rec->inv[R.item_health], 0);
        w.wsync();
      }
    } else if (kind == A_DO) {
      if (obj)
        do_object(obj);
      else
        do_material(tx, ty, material);
    } else if (kind == A_SLEEP) {
      if (rec->inv[R.item_energy] < energy_max) st(&rec->sleeping, 1);
    } else if (kind == A_PLACE) {
      place(arg, tx, ty, material, obj);
    } else if (kind == A_MAKE) {
      make(arg, px, py);
    }
    w.wsync();
    // _update_life_stats objects.py:133-151 (counters in units of 0.5)
    int sleeping = rec->sleeping;
    int food = rec->inv[R.item_food], drink = rec->inv[R.item_drink], energy = rec->inv[R.item_energy];
    int health = rec->inv[R.item_health];
    int hunger = rec->hunger2 + (sleeping ? 1 : 2);
    if (hunger > 50) {
      hunger = 0;
      food -= 1;
    }
    int thirst = rec->thirst2 + (sleeping ? 1 : 2);
    if (thirst > 40) {
      thirst = 0;
      drink -= 1;
    }
    int fatigue = rec->fatigue2;
    if (sleeping)
      fatigue = imin(fatigue - 2, 0);
    else
      fatigue += 2;
    if (fatigue < -20) {
      fatigue = 0;
      energy += 1;
    }
    if (fatigue > 60) {
      fatigue = 0;
      energy -= 1;
    }
    // _degen_or_regen_health objects.py:153-167
    int recover = rec->recover2;
    bool ok = food > 0 && drink > 0 && (energy > 0 || sleeping);
    if (ok)
      recover += sleeping ? 4 : 2;
    else
      recover -= sleeping ? 1 : 2;
    if (recover > 50) {
      recover = 0;
      health = imax(0, health + 1);
    }
    if (recover < -30) {
      recover = 0;
      health = imax(0, health - 1);
    }
    if (w.leader()) {
      rec->hunger2 = hunger;
      rec->thirst2 = thirst;
      rec->fatigue2 = fatigue;
      rec->recover2 = recover;
      rec->inv[R.item_food] = food;
      rec->inv[R.item_drink] = drink;
      rec->inv[R.item_energy] = energy;
      rec->inv[R.item_health] = health;
    }
    w.wsync();
    // clamp every item to [0, max] objects.py:126-128 (one lane per item)
    w.lanes(0, R.n_items, [&](int i, int) {
      int v = rec->inv[i];
      int most = rules_staged ? R.item_max[i] : (int)((kDefaultItemMax >> (4 * (i & 15))) & 15u);   // literal: no load (types.hpp)
      rec->inv[i] = imax(0, imin(v, most));
    });
    w.wsync();
    // _wake_up_when_hurt objects.py:169-172
    health = rec->inv[R.item_health];
    if (health < rec->player_last_health) st(&rec->sleeping, 0);
    st(&rec->player_last_health, health);
    w.wsync();
  }

  // ------------------------------------------------------------------ creatures (objects.py:264-411)
  __device__ __forceinline__ void update_cow(int slot, const Obj& o) {  // objects.py:274-279
    bool alive = o.health > 0;
    if (!alive) obj_remove(slot, o);
    if (uniform_below(mt_prob53(0.5))) {
      int dx, dy;
      random_dir(dx, dy);
      try_move(slot, o.x, o.y, dx, dy, R.walkable_mask, alive, 4);
    }
  }

  __device__ __forceinline__ void update_zombie(int slot, const Obj& o) {  // objects.py:294-312
    bool alive = o.health > 0;
    if (!alive) obj_remove(slot, o);
    int x = o.x, y = o.y;
    int dist = iabs(upx - x) + iabs(upy - y);
    int dx, dy;
    if (dist <= 8 && uniform_below(mt_prob53(0.9))) {
      bool long_axis = uniform_below(mt_prob53(0.8));
      toward(x, y, upx, upy, long_axis, dx, dy);
    } else {
      random_dir(dx, dy);
    }
    bool moved = try_move(slot, x, y, dx, dy, R.walkable_mask, alive, 2) && alive;
    // the position field after World.move (unchanged if blocked, or removed: a removed zombie still strikes from where it stood)
    int nx = moved ? x + dx : x, ny = moved ? y + dy : y;
    dist = iabs(upx - nx) + iabs(upy - ny);
    if (dist <= 1) {
      if (o.aux) {
        set_aux(slot, o.aux - 1);
      } else {
        damage(1, W::uni((int)rec->sleeping) ? 7 : 2);
        set_aux(slot, 5);
      }
      w.wsync();
    }
  }
  __device__ __forceinline__ void update_skeleton(int slot, const Obj& o) {  // objects.py:327-351
    bool alive = o.health > 0;
    if (!alive) obj_remove(slot, o);
    int reload = imax(0, o.aux - 1);
    set_aux(slot, reload);
    w.wsync();
    struct { int x, y; } p = {upx, upy};   // the player does not move while the objects update
    int x = o.x, y = o.y;
    int dist = iabs((int)p.x - x) + iabs((int)p.y - y);
    int dx, dy;
    if (dist <= 3) {
      bool long_axis = uniform_below(mt_prob53(0.6));
      toward(x, y, p.x, p.y, long_axis, dx, dy);
      if (try_move(slot, x, y, -dx, -dy, R.walkable_mask, alive, 3)) return;
    }
    if (dist <= 5 && uniform_below(mt_prob53(0.5))) {
      toward(x, y, p.x, p.y, true, dx, dy);  // _shoot objects.py:343-351
      if (reload > 0) return;
      if (dx == 0 && dy == 0) return;
      if (is_free(x + dx, y + dy, R.arrow_walkable_mask)) {
        obj_add(T_ARROW, x + dx, y + dy, 0, dx, dy, 0);
        set_aux(slot, 4);
        w.wsync();
      }
    } else if (dist <= 8 && uniform_below(mt_prob53(0.3))) {
      bool long_axis = uniform_below(mt_prob53(0.6));
      toward(x, y, p.x, p.y, long_axis, dx, dy);
      try_move(slot, x, y, dx, dy, R.walkable_mask, alive, 3);
    } else if (uniform_below(mt_prob53(0.2))) {
      random_dir(dx, dy);
      try_move(slot, x, y, dx, dy, R.walkable_mask, alive, 3);
    }
  }

  __device__ __forceinline__ void update_arrow(int slot, const Obj& o) {  // objects.py:373-384
    int tx = o.x + o.fx, ty = o.y + o.fy;
    int m, t;
    cell(tx, ty, m, t);
    if (t) {
      damage(t, 2);
      obj_remove(slot, o);
    } else if (!((R.arrow_walkable_mask >> m) & 1u)) {
      obj_remove(slot, o);
      if ((R.arrow_breaks_mask >> m) & 1u) set_mat(tx, ty, R.mat_path);
      w.wsync();
    } else {
      try_move(slot, o.x, o.y, o.fx, o.fy, R.arrow_walkable_mask);
    }
  }

  __device__ __forceinline__ void update_plant(int slot, const Obj& o) {  // objects.py:405-411
    set_aux(slot, o.aux + 1);
    bool eaten = false;
    for (int d = 0; d < 4; d++) {
      int dx = (d == 0) ? -1 : (d == 1) ? 1 : 0;
      int dy = (d == 2) ? -1 : (d == 3) ? 1 : 0;
      int m, t;
      cell(o.x + dx, o.y + dy, m, t);
      if (t) {
        int tt;
        if constexpr (kFar)
          tt = obj_rd(t).type;
        else
          tt = W::uni((int)objs[t].type);
        if (tt == T_ZOMBIE || tt == T_SKELETON || tt == T_COW) eaten = true;
      }
    }
    int h = o.health;
    if (eaten) {
      h = imax(0, h - 1);
      set_health(slot, h);
    }
    w.wsync();
    if (h <= 0) obj_remove(slot, o);
  }

  int upx = 0, upy = 0;   // the player's position while the objects update (it cannot change there)
  int lane_objs_base = -4096;      // first slot of the 64 records update_all holds in lane registers (far away: none)
  uint64_t lane_objs_stale = 0;    // which of them have been written behind the registers' back
  __device__ __forceinline__ void update_object(int slot, const Obj& o) {   // o: the object's record as of now
    int t = o.type;
    if (t == T_COW)
      update_cow(slot, o);
    else if (t == T_ZOMBIE)
      update_zombie(slot, o);
    else if (t == T_SKELETON)
      update_skeleton(slot, o);
    else if (t == T_ARROW)
      update_arrow(slot, o);
    else if (t == T_PLANT)
      update_plant(slot, o);
  }

  __device__ __forceinline__ void update_all(int action, uint64_t* prof = nullptr) {
    update_all(action, prof, [] {});
  }
  // after_player(): called once Player.update has run -- the player's position, his sleep and the map's materials are what
  // this step's frame will show (the objects that follow move sprites; an arrow that breaks something sets mat_dirty)
  template <class F>
  __device__ __forceinline__ void update_all(int action, uint64_t* prof, F after_player) {
    int n = nobj;  // list snapshot (engine.py:41-44): objects appended this step are not visited
    if (prof && w.leader()) prof[9] = w.clock();
    player_update(action);
    mat_dirty = 0;
    after_player();
    if (prof && w.leader()) prof[10] = w.clock();
    Obj p = obj_rd(1);
    int ppx = p.x, ppy = p.y, lim = cfg.update_dist;
    upx = ppx;
    upy = ppy;
    if constexpr (kFar) {
      update_near(n, ppx, ppy, lim);
      return;
    }
    // 64 records at a time go into lane registers (three dwords each; the fourth is padding): the distance filter is a
    // ballot over them, and the serial loop takes an object's record out of its lane (v_readlane) instead of paying an LDS
    // round trip for it.  The only writes to ANOTHER object's record inside the loop are an arrow's hit (damage()), which
    // marks the target stale: a stale object is read from LDS again.  (Lane slot 2 is the RNG look-ahead.)
    for (int base = 0; base < n; base += 64) {
      w.lane_set(0, base, n, [&](int i, int) -> uint32_t { return ((const uint32_t*)&objs[i])[0]; });
      w.lane_set(1, base, n, [&](int i, int) -> uint32_t { return ((const uint32_t*)&objs[i])[1]; });
      w.lane_set(3, base, n, [&](int i, int) -> uint32_t { return ((const uint32_t*)&objs[i])[2]; });
      lane_objs_base = base;
      lane_objs_stale = 0;
      uint64_t m = w.ballot(base, n, [&](int i) {
        if (i < 2) return false;
        uint32_t w0 = w.lane_get(0, i - base), w1 = w.lane_get(1, i - base);
        int ox = (int)(w1 & 0xFFFFu), oy = (int)(w1 >> 16);
        return (w0 & 0xFFu) != T_NONE && (iabs(ox - ppx) + iabs(oy - ppy)) < lim;
      });
      while (m) {
        int b = __builtin_ctzll(m);
        m &= m - 1;
        Obj o;
        uint32_t words[4] = {0u, 0u, 0u, 0u};
        if ((W::uni64(lane_objs_stale) >> b) & 1ull) {   // (read again from LDS: the same three words in every lane)
          const uint32_t* rw = (const uint32_t*)&objs[base + b];
          words[0] = rw[0];
          words[1] = rw[1];
          words[2] = rw[2];
        } else {
          words[0] = w.lane_read(0, b);
          words[1] = w.lane_read(1, b);
          words[2] = w.lane_read(3, b);
        }
        // ... and said to be the same in every lane wherever they came from: the update below then branches on scalar compares
        words[0] = (uint32_t)W::uni((int)words[0]);
        words[1] = (uint32_t)W::uni((int)words[1]);
        words[2] = (uint32_t)W::uni((int)words[2]);
        __builtin_memcpy(&o, words, sizeof(Obj));
        update_object(base + b, o);
      }
    }
    lane_objs_base = -4096;
  }

  // ------------------------------------------------------------------ FarSlot: the scan (every wave of the workgroup)
  // One pass over the env's slot table in global memory at the head of a step: which objects may be updated in it (near_mask),
  // their records into the cache, the live records counted.  Wave v takes the 64-record batches v, v + NW, v + 2 NW, ...;
  // kFarRound batches' loads are in flight per wave and round (three lane registers per batch), and the first round is issued
  // BLIND, with the rest of the stage-in's loads, before the table's length is known (1024 slots with four waves: a 256x256
  // world holds ~750 objects).
  //   far_clear + far_issue + far_publish ... barrier ... far_rounds ... barrier
  __device__ __forceinline__ void far_clear() {
    w.block_for(64, [&](int i) { ((uint32_t*)ndm)[i] = 0u; });
    w.block_for(2, [&](int i) { nctr[i] = i == 0 ? 1u : 0u; });   // ([0]: entry 0 is the player's)
  }
  __device__ __forceinline__ void far_issue(int b0) {
    constexpr int NW = W::num_waves();
    const int cap = cfg.max_objects;
    auto rec16 = [&](int i, int) -> vec16 { return *(const vec16*)&objs[i]; };
    w.template lane_set4<0>(64 * b0, cap, rec16);
    if constexpr (kFarRound > 1) w.template lane_set4<4>(64 * (b0 + NW), cap, rec16);
    if constexpr (kFarRound > 2) w.template lane_set4<8>(64 * (b0 + 2 * NW), cap, rec16);
    if constexpr (kFarRound > 3) w.template lane_set4<12>(64 * (b0 + 3 * NW), cap, rec16);
    if constexpr (kFarRound > 4) w.template lane_set4<16>(64 * (b0 + 4 * NW), cap, rec16);
  }
  // the player's position as staged, for every wave's distance filter (slot 1 = batch 0, lane 1, of the first wave's first round)
  __device__ __forceinline__ void far_publish() {
    if (w.wave0()) {
      uint32_t pp = w.lane_read(1, 1);
      if (w.leader()) nctr[2] = pp;
    }
  }
  template <int R0, int R1, int R2>
  __device__ __forceinline__ void far_eval(int b, int n, int ppx, int ppy) {
    const int base = 64 * b;
    if (base >= n) return;
    const int lim = cfg.update_dist + 1;   // (the player moves one cell at most before the objects update: objects.py:174-179)
    const uint64_t live = w.ballot(base, n, [&](int i) { return i >= 1 && (w.lane_get(R0, i - base) & 0xFFu) != T_NONE; });
    const uint64_t m = w.ballot(base, n, [&](int i) {
      if (i < 2) return false;
      uint32_t w0 = w.lane_get(R0, i - base), w1 = w.lane_get(R1, i - base);
      int ox = (int)(w1 & 0xFFFFu), oy = (int)(w1 >> 16);
      return (w0 & 0xFFu) != T_NONE && (iabs(ox - ppx) + iabs(oy - ppy)) < lim;
    });
    uint32_t at = 0;
    if (w.lane() == 0) {
      near_mask[b] = m;
      if (live) w.lds_fetch_add(&nctr[1], (uint32_t)__builtin_popcountll(live));
      if (m) at = w.lds_fetch_add(&nctr[0], (uint32_t)__builtin_popcountll(m));
    }
    if (b == 0) {   // the player's record: entry 0, whatever else the cache holds
      w.lanes(0, n, [&](int i, int lane) {
        if (i != 1) return;
        uint32_t* c = (uint32_t*)&ncache[0];
        c[0] = w.lane_get(R0, lane);
        c[1] = w.lane_get(R1, lane);
        c[2] = w.lane_get(R2, lane);
        c[3] = 1u;
        ndm[1] = 1;
      });
    }
    if (!m) return;
    at = (uint32_t)W::uni((int)at);
    w.lanes(base, n, [&](int i, int lane) {
      if (!((m >> lane) & 1ull)) return;
      int k = (int)at + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (k >= kFarCache || (i & 255) == 1) return;   // (no entry: the table serves it; the index byte of slot 1 is the player's)
      uint32_t* c = (uint32_t*)&ncache[k];
      c[0] = w.lane_get(R0, lane);
      c[1] = w.lane_get(R1, lane);
      c[2] = w.lane_get(R2, lane);
      c[3] = (uint32_t)i;
      ndm[i & 255] = (uint8_t)(k + 1);   // (two near slots with one low byte: one of them keeps the index, the other reads the table)
    });
  }
  // behind the barrier that published the record and far_publish's word: evaluate the blind round, run the others
  __device__ __forceinline__ void far_rounds(int n) {
    constexpr int NW = W::num_waves();
    const uint32_t pp = nctr[2];
    const int ppx = (int)(pp & 0xFFFFu), ppy = (int)(pp >> 16);
    int b0 = w.wave_index();
    for (;;) {
      far_eval<0, 1, 2>(b0, n, ppx, ppy);
      if constexpr (kFarRound > 1) far_eval<4, 5, 6>(b0 + NW, n, ppx, ppy);
      if constexpr (kFarRound > 2) far_eval<8, 9, 10>(b0 + 2 * NW, n, ppx, ppy);
      if constexpr (kFarRound > 3) far_eval<12, 13, 14>(b0 + 3 * NW, n, ppx, ppy);
      if constexpr (kFarRound > 4) far_eval<16, 17, 18>(b0 + 4 * NW, n, ppx, ppy);
      b0 += kFarRound * NW;
      if (64 * b0 >= n) break;
      far_issue(b0);
    }
  }
  // behind the barrier that follows: what the scan counted
  __device__ __forceinline__ void far_done() {
    far_live = (int)nctr[1];
    cur_slot = -1;
    cur_idx = -1;
  }

  // FarSlot: the object loop over the slots the scan marked (near_mask), their records out of the cache.  Same order, same
  // filter (the exact distance to the player where he stands now), same stale rule as the loop above.
  __device__ __forceinline__ void update_near(int n, int ppx, int ppy, int lim) {
    for (int base = 0; base < n; base += 64) {
      const uint64_t cm = W::uni64(near_mask[base >> 6]);
      if (!cm) continue;
      // (register 4: the lane's cache entry + 1, 0 = none: its record comes from the table)
      w.lane_set(4, base, n, [&](int i, int lane) -> uint32_t { return ((cm >> lane) & 1ull) ? (uint32_t)(far_find_lane(i) + 1) : 0u; });
      auto word = [&](int i, int lane, int k) -> uint32_t {
        if (!((cm >> lane) & 1ull)) return 0u;
        int e_ = (int)w.lane_get(4, lane) - 1;
        if (e_ >= 0) return ((const uint32_t*)&ncache[e_])[k];
        return W::load_fresh((const uint32_t*)&objs[i] + k);
      };
      w.lane_set(0, base, n, [&](int i, int lane) -> uint32_t { return word(i, lane, 0); });
      w.lane_set(1, base, n, [&](int i, int lane) -> uint32_t { return word(i, lane, 1); });
      w.lane_set(3, base, n, [&](int i, int lane) -> uint32_t { return word(i, lane, 2); });
      lane_objs_base = base;
      lane_objs_stale = 0;
      uint64_t m = w.ballot(base, n, [&](int i) {
        if (i < 2) return false;
        uint32_t w0 = w.lane_get(0, i - base), w1 = w.lane_get(1, i - base);
        int ox = (int)(w1 & 0xFFFFu), oy = (int)(w1 >> 16);
        return (w0 & 0xFFu) != T_NONE && (iabs(ox - ppx) + iabs(oy - ppy)) < lim;
      });
      m &= cm;
      while (m) {
        int b = __builtin_ctzll(m);
        m &= m - 1;
        cur_slot = -1;
        cur_idx = W::uni((int)w.lane_read(4, b)) - 1;
        cur_slot = base + b;
        Obj o;
        if ((W::uni64(lane_objs_stale) >> b) & 1ull) {
          o = obj_rd(base + b);
        } else {
          uint32_t words[4] = {0u, 0u, 0u, 0u};
          words[0] = (uint32_t)W::uni((int)w.lane_read(0, b));
          words[1] = (uint32_t)W::uni((int)w.lane_read(1, b));
          words[2] = (uint32_t)W::uni((int)w.lane_read(3, b));
          __builtin_memcpy(&o, words, sizeof(Obj));
        }
        update_object(base + b, o);
      }
      cur_slot = -1;
      cur_idx = -1;
    }
    lane_objs_base = -4096;
  }

  // ------------------------------------------------------------------ balance (env.py:141-179)
  // The census -- per chunk the number of grass / path cells (maintained by set_mat) and of zombies / skeletons / cows
  // (maintained by obj_add / obj_remove / obj_move) -- is read as it stands: each (chunk, class) pair is evaluated exactly
  // once and a spawn or despawn only changes its own entry, AFTER it has been read.
#ifdef CRAFTER_BALANCE_PROBE   // probe builds only (tools/r4_balance_probe.sh): where a balance pass spends its clocks
  uint64_t* bal_prof = nullptr;
  uint32_t bal_acc[6] = {0, 0, 0, 0, 0, 0};   // pair flags, speculation rounds, serial draws, despawn pass, cell search, apply loop
  uint32_t bal_n[2] = {0, 0};                 // hits, speculation rounds
#define BAL_T0 uint64_t bal_t0 = w.clock();
#define BAL_ADD(k) { uint64_t bal_t1 = w.clock(); bal_acc[k] += (uint32_t)(bal_t1 - bal_t0); bal_t0 = bal_t1; }
#else
#define BAL_T0
#define BAL_ADD(k)
#endif
  __device__ __forceinline__ void balance(double light) {
    // (the creature counts are current: World.add / remove / move keep them -- count_creature)
    // census_global: this step's updates are fire-and-forget atomic adds of the leader lane, the reads below relaxed
    // agent-scope loads of all lanes -- every one of those adds has to have reached the coherence point first (ADVICE r4:
    // nothing but same-address ordering in the memory pipeline stood between them).  One wait every tenth step.
    if (census_global) W::drain_stores();
    int zt = (int)(3.5 - 3 * light);  // int(target) of env.py:147, values are >= 0.5
    int ct = (int)(1.5 + light);      // env.py:155
    int nch = rec->nchunks_seen;  // chunk keys in dict insertion order; keys added during the
                                  // pass cannot appear (spawns stay inside the chunk being balanced)
    // One lane per (chunk in insertion order, class) pair, in the reference's visiting order
    // (env.py:143-155: Zombie, Skeleton, Cow per chunk).  A pair whose count sits inside
    // [int(target_min), int(target_max)] draws nothing and changes nothing, so only the pairs that
    // reach a uniform() are visited serially.  bit 0: spawn branch, bit 1: despawn branch.
    //
    // Round 4: the pass is split into WHAT IS DRAWN and WHAT IS DONE WITH IT.  Everything a pair draws depends on the
    // census alone -- uniform() against its probability, then randint(space) (the cell, env.py:166) or randint(n) (the
    // creature, env.py:176) -- and a pair's action changes no other pair's census entry, so the whole stream of a pass can
    // be consumed first: every pair that hits goes into the hit list (lane register 4, lane h = h-th hit: pair | drawn
    // index << 16 | spawn << 31) and the world is only touched when the list is applied, in order, 64 hits at a time
    // (apply_hits): all despawn victims of a round are found in ONE pass over the slot table, and the map cells a spawn
    // needs are no longer on the serial chain of the draws.
    int npair = nch * 3;
    int nh = 0;
    // (registers 8, 9: the census words of the NEXT 64 pairs -- creatures, cells -- fetched while this batch draws: with the
    // census in HBM (worlds that do not fit in LDS) a batch used to open with one full memory round trip, 23 of them per
    // pass on a 256x256 world.  Two registers and no arithmetic on the loaded words: anything computed from them here
    // would put the wait for the loads here too.)
    auto creatures_of = [&](int pidx, int) -> uint32_t {
      int j = pidx / 3, k = pidx - 3 * j;
      return (uint32_t)cen(chunk_order[j] * 5 + 2 + k);
    };
    auto cells_of = [&](int pidx, int) -> uint32_t {
      int j = pidx / 3, k = pidx - 3 * j;
      return (uint32_t)cen(chunk_order[j] * 5 + (k == 1 ? 1 : 0));
    };
    w.lane_set(8, 0, npair, creatures_of);
    w.lane_set(9, 0, npair, cells_of);
    BAL_T0
    for (int base = 0; base < npair; base += 64) {
      // (bits 8..: what the pair's randint would range over -- the material's cells for a spawn, the creatures for a despawn
      // -- so that a hit needs no second look at the census)
      // (register 7: apply_hits, which may run in the middle of a batch, uses 0, 1, 3, 5 and 6)
      w.lane_set(7, base, npair, [&](int pidx, int lane) -> uint32_t {
        int k = pidx % 3;
        int n = (int)w.lane_get(8, lane), space = (int)w.lane_get(9, lane);
        int tmin = (k == 0) ? (space < 50 ? 0 : zt) : (k == 1) ? (space < 6 ? 0 : 1) : (space < 30 ? 0 : 1);
        int tmax = (k == 0) ? zt : (k == 1) ? 2 : ct;
        uint32_t f = (uint32_t)((n < tmin ? 1 : 0) | (n > tmax ? 2 : 0));
        return f | ((uint32_t)((f & 1u) ? space : n) << 8);
      });
      if (base + 64 < npair) {
        w.lane_set(8, base + 64, npair, creatures_of);
        w.lane_set(9, base + 64, npair, cells_of);
      }
      uint64_t spawn = w.lane_ballot(7, 1), despawn = w.lane_ballot(7, 2);
      uint64_t act = spawn | despawn;
      BAL_ADD(0)
      // An active pair whose uniform() misses its probability consumes exactly two stream words and
      // changes nothing (most do: probabilities 0.01 .. 0.4).  So the pairs are resolved
      // speculatively: lane b assumes every active pair before it missed, reads its two words
      // straight from the generator state and tests its own probability; the first pair that
      // hits (or whose words lie past the state block) ends the speculation, the stream skips the
      // misses before it, that pair draws serially, and the rest is speculated again.
      while (act) {
        int pos = mt_pos;
        uint64_t stop = w.ballot(base, npair, [&](int pidx) {
          int b = pidx - base;
          if (!((act >> b) & 1ull)) return false;
          int at = pos + 2 * __builtin_popcountll(act & ((1ull << b) - 1ull));
          if (at + 1 >= MT_N) return true;            // needs a twist first: serial path
          int k = pidx % 3;
          double prob = ((spawn >> b) & 1ull) ? ((k == 0) ? 0.3 : (k == 1) ? 0.1 : 0.01) : ((k == 0) ? 0.4 : 0.1);
          return mt_double(mt_temper(mt[at]), mt_temper(mt[at + 1])) < prob;
        });
        stop &= act;
        BAL_ADD(1)
#ifdef CRAFTER_BALANCE_PROBE
        bal_n[1]++;
#endif
        if (!stop) {
          mt_pos = pos + 2 * __builtin_popcountll(act);
          break;
        }
        int b = __builtin_ctzll(stop);
        mt_pos = pos + 2 * __builtin_popcountll(act & ((1ull << b) - 1ull));
        act &= ~((2ull << b) - 1ull);
        int pidx = base + b;
        int k = pidx % 3;
        bool want_spawn = (spawn >> b) & 1ull;
        double prob = want_spawn ? ((k == 0) ? 0.3 : (k == 1) ? 0.1 : 0.01) : ((k == 0) ? 0.4 : 0.1);
        if (uniform() < prob) {   // env.py:165 / 174 (a pair stopped for the twist alone may still miss)
          // env.py:166-170: the i-th cell of the material, or env.py:176: the k-th creature (no draw when there is one)
          uint32_t drawn = randint(w.lane_read(7, b) >> 8);
          w.lane_put(4, nh, (uint32_t)pidx | (drawn << 16) | (want_spawn ? 0x80000000u : 0u));
#ifdef CRAFTER_BALANCE_PROBE
          bal_n[0]++;
#endif
          BAL_ADD(2)
          if (++nh == 64) {
            apply_hits(nh);
            nh = 0;
#ifdef CRAFTER_BALANCE_PROBE
            bal_t0 = w.clock();
#endif
          }
        }
        BAL_ADD(2)
      }
    }
    if (nh) apply_hits(nh);
#ifdef CRAFTER_BALANCE_PROBE
    if (bal_prof && w.leader()) {
      bal_prof[14] = (uint64_t)bal_acc[0] | ((uint64_t)bal_acc[1] << 32);
      bal_prof[15] = (uint64_t)bal_acc[2] | ((uint64_t)bal_acc[3] << 32);
      bal_prof[6] = (uint64_t)bal_acc[4] | ((uint64_t)bal_acc[5] << 32);
      bal_prof[12] = (uint64_t)bal_n[0] | ((uint64_t)bal_n[1] << 32);
    }
#endif
  }

  // one lane register's worth of a chunk's cells: is the i-th cell of `material` among them?
  template <int SLOT>
  __device__ __forceinline__ void scan_cells(int base, int ncell, int material, int& i, int& found) {
    if (found >= 0 || base >= ncell) return;
    uint64_t m = w.lane_match(SLOT, base, ncell, (uint32_t)material);
    int cnt = __builtin_popcountll(m);
    if (i < cnt)
      found = base + w.kth_set(m, i);
    else
      i -= cnt;
  }

  // one batch of 64 slot keys (lane register R) against the open despawn hits (apply_hits: registers 5, 6)
  template <int R>
  __device__ __forceinline__ void despawn_match(int base, int total, uint64_t& open) {
    if (base >= total || !open) return;
    uint64_t todo = open;
    while (todo) {
      int h = __builtin_ctzll(todo);
      todo &= todo - 1;
      uint32_t st5 = w.lane_read(5, h);
      uint64_t m = w.lane_match(R, base, total, st5 & 0xFFFFu);
      int cnt = __builtin_popcountll(m), kk = (int)(st5 >> 16);
      if (kk < cnt) {
        w.lane_put(6, h, (uint32_t)(base + w.kth_set(m, kk)));
        open &= ~(1ull << h);
      } else if (cnt) {
        w.lane_put(5, h, (st5 & 0xFFFFu) | ((uint32_t)(kk - cnt) << 16));
      }
    }
  }

  // The hits of a balance pass (lane register 4, lanes 0 .. nh - 1), applied to the world in their order.
  __device__ __forceinline__ void apply_hits(int nh) {
    static_assert(CHUNK * CHUNK <= 255, "a drawn index fits the hit record's 8 bits... and the chunk's cells three lane registers");
    uint64_t all = nh >= 64 ? ~0ull : ((1ull << nh) - 1ull);
    uint64_t dmask = ~w.lane_ballot(4, 0x80000000u) & all;   // the despawn hits
    BAL_T0
    // Despawn victims, all at once: lane h of register 5 = the pair's key (chunk * 3 + class) | creatures still to skip << 16,
    // register 6 = the victim's slot (0: not found yet).  One pass over the slot table: every batch of 64 records is keyed
    // once and matched against each open hit by ballot -- the k-th creature of the class in the chunk, in slot order
    // (the canonical set order).
    if (dmask) {
      w.lane_set(5, 0, nh, [&](int h, int) -> uint32_t {
        uint32_t r = w.lane_get(4, h);
        int pidx = (int)(r & 0xFFFFu);
        int j = pidx / 3, k = pidx - 3 * j;
        return (uint32_t)(chunk_order[j] * 3 + k) | (((r >> 16) & 0xFFu) << 16);
      });
      w.lane_set(6, 0, nh, [&](int, int) -> uint32_t { return 0u; });
      uint64_t open = dmask;
      int total = nobj;
      if constexpr (kFar) {
        // the table is in global memory: the keys of TEN batches per memory round trip (registers 10 .. 19), matched batch by batch
#ifndef CRAFTER_FAR_KEYS
#define CRAFTER_FAR_KEYS 4
#endif
        constexpr int K = CRAFTER_FAR_KEYS;
        for (int b0 = 0; b0 < total && open; b0 += 64 * K) {
          w.template lane_gather_map<10, K>(
              b0, total, 0xFFFFu, [&](int i) -> uint64_t { return *(const uint64_t*)&objs[i]; },
              [&](uint64_t v, int i) -> uint32_t {
                int col = creature_col((int)(v & 0xFFu));
                if (i < 2 || col < 0) return 0xFFFFu;
                return (uint32_t)(chunk_of((int)((v >> 32) & 0xFFFFu), (int)(v >> 48)) * 3 + col - 2);
              });
          despawn_match<10>(b0, total, open);
          if constexpr (K > 1) despawn_match<11>(b0 + 64, total, open);
          if constexpr (K > 2) despawn_match<12>(b0 + 128, total, open);
          if constexpr (K > 3) despawn_match<13>(b0 + 192, total, open);
          if constexpr (K > 4) despawn_match<14>(b0 + 256, total, open);
          if constexpr (K > 5) despawn_match<15>(b0 + 320, total, open);
          if constexpr (K > 6) despawn_match<16>(b0 + 384, total, open);
          if constexpr (K > 7) despawn_match<17>(b0 + 448, total, open);
          if constexpr (K > 8) despawn_match<18>(b0 + 512, total, open);
          if constexpr (K > 9) despawn_match<19>(b0 + 576, total, open);
        }
      } else
      for (int base = 0; base < total && open; base += 64) {
        // this batch's keys (0xFFFF: not a creature)
        w.lane_set(0, base, total, [&](int i, int) -> uint32_t {
          if (i < 2) return 0xFFFFu;
          Obj o = objs[i];
          int col = creature_col(o.type);
          return col < 0 ? 0xFFFFu : (uint32_t)(chunk_of(o.x, o.y) * 3 + col - 2);
        });
        uint64_t todo = open;
        while (todo) {
          int h = __builtin_ctzll(todo);
          todo &= todo - 1;
          uint32_t st5 = w.lane_read(5, h);
          uint64_t m = w.lane_match(0, base, total, st5 & 0xFFFFu);
          int cnt = __builtin_popcountll(m), kk = (int)(st5 >> 16);
          if (kk < cnt) {
            w.lane_put(6, h, (uint32_t)(base + w.kth_set(m, kk)));
            open &= ~(1ull << h);
          } else if (cnt) {
            w.lane_put(5, h, (st5 & 0xFFFFu) | ((uint32_t)(kk - cnt) << 16));
          }
        }
      }
    }
    if constexpr (kFar) {
      // the victims' positions, all at once (register 10; the keys are done with): a record read one by one from the table in
      // global memory is a memory round trip per despawn on the serial chain below
      if (dmask) {
        W::drain_stores();   // (this step's moves have arrived where the loads below read)
        w.lane_set(10, 0, nh, [&](int h, int) -> uint32_t {
          int s = (int)w.lane_get(6, h);
          return s ? W::load_fresh((const uint32_t*)&objs[s] + 1) : 0u;
        });
      }
    }
    BAL_ADD(3)
    Obj p = obj_rd(1);
    // Spawn cells.  Where the maps are plain arrays (every layout but LaneSlots) and a chunk's rows are whole dwords, ALL
    // spawn hits of the round look for their cell at once, one lane per hit: two rows of the chunk at a time as dwords
    // (six loads in flight per lane), the bytes equal to the material counted with integer arithmetic, the drawn index
    // located -- six memory round trips per ROUND of up to 64 hits, where the wave used to gather and scan one chunk per HIT (a 256x256
    // world balancing at night: ~100 hits, maps in HBM: 200-380 k clocks per balance step, r4e).  The cells' slot-map
    // entries are fetched the same way; an entry is only read again, serially, if an earlier hit of this very round
    // touched that cell.  Registers: 0 = the cell found (x | y << 16), 1 = its slot-map entry, 3 = the cell a hit touched.
    bool batched = spawn_batched && !kLane && (cfg.H & 3) == 0 && (CHUNK & 3) == 0;
    if (batched) {
      const uint8_t* mp = (const uint8_t*)mat;
      w.lane_set(0, 0, nh, [&](int h, int) -> uint32_t {
        uint32_t r = w.lane_get(4, h);
        if (!(r & 0x80000000u)) return 0xFFFFFFFFu;
        int pidx = (int)(r & 0xFFFFu), want = (int)((r >> 16) & 0xFFu);
        int j = pidx / 3, k = pidx - 3 * j;
        int c = chunk_order[j];
        uint32_t pat = (uint32_t)((k == 1) ? R.mat_path : R.mat_grass) * 0x01010101u;
        int cx = c / cfg.nchunk_y, cy = c - cx * cfg.nchunk_y;
        int xmin = cx * CHUNK, ymin = cy * CHUNK;
        int nx = imin(xmin + CHUNK, cfg.W) - xmin, nd = (imin(ymin + CHUNK, cfg.H) - ymin) >> 2;   // rows, dwords per row
        uint32_t found = 0xFFFFFFFFu;
        int acc = 0;
        constexpr int kRows = 2;   // rows per pass (six dwords in flight per lane; more cost the step kernel its registers)
#pragma clang loop unroll(disable)
        for (int x0 = 0; x0 < nx && found == 0xFFFFFFFFu; x0 += kRows) {
          uint32_t v[kRows][CHUNK / 4];
#pragma unroll
          for (int a = 0; a < kRows; a++)
#pragma unroll
            for (int d = 0; d < CHUNK / 4; d++)   // clamped, unconditional: all of them in flight together
              v[a][d] = *(const uint32_t*)(mp + cidx(xmin + imin(x0 + a, nx - 1), ymin) + 4 * imin(d, nd - 1));
#pragma unroll
          for (int a = 0; a < kRows; a++)
#pragma unroll
            for (int d = 0; d < CHUNK / 4; d++) {
              if (x0 + a >= nx || d >= nd) continue;
              uint32_t m = v[a][d] ^ pat;
              uint32_t z = ~(((m & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m | 0x7F7F7F7Fu);   // 0x80 in every byte that equals the material (exact)
              int c4 = __builtin_popcount(z);
              if (found == 0xFFFFFFFFu && want < acc + c4) {
                int skip = want - acc;
#pragma unroll
                for (int q = 0; q < 4; q++)
                  if ((z >> (8 * q + 7)) & 1u) {
                    if (skip == 0 && found == 0xFFFFFFFFu) found = (uint32_t)(xmin + x0 + a) | ((uint32_t)(ymin + 4 * d + q) << 16);
                    skip--;
                  }
              }
              acc += c4;
            }
        }
        return found;
      });
      w.lane_set(1, 0, nh, [&](int h, int) -> uint32_t {
        uint32_t f = w.lane_get(0, h);
        if (f == 0xFFFFFFFFu) return 0u;
        return (uint32_t)(int)objmap[cidx((int)(f & 0xFFFFu), (int)(f >> 16))];
      });
      w.lane_set(3, 0, nh, [&](int, int) -> uint32_t { return 0xFFFFFFFFu; });
    }
    BAL_ADD(4)
    for (int h = 0; h < nh; h++) {
      uint32_t r = w.lane_read(4, h);
      int pidx = (int)(r & 0xFFFFu), drawn = (int)((r >> 16) & 0xFFu);
      int j = pidx / 3, k = pidx - 3 * j;
      int c = chunk_order[j];
      uint64_t before = (1ull << h) - 1ull;
      if (r & 0x80000000u) {   // env.py:165-172
        int span_dist = (k == 0) ? 6 : (k == 1) ? 7 : 5;
        int health = (k == 0) ? 5 : 3;
        int type = (k == 0) ? T_ZOMBIE : (k == 1) ? T_SKELETON : T_COW;
        int x, y;
        bool empty;
        if (batched) {
          uint32_t f = w.lane_read(0, h);
          if (f == 0xFFFFFFFFu) continue;  // unreachable: space counts exactly these cells
          x = (int)(f & 0xFFFFu);
          y = (int)(f >> 16);
          // (an earlier hit of this round spawned on, or despawned from, this very cell: the prefetched entry is stale)
          bool touched = (w.lane_match(3, 0, nh, f) & before) != 0;
          empty = touched ? slot_at(x, y) == 0 : w.lane_read(1, h) == 0;
        } else {
          int material = (k == 1) ? R.mat_path : R.mat_grass;
          int cx = c / cfg.nchunk_y, cy = c - cx * cfg.nchunk_y;
          int xmin = cx * CHUNK, ymin = cy * CHUNK;
          int xmax = imin(xmin + CHUNK, cfg.W), ymax = imin(ymin + CHUNK, cfg.H);
          int ch = ymax - ymin, ncell = (xmax - xmin) * ch;
          uint32_t inv_ch = (65536u + (uint32_t)ch - 1u) / (uint32_t)ch;   // q / ch for q < 144 by multiplication
          int i = drawn;  // i-th material cell in x-major order (env.py:166-170)
          int found = -1;
          // the chunk's cells (<= 144) into three lane registers at once (LaneSlots: far chunks come from HBM)
          w.lane_gather3(ncell, [&](int q) -> uint32_t {
            int dx = (int)(((uint32_t)q * inv_ch) >> 16);
            int dy = q - dx * ch;
            return (uint32_t)mat_at(xmin + dx, ymin + dy);
          });
          // (the register is named by a literal in every copy of the loop body: indexed by a run-time value the wave's
          // register array would be put in scratch memory)
          scan_cells<0>(0, ncell, material, i, found);
          scan_cells<1>(64, ncell, material, i, found);
          scan_cells<3>(128, ncell, material, i, found);
          if (found < 0) continue;  // unreachable: space counts exactly these cells
          int dx = (int)(((uint32_t)found * inv_ch) >> 16);
          x = xmin + dx;
          y = ymin + found - dx * ch;
          empty = slot_at(x, y) == 0;
        }
        bool away = (iabs(x - (int)p.x) + iabs(y - (int)p.y)) >= span_dist;
        if (empty && away) {
          obj_add(type, x, y, health, 0, 0, 0);
          if (batched) w.lane_put(3, h, (uint32_t)x | ((uint32_t)y << 16));
        }
      } else {                 // env.py:174-179
        int despan_dist = (k == 0) ? 0 : (k == 1) ? 7 : 5;
        int slot = (int)w.lane_read(6, h);
        if (slot == 0) continue;   // unreachable: the census counts exactly these creatures
        Obj o;
        if constexpr (kFar) {   // (its class is the pair's, its position was fetched above)
          uint32_t w1 = (uint32_t)W::uni((int)w.lane_read(10, h));
          uint32_t words[4] = {(uint32_t)((k == 0) ? T_ZOMBIE : (k == 1) ? T_SKELETON : T_COW), w1, 0u, 0u};
          __builtin_memcpy(&o, words, sizeof(Obj));
        } else {
          o = objs[slot];
        }
        bool away = (iabs((int)o.x - (int)p.x) + iabs((int)o.y - (int)p.y)) >= despan_dist;
        if (away) {
          if constexpr (kFar)
            obj_remove(slot, o);
          else
            obj_remove(slot);
          if (batched) w.lane_put(3, h, (uint32_t)o.x | ((uint32_t)o.y << 16));
        }
      }
    }
    BAL_ADD(5)
  }

  // ------------------------------------------------------------------ slot compaction
  // The reference's slot list is append-only (engine.py:54-55) but only the ORDER of slots is
  // observable (update order, despawn choice), so freed slots are squeezed out, order kept.
  __device__ __forceinline__ void compact() {
    if constexpr (kFar) {
      // lazily: the table is in global memory, a pass over it is a memory round trip per 64 records -- and holes cost nothing
      // but their slots (every pass skips T_NONE records, the reference's list keeps its None entries for ever: engine.py:62).
      // Squeezed out when there are kFarHoles (128) of them -- the scan's blind round covers 1024 slots --, or as soon as there
      // is one in a table that is about to look three quarters full to the host, which would double it (BatchedEnv._grow_objects)
      int holes = nobj - 1 - far_live;
      if (!(holes >= kFarHoles || (holes > 0 && 4 * (nobj + 32) >= 3 * cfg.max_objects))) return;
    } else {
      if (!dirty_slots) return;
    }
    int n = nobj;
    int out = 0;
    for (int base = 0; base < n; base += 64) {
      uint64_t m = w.ballot(base, n, [&](int i) { return i == 0 || objs[i].type != T_NONE; });
      w.lanes(base, n, [&](int i, int lane) {
        if (!((m >> lane) & 1ull)) return;
        int ni = out + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (ni != i) {
          Obj o = objs[i];     // every lane reads before any lane writes (lock-step wave)
          objs[ni] = o;
          if constexpr (kFar) {
            far_put_objmap(o.x, o.y, ni);
          } else if constexpr (!kLane) {
            int ci = cidx(o.x, o.y);
            objmap[ci] = (SlotT)ni;
            if (g_objmap && (const void*)g_objmap != (const void*)objmap) g_objmap[ci] = (uint16_t)ni;
          }
        }
      });
      out += __builtin_popcountll(m);
      w.wsync();
    }
    nobj = out;
    dirty_slots = 0;
    occ_rebuild();
    if constexpr (kFar) {   // every slot behind the first hole has a new number: the cache is keyed by the old ones
      far_live = out - 1;
      w.lanes(0, 64, [&](int i, int) { ((uint32_t*)ndm)[i] = 0u; });
      w.wsync();
      if (w.leader()) ndm[1] = 1;   // (slots 0 and 1 never move: the player's entry stands)
      w.wsync();
    }
  }

  // ------------------------------------------------------------------ episode start (env.py:70-79)
  // Everything Env.reset does except World.reset's maps and the terrain: fresh Player
  // (objects.py:70-82), Env bookkeeping.  Called by every wave of the workgroup.
  __device__ __forceinline__ void begin_episode(int episode) {
    w.block_for(R.n_items, [&](int i) { rec->inv[i] = R.item_init[i]; });
    w.block_for(MAX_ACH, [&](int i) { rec->ach[i] = 0; });
    w.sync();
    if (w.leader()) {
      rec->episode = episode;
      rec->step = 0;
      rec->hunger2 = 0;
      rec->thirst2 = 0;
      rec->fatigue2 = 0;
      rec->recover2 = 0;
      rec->sleeping = 0;
      rec->unlocked = 0;
      rec->dhealth = 0;
      rec->new_unlocked = 0;
      rec->ep_dhealth = 0;
      rec->ep_unlock_steps = 0;
      rec->dead = 0;
      rec->done = 0;
      rec->needs_reset = 0;
      rec->status &= ~ST_STEP_OVERFLOW;   // stepping a finished env past `length` is legal in the reference; the
                                          // clamped daylight index it caused ends with that episode
      int h0 = rec->inv[R.item_health];
      rec->player_last_health = h0;   // objects.py:78
      rec->env_last_health = h0;      // env.py:77
    }
    w.sync();
  }

  // ------------------------------------------------------------------ reward / done (env.py:96-118)
  __device__ __forceinline__ void finish_step(float* reward_out, uint8_t* done_out, int reward_enabled) {
    int health = rec->inv[R.item_health];
    int dh = health - rec->env_last_health;
    uint64_t have = w.ballot(0, R.n_achievements, [&](int i) { return rec->ach[i] > 0; });
    uint32_t fresh = (uint32_t)have & ~rec->unlocked;
    double r = (double)dh / 10.0;
    if (fresh) r += 1.0;
    int dead = health <= 0;
    int over = cfg.length > 0 && rec->step >= cfg.length;
    int done = dead || over;
    if (w.leader()) {
      rec->env_last_health = health;
      rec->unlocked |= fresh;
      rec->dhealth = dh;
      rec->new_unlocked = fresh;
      rec->ep_dhealth += dh;
      rec->ep_unlock_steps += fresh ? 1 : 0;
      rec->dead = dead;
      rec->done = done;
      rec->needs_reset = (done && cfg.auto_reset) ? 1 : 0;
      *reward_out = reward_enabled ? (float)r : 0.0f;
      *done_out = (uint8_t)done;
    }
    w.wsync();
  }
};

}  // namespace crafter
