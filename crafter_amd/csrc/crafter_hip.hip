// libcrafter_hip.so: gfx950 kernels + the C ABI of include/crafter_hip.h.
//
// Build (see __graft_entry__.build):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
// -ffp-contract=off is part of the numerical contract: the render filters, the terrain noise and
// the balance thresholds are float expressions the reference evaluates operation by operation.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#define CRAFTER_HIP_INTERNAL
#include "../../include/crafter_hip.h"
#include "crafter_rollout.hpp"
#include "dispatch_order.hpp"
#include "env_kernels.hpp"
#include "wave_gfx950.hpp"

using namespace crafter;

// crafter_config / crafter_rules / crafter_state_ptrs are the C names of these structs
struct crafter_config : Config {};
struct crafter_rules : Rules {};
struct crafter_state_ptrs : StatePtrs {};

// The C99 layouts a binding compiles against (include/crafter_hip_types.h), checked field by field against the kernels'
// own definitions (types.hpp): a drift is a build error, not a silent misread.
namespace cabi {
#include "../../include/crafter_hip_types.h"
}
#define CRAFTER_SAME_FIELD(CT, T, f) \
  static_assert(offsetof(cabi::CT, f) == offsetof(T, f) && sizeof(((cabi::CT*)0)->f) == sizeof(((T*)0)->f), #CT "." #f)
#define CRAFTER_SAME_SIZE(CT, T) static_assert(sizeof(cabi::CT) == sizeof(T) && alignof(cabi::CT) == alignof(T), #CT)
CRAFTER_SAME_SIZE(crafter_obj, Obj);
CRAFTER_SAME_FIELD(crafter_obj, Obj, type); CRAFTER_SAME_FIELD(crafter_obj, Obj, health); CRAFTER_SAME_FIELD(crafter_obj, Obj, fx);
CRAFTER_SAME_FIELD(crafter_obj, Obj, fy); CRAFTER_SAME_FIELD(crafter_obj, Obj, x); CRAFTER_SAME_FIELD(crafter_obj, Obj, y);
CRAFTER_SAME_FIELD(crafter_obj, Obj, aux); CRAFTER_SAME_FIELD(crafter_obj, Obj, pad);
CRAFTER_SAME_SIZE(crafter_item_list, ItemList);
CRAFTER_SAME_FIELD(crafter_item_list, ItemList, n); CRAFTER_SAME_FIELD(crafter_item_list, ItemList, item);
CRAFTER_SAME_FIELD(crafter_item_list, ItemList, amount); CRAFTER_SAME_FIELD(crafter_item_list, ItemList, ach);
CRAFTER_SAME_SIZE(crafter_collect_rule, CollectRule);
CRAFTER_SAME_FIELD(crafter_collect_rule, CollectRule, valid); CRAFTER_SAME_FIELD(crafter_collect_rule, CollectRule, leaves);
CRAFTER_SAME_FIELD(crafter_collect_rule, CollectRule, probability); CRAFTER_SAME_FIELD(crafter_collect_rule, CollectRule, require);
CRAFTER_SAME_FIELD(crafter_collect_rule, CollectRule, receive);
CRAFTER_SAME_SIZE(crafter_place_rule, PlaceRule);
CRAFTER_SAME_FIELD(crafter_place_rule, PlaceRule, valid); CRAFTER_SAME_FIELD(crafter_place_rule, PlaceRule, is_object);
CRAFTER_SAME_FIELD(crafter_place_rule, PlaceRule, material); CRAFTER_SAME_FIELD(crafter_place_rule, PlaceRule, ach);
CRAFTER_SAME_FIELD(crafter_place_rule, PlaceRule, where_mask); CRAFTER_SAME_FIELD(crafter_place_rule, PlaceRule, uses);
CRAFTER_SAME_SIZE(crafter_make_rule, MakeRule);
CRAFTER_SAME_FIELD(crafter_make_rule, MakeRule, valid); CRAFTER_SAME_FIELD(crafter_make_rule, MakeRule, item);
CRAFTER_SAME_FIELD(crafter_make_rule, MakeRule, gives); CRAFTER_SAME_FIELD(crafter_make_rule, MakeRule, ach);
CRAFTER_SAME_FIELD(crafter_make_rule, MakeRule, nearby_mask); CRAFTER_SAME_FIELD(crafter_make_rule, MakeRule, uses);
CRAFTER_SAME_SIZE(crafter_rules, Rules);
#define R_(f) CRAFTER_SAME_FIELD(crafter_rules, Rules, f)
R_(n_actions); R_(n_materials); R_(n_items); R_(n_achievements); R_(action_kind); R_(action_arg); R_(item_max); R_(item_init);
R_(walkable_mask); R_(player_walkable_mask); R_(arrow_walkable_mask); R_(arrow_breaks_mask);
R_(mat_water); R_(mat_grass); R_(mat_stone); R_(mat_path); R_(mat_sand); R_(mat_tree); R_(mat_lava); R_(mat_coal); R_(mat_iron);
R_(mat_diamond); R_(mat_table); R_(mat_furnace); R_(item_health); R_(item_food); R_(item_drink); R_(item_energy);
R_(item_wood_sword); R_(item_stone_sword); R_(item_iron_sword); R_(ach_wake_up); R_(ach_eat_plant); R_(ach_defeat_zombie);
R_(ach_defeat_skeleton); R_(ach_eat_cow); R_(collect); R_(place); R_(make);
#undef R_
CRAFTER_SAME_SIZE(crafter_config, Config);
#define C_(f) CRAFTER_SAME_FIELD(crafter_config, Config, f)
C_(num_envs); C_(W); C_(H); C_(view_w); C_(view_h); C_(size_w); C_(size_h); C_(unit_x); C_(unit_y); C_(local_gw); C_(local_gh);
C_(item_gw); C_(item_gh); C_(border_x); C_(border_y); C_(icon_w); C_(icon_h); C_(digit_w); C_(digit_h); C_(max_objects);
C_(nchunk_x); C_(nchunk_y); C_(length); C_(update_dist); C_(n_daylight); C_(auto_reset); C_(want_semantic); C_(render_obs);
C_(reward); C_(step_threads); C_(reset_threads); C_(gen_period);
#undef C_
CRAFTER_SAME_SIZE(crafter_env_rec, EnvRec);
#define E_(f) CRAFTER_SAME_FIELD(crafter_env_rec, EnvRec, f)
E_(mt_pos); E_(step); E_(episode); E_(nobj); E_(seed_lane); E_(nchunks_seen); E_(status); E_(inv); E_(ach); E_(hunger2);
E_(thirst2); E_(fatigue2); E_(recover2); E_(player_last_health); E_(env_last_health); E_(unlocked); E_(sleeping); E_(dhealth);
E_(new_unlocked); E_(dead); E_(done); E_(needs_reset); E_(ep_dhealth); E_(ep_unlock_steps); E_(pad);
#undef E_
CRAFTER_SAME_SIZE(crafter_pool_hdr, PoolHdr);
CRAFTER_SAME_FIELD(crafter_pool_hdr, PoolHdr, ready); CRAFTER_SAME_FIELD(crafter_pool_hdr, PoolHdr, mt_pos);
CRAFTER_SAME_FIELD(crafter_pool_hdr, PoolHdr, nobj); CRAFTER_SAME_FIELD(crafter_pool_hdr, PoolHdr, nchunks_seen);
CRAFTER_SAME_FIELD(crafter_pool_hdr, PoolHdr, pad); CRAFTER_SAME_FIELD(crafter_pool_hdr, PoolHdr, pending); CRAFTER_SAME_FIELD(crafter_pool_hdr, PoolHdr, pad2);
CRAFTER_SAME_SIZE(crafter_state_ptrs, StatePtrs);
#define S_(f) CRAFTER_SAME_FIELD(crafter_state_ptrs, StatePtrs, f)
S_(mat); S_(objmap); S_(objs); S_(mt); S_(rec); S_(chunk_order); S_(chunk_seen); S_(census); S_(semantic); S_(prof); S_(reset_q);
S_(pool_mat); S_(pool_objs); S_(pool_mt); S_(pool_hdr); S_(pool_chunk_order); S_(gen_q); S_(gen_latest); S_(terminal);
S_(pool_stats); S_(pool_perm); S_(pool_census);
#undef S_
static_assert(cabi::CRAFTER_TEX_COUNT == TEX_COUNT && cabi::CRAFTER_TEX_PLANT_RIPE == TEX_PLANT_RIPE && CRAFTER_MT_N == MT_N &&
                  CRAFTER_MAX_ITEMS == MAX_ITEMS && CRAFTER_MAX_ACH == MAX_ACH && CRAFTER_MAX_MATERIALS == MAX_MATERIALS &&
                  CRAFTER_MAX_ACTIONS == MAX_ACTIONS && CRAFTER_MAX_PLACE == MAX_PLACE && CRAFTER_MAX_MAKE == MAX_MAKE &&
                  CRAFTER_MAX_USES == MAX_USES && CRAFTER_CHUNK == CHUNK,
              "constants of crafter_hip_types.h");

namespace {

constexpr int kStepThreads = 256;    // step / render workgroup (compile-time: see WaveGfx950)
constexpr int kResetThreads = 1024;  // reset / generation workgroup
constexpr int kRequeueGrid = 256;
constexpr int kRequeueGridPooled = 8;
constexpr int kEarlyMinEnvs = 8 * 256;   // crafter_step_early_kernel from this many envs on (same-box A/B, profiles/r6_early_frame_ab.txt: 1536 envs -1.0 %, 2048 / 3072 +0.3 %, 4096 +0.1 ... +1.0 %, 8192 +1.2 %)
constexpr int kOrderMinEnvs = 5 * 256;   // the dispatch order can only matter when a launch has more workgroups than the chip holds at once (5 per CU)
constexpr int kRequeueThreads = 256;   // inline regeneration (rare): sized like a step workgroup, NOT like crafter_reset_kernel -- a
                                       // 1024-thread workgroup needs a CU with all registers free, and with the world pool's kernels
                                       // resident next to the step kernel even the EMPTY queue check would wait for one (measured: 32 us / step)
constexpr int kGenSeedThreads = 64;       // world pool: seeding, one wave per world
constexpr int kGenClassifyThreads = 256;  // terrain classification, four waves per world
constexpr int kGenResolveThreads = 64;    // ordered draws, one wave per world
constexpr int kGenSerialGrid = 2048;      // at most this many single-wave workgroups per batch kernel (they loop over the queue)
constexpr int kGenClassifyGridStep = 304; // ... behind a closed-loop step of the default geometry (round 6, final kernels: a batch's latency is what a device-wide
                                          // synchronize behind a few steps waits for -- the driver's 20-step window 56.3 -> 61.0 M env-steps/s on one box, the
                                          // steady state 65.81 -> 65.58 M; 272: 59.3 / 65.69 M; profiles/r6_headline_grid2.txt -- behind a rollout stretch 256 stays:
                                          // 320 workgroups cost the open loop 3 %, profiles/r6_classify_grid_probe.txt)
constexpr int kGenClassifyGrid = 256;     // workgroups of the classification kernel (they loop over the batch: a world is gen_classify_parts items).
                                          // About one per CU: a classification wave holds 136 VGPRs, and a batch launched at its full width (780
                                          // workgroups at 4096 envs) takes every SIMD's registers, leaving the step kernel one wave slot per SIMD
                                          // instead of five (same-box A/B: 45.0 -> 48.6 M env-steps/s going from 4096 to 256; 128: 47.3, 512: 47.0)
constexpr int kDefaultGenPeriod = 16;  // steps between generation batches (same-box A/B at 4096 envs: 8: 48.5 M, 16: 50.6 M, 32: 36.7 M env-steps/s)
constexpr int kUnpooledStretch = 16;   // crafter_step_n without the world pool: steps per launch
constexpr int kGenRing = 8;   // request-queue segments / batch events
constexpr int kGenLag = 3;    // the launch stream waits for batch j - kGenLag when batch j is launched (<= kGenRing - 2)
constexpr int kGenStreams = 2; // batches alternate between side streams, so two can be in flight
constexpr int kMaxLds = 160 * 1024;

template <int LM, int GEO, int RUL>   // LM 1: maps staged in LDS, 0: large world, maps stay in HBM (env_kernels.hpp bind_lds);
                                     // RUL 1: the uploaded rules equal the compiled-in kDefaultRules (types.hpp)
#ifndef CRAFTER_BIG_WAVES
#define CRAFTER_BIG_WAVES 6
#endif
__global__ void __launch_bounds__(kStepThreads, LM == 0 ? CRAFTER_BIG_WAVES : 1)   // (maps and slot table in global memory: 19 KB of LDS, registers decide how many share a CU)
crafter_step_kernel(Config cfg_in, TablePtrs tb, StatePtrs st, const int32_t* __restrict__ actions,
                    uint8_t* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done,
                    StepCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  typedef WaveGfx950<kStepThreads> WS;
  WS w;
  const Config cfg = GEO == 1 ? with_default_geometry(cfg_in) : GEO == 2 ? with_default_view(cfg_in) : cfg_in;   // (GEO 2: the default view on a world of any size)
  int env = (int)blockIdx.x;
  if (ctl.order_build) {   // dispatch order in use: block 0 sorts for the launch after this one, block b + 1 steps env order[b]
    if (env == 0) {
      build_order(cfg, tb, ctl.order_build, ctl.next_step, (uint32_t*)smem);
      return;
    }
    env -= 1;
    if (ctl.order) env = ctl.order[env];
  }
  if constexpr (GEO == 1)   // max_objects == 256: one-byte slot ids, 4 KB less LDS per environment
    step_body<WS, LM, RUL, uint8_t>(w, smem, env, cfg, tb, st, actions, obs, reward, done, ctl);
  else
    step_body<WS, LM, RUL, typename StepSlot<LM>::type>(w, smem, env, cfg, tb, st, actions, obs, reward, done, ctl);
}


// The default instance for batches well beyond what the chip holds at once (kEarlyMinEnvs): the same body on a wave policy that
// lets the frame begin before the rules end (wave_gfx950.hpp kEarlyFrame, render.hpp early_frame).
__global__ void __launch_bounds__(kStepThreads)
crafter_step_early_kernel(Config cfg_in, TablePtrs tb, StatePtrs st, const int32_t* __restrict__ actions,
                          uint8_t* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done, StepCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  typedef WaveGfx950<kStepThreads, 0, 1> WS;
  WS w;
  const Config cfg = with_default_geometry(cfg_in);
  int env = (int)blockIdx.x;
  if (ctl.order_build) {   // (dispatch order: as crafter_step_kernel)
    if (env == 0) {
      build_order(cfg, tb, ctl.order_build, ctl.next_step, (uint32_t*)smem);
      return;
    }
    env -= 1;
    if (ctl.order) env = ctl.order[env];
  }
  step_body<WS, 1, 1, uint8_t>(w, smem, env, cfg, tb, st, actions, obs, reward, done, ctl);
}

// The default instance for batches of at most two envs per CU (kWideMaxEnvs -- one GPU's share of configs[2]): 512 threads
// per env.  A launch of so few envs lasts as long as its slowest env's step, most of the CUs' wave slots are empty, and the
// frame is drawn by eight waves instead of four.  Same body, same LDS layout.  Measured (profiles/r4zy_wide_ab.txt): kernel
// 27.8 -> 26.3 us at 512 envs (+4.8 % env-steps/s), +5 % at 256; at 768 envs -5 %, at 1024 -18 % (the frame's phases are
// barrier to barrier: eight waves shorten them far less than they crowd a CU that holds three or four envs).
constexpr int kWideThreads = 512;
constexpr int kWideMaxEnvs = 2 * 256;
__global__ void __launch_bounds__(kWideThreads, 6)
crafter_step_wide_kernel(Config cfg_in, TablePtrs tb, StatePtrs st, const int32_t* __restrict__ actions,
                         uint8_t* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done, StepCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  typedef WaveGfx950<kWideThreads> WS;
  WS w;
  const Config cfg = with_default_geometry(cfg_in);
  step_body<WS, 1, 1, uint8_t>(w, smem, (int)blockIdx.x, cfg, tb, st, actions, obs, reward, done, ctl);
}

// Split step of the default instance (env_kernels.hpp "Split step"): the rule half, one wave per env ...
constexpr int kRulesThreads = 64;
__global__ void __launch_bounds__(kRulesThreads)
crafter_rules_kernel(Config cfg_in, TablePtrs tb, StatePtrs st, const int32_t* __restrict__ actions,
                     uint8_t* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done, StepCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  WaveGfx950<kRulesThreads> w;
  const Config cfg = with_default_geometry(cfg_in);
  step_body<WaveGfx950<kRulesThreads>, 1, 1, LaneSlots, 1>(w, smem, (int)blockIdx.x, cfg, tb, st, actions, obs, reward, done, ctl);
}

// ... and the frame half, four waves per env, from the frame record the rule half left behind.
__global__ void __launch_bounds__(kStepThreads)
crafter_frame_kernel(Config cfg_in, TablePtrs tb, StatePtrs st, uint8_t* __restrict__ obs, uint32_t* __restrict__ night_px) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  WaveGfx950<kStepThreads> w;
  const Config cfg = with_default_geometry(cfg_in);
  frame_body(w, smem, (int)blockIdx.x, cfg, tb, st, obs, night_px);
}

// One queue entry each.  Kept inlined on purpose: as real functions they need stack copies of the
// argument structs, and that scratch set-up costs the (almost always empty) requeue kernel +10 us
// per step -- measured 10.8 M vs 12.8 M env-steps/s; the spills of the inlined loop only hurt the
// rare fallback path.
__device__ __forceinline__ void gen_one(uint8_t* smem, int env, int episode, uint32_t seq, const Config& cfg,
                                                  const TablePtrs& tb, const StatePtrs& st) {
  WaveGfx950<kResetThreads> w;
  gen_body(w, smem, env, episode, seq, cfg, tb, st);
}

__device__ __forceinline__ int reset_one(uint8_t* smem, int env, const Config& cfg, const TablePtrs& tb,
                                                   const StatePtrs& st, uint8_t* obs, int gen_parity) {
  WaveGfx950<kResetThreads> w;
  return reset_body(w, smem, env, cfg, tb, st, obs, gen_parity);
}

// Regenerates the envs queued by the step kernel (auto-reset without a ready pooled world): a small
// grid walks the queue of this step's parity and clears the other parity's counter for the next step.
__global__ void __launch_bounds__(kRequeueThreads, 5)
crafter_requeue_reset_kernel(Config cfg, TablePtrs tb, StatePtrs st, int parity, int gen_parity,
                             uint8_t* __restrict__ obs) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int32_t* q = st.reset_q + (size_t)parity * (cfg.num_envs + 4);
  int count = q[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) st.reset_q[(size_t)(1 - parity) * (cfg.num_envs + 4)] = 0;
  for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
    WaveGfx950<kRequeueThreads> w;
    reset_body(w, smem, q[4 + k], cfg, tb, st, obs, gen_parity);
    __syncthreads();
  }
}

// Env.reset.  With the world pool on (gen_parity >= 0) the workgroup goes on to generate the NEXT episode's world into
// the env's pool entry, stamped with batch sequence 1 (trusted from the start: this kernel precedes every later step in
// stream order) -- that world may be needed a few dozen steps from now, earlier than any batch could deliver it.
__global__ void __launch_bounds__(kResetThreads)
crafter_reset_kernel(Config cfg, TablePtrs tb, StatePtrs st, const uint8_t* __restrict__ mask,
                     int gen_parity, uint8_t* __restrict__ obs) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int env = (int)blockIdx.x;
  if (mask && !mask[env]) return;
  int episode = reset_one(smem, env, cfg, tb, st, obs, -1);
  if (gen_parity < 0) return;
  using WR = WaveGfx950<kResetThreads>;
  if (threadIdx.x == 0 && WR::agent_load(st.gen_latest + env) < episode + 1) WR::agent_store(st.gen_latest + env, (int32_t)(episode + 1));
  __threadfence();
  __syncthreads();
  // (not if the entry already holds that world -- a batch delivered it before a reset in mid-episode -- or if an older
  // generation into the entry is still queued: the env then regenerates inline when it gets there.  crafter_reset has
  // ordered this kernel behind every batch in flight, so whatever is stamped is complete.)
  PoolHdr* next = st.pool_hdr + pool_slot(cfg, env, episode + 1);
  int32_t pend = WR::agent_load(&next->pending);
  bool have = gen_done_already<WR>(cfg, st, env, episode + 1), busy = pend != 0 && pend != episode + 1;
  __syncthreads();
  if (!have && !busy) gen_one(smem, env, episode + 1, 1u, cfg, tb, st);
  // ... and asks the pool for the one after it right away (it is due two episodes from now; waiting for the first
  // auto-reset to ask would leave a short second episode without its successor)
  WaveGfx950<kResetThreads> w;
  request_generation(w, cfg, st, gen_parity, env, episode + 2);
}

// World pool generation (side streams): three kernels per batch, each walking the batch's segment of the request queue
// (env_kernels.hpp gen_seed_body / gen_classify_body / gen_resolve_body).  GEO as for the step kernel.
template <int GEO>
__global__ void __launch_bounds__(kGenSeedThreads)
crafter_gen_seed_kernel(Config cfg_in, TablePtrs tb, StatePtrs st, int parity, int prio) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  WaveGfx950<kGenSeedThreads>::set_priority(prio);   // (pool_schedule: one wave per world, a chain -- see there)
  const Config cfg = GEO ? with_default_geometry(cfg_in) : cfg_in;
  const int32_t* q = st.gen_q + (size_t)parity * gen_q_stride(cfg);
  int count = q[0];
  if (count > gen_q_capacity(cfg)) count = gen_q_capacity(cfg);
  WaveGfx950<kGenSeedThreads> w;
  for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
    gen_seed_body(w, smem, q[4 + 2 * k], q[4 + 2 * k + 1], cfg, tb, st);
    __syncthreads();
  }
}

#ifndef CRAFTER_CLASSIFY_WAVES
#define CRAFTER_CLASSIFY_WAVES 6
#endif
template <int GEO>
__global__ void __launch_bounds__(kGenClassifyThreads, GEO ? CRAFTER_CLASSIFY_WAVES : 1)   // (the default geometry only: configs[3] is bound by its generator, profiles/r6_classify_bounds.txt)
crafter_gen_classify_kernel(Config cfg_in, TablePtrs tb, StatePtrs st, int parity, int prio) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  WaveGfx950<kGenClassifyThreads>::set_priority(prio);
  const Config cfg = GEO ? with_default_geometry(cfg_in) : cfg_in;
  const int32_t* q = st.gen_q + (size_t)parity * gen_q_stride(cfg);
  int count = q[0];
  if (count > gen_q_capacity(cfg)) count = gen_q_capacity(cfg);
  WaveGfx950<kGenClassifyThreads> w;
  const int parts = gen_classify_parts(cfg);
  for (int item = (int)blockIdx.x; item < count * parts; item += (int)gridDim.x) {
    int k = item / parts;
    gen_classify_body(w, smem, q[4 + 2 * k], q[4 + 2 * k + 1], item - k * parts, parts, cfg, tb, st);
    __syncthreads();
  }
}

template <int GEO>
__global__ void __launch_bounds__(kGenResolveThreads)
crafter_gen_resolve_kernel(Config cfg_in, TablePtrs tb, StatePtrs st, int parity, uint32_t seq, int prio) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  WaveGfx950<kGenResolveThreads>::set_priority(prio);
  const Config cfg = GEO ? with_default_geometry(cfg_in) : cfg_in;
  const int32_t* q = st.gen_q + (size_t)parity * gen_q_stride(cfg);
  int count = q[0];
  if (count > gen_q_capacity(cfg)) count = gen_q_capacity(cfg);
  WaveGfx950<kGenResolveThreads> w;
  for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
    gen_resolve_body(w, smem, q[4 + 2 * k], q[4 + 2 * k + 1], seq, cfg, tb, st);
    __syncthreads();
  }
}

#ifdef CRAFTER_PROBES
// TIMING PROBE (CRAFTER_PROBE_FREE_GEN=2): workgroups that only OCCUPY what a generation kernel's workgroups occupy -- threads,
// registers (the launch bound's share), LDS -- for `us` microseconds each, asleep: what does the generator cost the step loop by
// being resident, as opposed to by what it executes?
template <int BIG>   // BIG: the wave holds 112 vector registers (a classification wave's allocation), else a handful
__global__ void __launch_bounds__(kStepThreads)
crafter_gen_occupy_kernel(int us, int items, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (BIG) asm volatile("v_mov_b32 v111, 0" ::: "v111");
  for (int k = (int)blockIdx.x; k < items; k += (int)gridDim.x) {
    uint64_t t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < (long long)us * 100) __builtin_amdgcn_s_sleep(32);   // 100 MHz constant clock
  }
  if (sink && threadIdx.x == 0 && smem[0] == 0xA5) sink[0] = 1;   // (keeps the LDS allocation alive)
}

// TIMING PROBE (CRAFTER_PROBE_FREE_GEN=1, probe builds only -- results are WRONG by construction): what would the step loop
// do with a generator that costs nothing?  Instead of the three generation kernels a batch launches this one, which stamps every
// requested pool entry as ready without generating it: the entry keeps whatever world it held (an entry that never held one
// takes a copy of its env's other entry first -- crafter_reset_kernel filled that), so finished envs adopt worlds of the right
// shape and every step does its usual work, while no generation kernel runs beside it (VERDICT r5 #1a).
template <int GEO>
__global__ void __launch_bounds__(kStepThreads)
crafter_gen_stamp_kernel(Config cfg_in, TablePtrs tb, StatePtrs st, int parity, uint32_t seq) {
  typedef WaveGfx950<kStepThreads> WS;
  const Config cfg = GEO ? with_default_geometry(cfg_in) : cfg_in;
  const int32_t* q = st.gen_q + (size_t)parity * gen_q_stride(cfg);
  int count = q[0];
  if (count > gen_q_capacity(cfg)) count = gen_q_capacity(cfg);
  const int cells = cfg.W * cfg.H, nch = cfg.nchunk_x * cfg.nchunk_y;
  for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
    int env = q[4 + 2 * k], episode = q[4 + 2 * k + 1];
    size_t slot = pool_slot(cfg, env, episode), other = pool_slot(cfg, env, episode + 1);
    PoolHdr* h = st.pool_hdr + slot;
    bool skip = !gen_wanted<WS>(st, env, episode) || gen_done_already<WS>(cfg, st, env, episode);
    bool never = (WS::agent_load(&h->ready) >> 32) == 0;
    __syncthreads();
    if (!skip && never) {
      const uint4* a = (const uint4*)(st.pool_mat + other * cells);
      uint4* b = (uint4*)(st.pool_mat + slot * cells);
      for (int i = (int)threadIdx.x; i < cells / 16; i += kStepThreads) b[i] = a[i];
      const uint4* oa = (const uint4*)(st.pool_objs + other * cfg.max_objects);
      uint4* ob = (uint4*)(st.pool_objs + slot * cfg.max_objects);
      for (int i = (int)threadIdx.x; i < cfg.max_objects; i += kStepThreads) ob[i] = oa[i];
      for (int i = (int)threadIdx.x; i < MT_N; i += kStepThreads) st.pool_mt[slot * MT_N + i] = st.pool_mt[other * MT_N + i];
      for (int i = (int)threadIdx.x; i < nch; i += kStepThreads) st.pool_chunk_order[slot * nch + i] = st.pool_chunk_order[other * nch + i];
      for (int i = (int)threadIdx.x; i < nch * 5; i += kStepThreads) st.pool_census[slot * nch * 5 + i] = st.pool_census[other * nch * 5 + i];
      if (threadIdx.x == 0) {
        const PoolHdr* o = st.pool_hdr + other;
        h->mt_pos = o->mt_pos; h->nobj = o->nobj; h->nchunks_seen = o->nchunks_seen; h->pad = o->pad;
      }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      if (!skip) WS::agent_store(&h->ready, ((uint64_t)seq << 32) | (uint32_t)episode);
      gen_retire<WS>(cfg, st, env, episode);
    }
    __syncthreads();
  }
}
#endif

__global__ void __launch_bounds__(kStepThreads)
crafter_render_kernel(Config cfg, TablePtrs tb, StatePtrs st, const uint8_t* __restrict__ mask,
                      uint8_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int env = (int)blockIdx.x;
  if (mask && !mask[env]) return;
  WaveGfx950<kStepThreads> w;
  render_body(w, smem, env, cfg, tb, st, out);
}

// Builds the renderer's static block once per table upload (one workgroup).
__global__ void __launch_bounds__(kStepThreads)
crafter_init_tables_kernel(Config cfg, TablePtrs tb, uint8_t* dst) {
  WaveGfx950<kStepThreads> w;
  Env<WaveGfx950<kStepThreads>> e(w, cfg, tb);
  RenderTarget rt = obs_target<WaveGfx950<kStepThreads>>(cfg, tb, nullptr, 0);
  Renderer<WaveGfx950<kStepThreads>> r(e, rt, dst, nullptr, nullptr);
  r.build_static(dst);
}

// ... and the lit sprite rows behind it (render.hpp render_lit_sprite_bytes: 78 MB), one workgroup per (asleep, step)
__global__ void __launch_bounds__(kStepThreads)
crafter_init_lit_sprites_kernel(Config cfg, TablePtrs tb, uint8_t* dst) {
  WaveGfx950<kStepThreads> w;
  Env<WaveGfx950<kStepThreads>> e(w, cfg, tb);
  RenderTarget rt = obs_target<WaveGfx950<kStepThreads>>(cfg, tb, nullptr, 0);
  Renderer<WaveGfx950<kStepThreads>> r(e, rt, dst, nullptr, nullptr);
  r.build_lit_sprites(dst, (int)blockIdx.x, (int)gridDim.x);
}

// Unit-test access to the device's own transcendental-free noise and to the two libm calls of worldgen.py:25-27 as the
// generation kernels evaluate them (the pinned exp_cr of worldgen.hpp / sqrt): crafter_debug_eval.  mode 0: out = noise3(x, y, z)
// with the permutation perm[256]; 1: out = 1 / (1 + exp_cr(-x)); 2: out = 4 - sqrt(x); 3: out = exp_cr(x).
__global__ void __launch_bounds__(kStepThreads)
crafter_debug_eval_kernel(const uint8_t* __restrict__ perm, const double* __restrict__ x, const double* __restrict__ y,
                          const double* __restrict__ z, double* __restrict__ out, long long n, int mode) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* s_perm = smem;
  uint8_t* s_pg3 = smem + 256;
  SimplexLds* s_tab = (SimplexLds*)(smem + 512);
  if (mode == 0) {
    for (int i = (int)threadIdx.x; i < 256; i += kStepThreads) {
      s_perm[i] = perm[i];
      s_pg3[i] = (uint8_t)(perm[i] % 24);
    }
    simplex_fill_tables(s_tab, [&](int n, auto body) { for (int i = (int)threadIdx.x; i < n; i += kStepThreads) body(i); });
  }
  __syncthreads();
  Simplex<WaveGfx950<kStepThreads>> sx{s_perm, s_pg3, s_tab};
  for (long long i = (long long)blockIdx.x * kStepThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kStepThreads) {
    double v;
    if (mode == 0) v = sx.noise3(x[i], y[i], z[i]);
    else if (mode == 1) v = 1 / (1 + exp_cr(-x[i]));
    else if (mode == 3) v = exp_cr(x[i]);
    else v = 4 - __builtin_sqrt(x[i]);
    out[i] = v;
  }
}

thread_local std::string g_create_error;

// Experiment knobs.  The shipped library reads NONE of them: only a probe build (tools/ab_make.sh <name> tree -DCRAFTER_PROBES ->
// gpurun_ab/<name>.so, loaded through CRAFTER_HIP_LIB by the tools under tools/) looks at the environment.  What stays in every
// build are the four DISPATCH OVERRIDES documented in include/crafter_hip.h (CRAFTER_SPLIT, CRAFTER_ORDER, CRAFTER_STEP_WIDE, CRAFTER_STEP_EARLY):
// they choose between kernels the product ships, and the GPU tests use them to drive each of those at every batch size.
static const char* probe_env(const char* name) {
#ifdef CRAFTER_PROBES
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

}  // namespace

// The renderer's derived tables (static block, inventory cells, lit rows, night pixel records: 5.7 MB for the default
// geometry, render.hpp; 85 MB while they held lit sprite rows too) depend on the frame geometry, the rules' sizes and the uploaded tables only: handles
// with the same inputs on the same device share one allocation -- a process with a few hundred `crafter_amd.Env` objects
// (each its own handle) would otherwise hold that many copies, and build them.
struct SharedBlock {
  void* ptr = nullptr;
  int refs = 0;
};
static std::mutex g_blocks_mutex;
static std::map<std::string, SharedBlock> g_blocks;   // key: device | content hash | size

static uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p;
  for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}
// a second, independent 64-bit digest of the same bytes (the cache key carries both: ADVICE r2 -- a hit is trusted without
// comparing the megabytes it stands for)
static uint64_t mix64(uint64_t h, const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p;
  for (size_t i = 0; i < n; i++) {
    h += b[i] + 0x9e3779b97f4a7c15ull;
    h = (h ^ (h >> 30)) * 0xbf58476d1ce4e5b9ull;
    h = (h ^ (h >> 27)) * 0x94d049bb133111ebull;
    h ^= h >> 31;
  }
  return h;
}

struct crafter_handle {
  Config cfg;
  TablePtrs tb;
  StatePtrs st;
  bool have_tables = false;
  bool have_state = false;
  std::vector<void*> owned;   // device allocations of the handle (tables)
  std::string shared_block_key;   // its entry in g_blocks (the renderer's derived tables), empty: none
  int lds_bytes = 0;
  int reset_lds_bytes = 0;
  int step_lds_bytes = 0;   // the default-geometry step kernel keeps one-byte slot ids (env_kernels.hpp lds_layout); worlds whose maps
                            // stay in HBM step in big_layout
  bool default_rules = false;   // the uploaded rules are byte-identical to kDefaultRules
  int gen_resolve_lds_bytes = 0;
  int split = -1;                         // the default instance steps as rules kernel (+ frame kernel): -1 = when no frame is drawn
                                          // (with frames the fused step kernel is faster at every batch size measured: round 3,
                                          // 4096 envs: fused 55.4 M, split pair 42-43 M, overlapped pair 31.5 M env-steps/s),
                                          // CRAFTER_SPLIT=0 / 1 = never / always
  int rules_lds_bytes = 0, frame_lds_bytes = 0;
  int requeue_grid = kRequeueGridPooled;  // CRAFTER_REQUEUE_GRID (A/B): workgroups of the inline-regeneration kernel while the pool runs
  int gen_lag = kGenLag;                  // CRAFTER_GEN_LAG (A/B): back-pressure distance in batches, 1 .. kGenRing - 2
  int classify_grid = kGenClassifyGrid;   // CRAFTER_GEN_CLASSIFY_GRID: workgroups of the classification kernel (A/B)
  int classify_grid_step = kGenClassifyGrid;   // ... of a batch launched behind a closed-loop step (kGenClassifyGridStep for the default geometry)
  int gen_lds_bytes = 0;   // the generation kernel never draws: no renderer region (4 step workgroups + 1 generator per CU)
  long long steps = 0;
  std::string err;
  // world pool (asynchronous generation on a side stream)
  bool pool = false;
  // Schedule: batch j is launched on side stream j % kGenStreams every gen_period steps over request-queue
  // segment j % kGenRing and records event j % kGenRing behind itself.  Trust (safe_seq: the step kernel only adopts
  // worlds of batches <= safe_seq and regenerates inline otherwise -- unobservable: same generator, same
  // (seed, episode)) advances two ways: every crafter_step polls the oldest untrusted batch's event (non-blocking,
  // so a batch is usually trusted one period after its launch), and when batch j is launched the launch stream is
  // ordered behind batch j - kGenLag (hipStreamWaitEvent: the host does not block).  That wait is the back-pressure
  // that keeps generation ahead of demand: generation workgroups need a whole CU's registers and only get one when a
  // step kernel drains.  The burst a reset of all envs would cause does not exist: crafter_reset_kernel generates the
  // next world itself (sequence number 1).
  hipStream_t side[2] = {nullptr, nullptr};
  // dispatch order of the step launch (StepCtl::order): slow envs first.  Only where it can pay -- more envs than the chip
  // holds at once -- and only for the fused step kernel
  int32_t* order = nullptr;       // [2][N]: launch k reads half k & 1 (built during launch k - 1), builds half (k + 1) & 1
  int32_t* next_step = nullptr;   // [N]
  uint64_t ordered_launches = 0;
  const int32_t* order_override = nullptr;   // diagnostics: crafter_debug_set_dispatch_order
  int lds_pad = 0;                        // CRAFTER_LDS_PAD
  bool lds_pad_given = false;
  int rollout_order = 1;                  // CRAFTER_ROLLOUT_ORDER=0 (A/B): rollout launches in arrival order
  int rollout_lds_pad = 0;                // CRAFTER_ROLLOUT_LDS_PAD (A/B): extra LDS per workgroup of crafter_rollout_kernel<1, 1, 1> (26,872 B: six per CU)
  int32_t* stalled_at = nullptr;          // crafter_step_n: per env, the step of the call it stopped at for want of a world (-1: none)
  uint32_t* night_px = nullptr;           // split step: scratch of the frame kernel, a night frame's pixels in noise-stream order per env
  uint32_t* noise_raw = nullptr;          // fused step: the MT19937 states a night frame's noise comes from, generated ahead of the rules
                                          // (env_kernels.hpp noise_chain), kNoiseStates * 624 words per env; CRAFTER_NOISE_AHEAD=0: off (A/B)
  int noise_ahead = 1;
  hipStream_t aux = nullptr;              // split step: the regeneration kernel runs here, beside the frame kernel
  int wide = -1;                          // CRAFTER_STEP_WIDE=0|1: never / always the 512-thread step kernel of the default instance (default: batches of <= kWideMaxEnvs)
  hipEvent_t ev_rules = nullptr, ev_requeue = nullptr;
  hipEvent_t ev_main = nullptr;
  int probe_lds[3] = {-1, -1, -1}, probe_big = 0;   // CRAFTER_PROBE_OCCUPY_LDS="seed,classify,resolve" bytes; CRAFTER_PROBE_OCCUPY_BIG=1: 112 VGPRs per classify / resolve wave
  int probe_worlds = 0, probe_us[3] = {80, 120, 300};   // CRAFTER_PROBE_OCCUPY="worlds,seed_us,classify_item_us,resolve_us"
  int probe_free_gen = 0;            // CRAFTER_PROBE_FREE_GEN (probe builds): batches stamp their requests ready without generating
  int early_frame = -1;                   // CRAFTER_STEP_EARLY=0|1: never / always crafter_step_early_kernel for the default instance (default: batches of at least kEarlyMinEnvs envs)
  int gen_classify_prio = 0;              // CRAFTER_GEN_CLASSIFY_PRIO (probe builds): s_setprio of the classification kernel's waves
  int gen_serial_prio = -1;               // CRAFTER_GEN_SERIAL_PRIO (A/B): s_setprio of the seeding / draw kernels of every batch; -1: 2 behind a rollout stretch, 1 behind a step
  bool fold_main_event = true;            // CRAFTER_FOLD_MAIN_EVENT=0 (A/B): crafter_step_n marks the launch stream with a packet of its own
  hipEvent_t ev_gen[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint32_t batches = 1;        // launched so far; sequence number 1 = worlds generated inside crafter_reset_kernel
  uint32_t safe_seq = 1;       // trusted so far: every batch <= safe_seq is known complete ON safe_stream (or to the host)
  uint32_t polled_seq = 1;     // ... known complete to the host (event query): valid whatever stream the next call uses
  hipStream_t safe_stream = nullptr;   // the stream the stream-side waits behind safe_seq were enqueued on
  bool have_safe_stream = false;
  hipStream_t last_stream = nullptr;   // the stream of the last call that enqueued kernels (adopt_stream: ADVICE r4)
  bool have_last_stream = false;
  hipEvent_t ev_switch = nullptr;
  bool pool_failed = false;    // a HIP call of the scheduler failed: no more batches, finished envs regenerate inline
  std::string pool_err;
  int gen_parity = 0;          // segment collecting requests now
  int steps_since_gen = 0;
  int gen_period = 8;
  // optional per-kernel timing (HIP events on the launch stream)
  bool timing = false;
  std::vector<hipEvent_t> events;   // triples: before step, between, after reset
};

static int fail(crafter_handle* h, const std::string& msg) {
  if (h)
    h->err = msg;
  else
    g_create_error = msg;
  return 1;
}

static int hip_fail(crafter_handle* h, const char* what, hipError_t e) {
  return fail(h, std::string(what) + ": " + hipGetErrorString(e));
}

extern "C" {

void crafter_struct_sizes(int32_t out[6]) {
  out[0] = sizeof(Obj);
  out[1] = sizeof(EnvRec);
  out[2] = sizeof(Rules);
  out[3] = sizeof(Config);
  out[4] = sizeof(StatePtrs);
  out[5] = sizeof(TablePtrs);
}

int32_t crafter_abi_version(void) { return 7; }

int crafter_create(const crafter_config* cfg, crafter_handle** out) {
  if (!cfg || !out) return fail(nullptr, "crafter_create: null argument");
  const Config& c = *cfg;
  if (c.num_envs < 1) return fail(nullptr, "crafter_create: num_envs < 1");
  if (c.W < 1 || c.H < 1 || c.W * c.H > 4 * 65536) return fail(nullptr, "crafter_create: bad area");
  if (c.max_objects < 2 || c.max_objects > 65535) return fail(nullptr, "crafter_create: bad max_objects");
  if (c.unit_x < 1 || c.unit_y < 1 || c.local_gw < 1 || c.local_gh < 1)
    return fail(nullptr, "crafter_create: bad view geometry");
  if (c.size_w < 1 || c.size_h < 1 || (long long)c.size_w * c.size_h > (1 << 22))   // pixel offsets use 24-bit multiplies
    return fail(nullptr, "crafter_create: bad image size (at most 2^22 pixels)");
  if (render_static_bytes(c) + sprite_rows_bytes(c) > kRenderStaticBound)
    return fail(nullptr, "crafter_create: image size too large for the renderer's LDS tables (width + height of the view in pixels)");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count < 1)
    return fail(nullptr, "crafter_create: no HIP device visible (this library has no CPU path)");
  crafter_handle* h = new crafter_handle();
  h->cfg = c;
  h->lds_bytes = lds_layout(c).total;
  h->reset_lds_bytes = big_reset_layout(c).total;   // Env.reset / regeneration kernels (= lds_bytes unless the maps stay in HBM)
  h->step_lds_bytes = is_default_geometry(c) ? lds_layout(c, 1).total : !lds_layout(c).maps_in_lds ? big_layout(c).total : h->lds_bytes;
  if (const char* pad = probe_env("CRAFTER_LDS_PAD")) {   // occupancy experiments: unused extra LDS per workgroup
    h->lds_pad = atoi(pad) > 0 ? atoi(pad) : 0;
    h->lds_pad_given = true;
  }
  h->step_lds_bytes += h->lds_pad;
  if (const char* v = probe_env("CRAFTER_ROLLOUT_ORDER")) h->rollout_order = atoi(v) != 0;
  if (const char* v = probe_env("CRAFTER_FOLD_MAIN_EVENT")) h->fold_main_event = atoi(v) != 0;
  if (const char* v = probe_env("CRAFTER_GEN_SERIAL_PRIO")) h->gen_serial_prio = atoi(v);
  if (const char* v = probe_env("CRAFTER_GEN_CLASSIFY_PRIO")) h->gen_classify_prio = atoi(v);
  if (const char* v = probe_env("CRAFTER_PROBE_FREE_GEN")) h->probe_free_gen = atoi(v);
  if (const char* v = probe_env("CRAFTER_PROBE_OCCUPY_LDS")) sscanf(v, "%d,%d,%d", &h->probe_lds[0], &h->probe_lds[1], &h->probe_lds[2]);
  if (const char* v = probe_env("CRAFTER_PROBE_OCCUPY_BIG")) h->probe_big = atoi(v);
  if (const char* v = probe_env("CRAFTER_PROBE_OCCUPY")) sscanf(v, "%d,%d,%d,%d", &h->probe_worlds, &h->probe_us[0], &h->probe_us[1], &h->probe_us[2]);
  if (const char* pad = probe_env("CRAFTER_ROLLOUT_LDS_PAD")) h->rollout_lds_pad = atoi(pad) > 0 ? atoi(pad) : 0;   // ... of the resident rollout kernel
  h->gen_lds_bytes = big_reset_layout(c).total_no_render;   // fused generation appended to crafter_reset_kernel runs in that kernel's LDS
  h->gen_resolve_lds_bytes = gen_resolve_layout(c).total;
  if (const char* v = probe_env("CRAFTER_GEN_LAG")) h->gen_lag = atoi(v) >= 1 && atoi(v) <= kGenRing - 2 ? atoi(v) : kGenLag;
  if (const char* v = getenv("CRAFTER_SPLIT")) h->split = atoi(v) < 0 ? -1 : atoi(v) != 0 ? 1 : 0;
  if (const char* v = probe_env("CRAFTER_NOISE_AHEAD")) h->noise_ahead = atoi(v) != 0;
  if (const char* v = getenv("CRAFTER_STEP_WIDE")) h->wide = atoi(v) != 0 ? 1 : 0;
  if (const char* v = getenv("CRAFTER_STEP_EARLY")) h->early_frame = atoi(v) != 0 ? 1 : 0;
  if (const char* v = probe_env("CRAFTER_REQUEUE_GRID")) h->requeue_grid = atoi(v) >= 1 && atoi(v) <= kRequeueGrid ? atoi(v) : kRequeueGridPooled;
  // large worlds (maps in HBM): two classification workgroups per CU -- their step workgroups leave the registers, and a batch
  // is sixteen times the cells (8192 x 256x256, r4i: 256 / 512 / 1024 workgroups = 9.18 / 10.14 / 9.16 M env-steps/s)
  if (!lds_layout(c).maps_in_lds) h->classify_grid = 2 * kGenClassifyGrid;
  h->classify_grid_step = is_default_geometry(c) ? kGenClassifyGridStep : h->classify_grid;
  if (const char* v = probe_env("CRAFTER_GEN_CLASSIFY_GRID")) h->classify_grid = h->classify_grid_step = atoi(v) > 0 ? atoi(v) : h->classify_grid;
  if (h->lds_bytes > kMaxLds) {
    std::string msg = "crafter_create: one environment needs " + std::to_string(h->lds_bytes) +
                      " B of LDS (> 160 KiB): area / max_objects too large for the LDS-resident kernels";
    delete h;
    return fail(nullptr, msg);
  }
  if ((c.step_threads != 0 && c.step_threads != kStepThreads) || (c.reset_threads != 0 && c.reset_threads != kResetThreads)) {
    delete h;
    return fail(nullptr, "crafter_create: workgroup sizes are fixed in this build (step_threads 0 or " +
                             std::to_string(kStepThreads) + ", reset_threads 0 or " + std::to_string(kResetThreads) + ")");
  }
  if (h->lds_bytes > 64 * 1024) {   // large worlds only: the generic instances (the default geometry needs 31 KB)
    const void* big[] = {(const void*)crafter_step_kernel<0, 0, 0>, (const void*)crafter_step_kernel<0, 2, 1>, (const void*)crafter_step_kernel<1, 0, 0>,
                         (const void*)crafter_reset_kernel,         (const void*)crafter_gen_resolve_kernel<0>,
                         (const void*)crafter_requeue_reset_kernel, (const void*)crafter_render_kernel};
    hipError_t er = rollout_allow_lds(h->lds_bytes);   // the rollout kernels live in crafter_rollout.hip
    if (er != hipSuccess) {
      std::string msg = std::string("crafter_create: hipFuncSetAttribute(rollout kernels): ") + hipGetErrorString(er);
      delete h;
      return fail(nullptr, msg);
    }
    for (const void* f : big) {
      hipError_t ea = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes);
      if (ea != hipSuccess) {
        std::string msg = std::string("crafter_create: hipFuncSetAttribute(MaxDynamicSharedMemorySize, ") +
                          std::to_string(h->lds_bytes) + "): " + hipGetErrorString(ea);
        delete h;
        return fail(nullptr, msg);
      }
    }
  }
  {   // dispatch order (see crafter_handle::order).  CRAFTER_ORDER=0 / 1: never / whenever possible (A/B)
    const char* v = getenv("CRAFTER_ORDER");
    int want = v ? atoi(v) : -1;
    bool pays = c.num_envs > kOrderMinEnvs;
    if (c.num_envs <= 64 * kStepThreads && (want > 0 || (want < 0 && pays))) {
      hipError_t ea = hipMalloc((void**)&h->order, 2 * (size_t)c.num_envs * sizeof(int32_t));
      if (ea == hipSuccess) ea = hipMalloc((void**)&h->next_step, (size_t)c.num_envs * sizeof(int32_t));
      if (ea == hipSuccess) ea = hipMemset(h->next_step, 0, (size_t)c.num_envs * sizeof(int32_t));
      if (ea != hipSuccess) {
        std::string msg = std::string("crafter_create: dispatch order buffers: ") + hipGetErrorString(ea);
        delete h;
        return fail(nullptr, msg);
      }
      h->owned.push_back(h->order);
      h->owned.push_back(h->next_step);
    }
  }
  if (c.auto_reset) {   // (a failure here only costs the overlap: the regeneration kernel then stays on the launch stream)
    if (hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_rules, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_requeue, hipEventDisableTiming) != hipSuccess) {
      if (h->aux) (void)hipStreamDestroy(h->aux);
      h->aux = nullptr;
    }
  }
  if (c.auto_reset && c.gen_period >= 0) {
    h->pool = true;
    h->gen_period = c.gen_period > 0 ? c.gen_period : kDefaultGenPeriod;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = least priority: generation yields to stepping
    bool ok = hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < kGenStreams && ok; i++)
      ok = hipStreamCreateWithPriority(&h->side[i], hipStreamNonBlocking, lo) == hipSuccess;
    for (int i = 0; i < kGenRing && ok; i++)
      ok = hipEventCreateWithFlags(&h->ev_gen[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      delete h;
      return fail(nullptr, "crafter_create: cannot create the world-pool stream / events");
    }
  }
  *out = h;
  return 0;
}

void crafter_destroy(crafter_handle* h) {
  if (!h) return;
  for (int i = 0; i < kGenStreams; i++)
    if (h->side[i]) {
      (void)hipStreamSynchronize(h->side[i]);
      (void)hipStreamDestroy(h->side[i]);
    }
  if (h->aux) {
    (void)hipStreamSynchronize(h->aux);
    (void)hipStreamDestroy(h->aux);
  }
  if (h->ev_rules) (void)hipEventDestroy(h->ev_rules);
  if (h->ev_requeue) (void)hipEventDestroy(h->ev_requeue);
  if (h->ev_main) (void)hipEventDestroy(h->ev_main);
  if (h->ev_switch) (void)hipEventDestroy(h->ev_switch);
  for (int i = 0; i < kGenRing; i++)
    if (h->ev_gen[i]) (void)hipEventDestroy(h->ev_gen[i]);
  for (void* p : h->owned) (void)hipFree(p);
  if (!h->shared_block_key.empty()) {
    std::lock_guard<std::mutex> lock(g_blocks_mutex);
    auto it = g_blocks.find(h->shared_block_key);
    if (it != g_blocks.end() && --it->second.refs == 0) {
      (void)hipFree(it->second.ptr);
      g_blocks.erase(it);
    }
  }
  for (hipEvent_t ev : h->events)
    if (ev) (void)hipEventDestroy(ev);
  delete h;
}

static int upload(crafter_handle* h, const void* src, size_t bytes, const void** dst) {
  void* d = nullptr;
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMalloc(&d, bytes);
  if (e != hipSuccess) return hip_fail(h, "hipMalloc(tables)", e);
  h->owned.push_back(d);
  if (src) {
    e = hipMemcpy(d, src, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_fail(h, "hipMemcpy(tables)", e);
  }
  *dst = d;
  return 0;
}

int crafter_upload_tables(crafter_handle* h, const crafter_host_tables* t) {
  if (!h || !t) return fail(h, "crafter_upload_tables: null argument");
  if (h->have_tables) return fail(h, "crafter_upload_tables: tables already uploaded");
  const Config& c = h->cfg;
  if (t->n_tex_tile != TEX_COUNT || t->n_tex_icon != MAX_ITEMS || t->n_tex_digit != 11 ||
      t->n_tex_alpha != TEX_COUNT + MAX_ITEMS + 11 || t->n_item_pos != MAX_ITEMS * 4 || t->n_unit255 != 256)
    return fail(h, "crafter_upload_tables: table sizes do not match types.hpp");
  if (t->n_daylight != c.n_daylight) return fail(h, "crafter_upload_tables: daylight table size != cfg.n_daylight");
  if (t->n_daylight >= (1 << 24)) return fail(h, "crafter_upload_tables: more than 2^24 - 1 steps per episode");   // (the step kernels keep the step counter in 24 bits of an LDS word)
  if (t->n_vignette != c.local_gw * c.unit_x * c.local_gh * c.unit_y)
    return fail(h, "crafter_upload_tables: vignette size != LocalView canvas");
  const Rules* r = t->rules;
  if (r->n_items > MAX_ITEMS || r->n_achievements > MAX_ACH || r->n_actions > MAX_ACTIONS ||
      r->n_materials >= MAX_MATERIALS)
    return fail(h, "crafter_upload_tables: rule tables exceed compiled limits");
  if (r->n_materials + 1 > kTileRows)
    return fail(h, "crafter_upload_tables: the renderer's row table holds " + std::to_string(kTileRows - 1) + " materials (render.hpp kTileRows)");
  TablePtrs& tb = h->tb;
  if (upload(h, t->rules, sizeof(Rules), (const void**)&tb.rules)) return 1;
  h->default_rules = memcmp(t->rules, &kDefaultRules, sizeof(Rules)) == 0;
  if (upload(h, t->atlas, t->atlas_bytes, (const void**)&tb.atlas)) return 1;
  if (upload(h, t->tex_tile, sizeof(int32_t) * t->n_tex_tile, (const void**)&tb.tex_tile)) return 1;
  if (upload(h, t->tex_icon, sizeof(int32_t) * t->n_tex_icon, (const void**)&tb.tex_icon)) return 1;
  if (upload(h, t->tex_digit, sizeof(int32_t) * t->n_tex_digit, (const void**)&tb.tex_digit)) return 1;
  if (upload(h, t->tex_alpha, t->n_tex_alpha, (const void**)&tb.tex_alpha)) return 1;
  if (upload(h, t->item_pos, sizeof(int32_t) * t->n_item_pos, (const void**)&tb.item_pos)) return 1;
  if (upload(h, t->daylight, sizeof(double) * t->n_daylight, (const void**)&tb.daylight)) return 1;
  if (upload(h, t->vignette, sizeof(double) * t->n_vignette, (const void**)&tb.vignette)) return 1;
  if (upload(h, t->unit255, sizeof(float) * t->n_unit255, (const void**)&tb.unit255)) return 1;
  {   // the renderer's derived tables, built once on the device (TablePtrs.render_static), shared between equal handles
    int dev = 0;
    (void)hipGetDevice(&dev);
    uint64_t hash = 14695981039346656037ull;
    const int32_t geo[] = {c.unit_x, c.unit_y, c.local_gw, c.local_gh, c.item_gw, c.item_gh, c.size_w, c.size_h, c.icon_w, c.icon_h,
                           c.digit_w, c.digit_h, c.n_daylight, r->n_items, r->n_materials};
    hash = fnv1a(hash, geo, sizeof(geo));
    hash = fnv1a(hash, t->atlas, t->atlas_bytes);
    hash = fnv1a(hash, t->tex_tile, sizeof(int32_t) * t->n_tex_tile);
    hash = fnv1a(hash, t->tex_icon, sizeof(int32_t) * t->n_tex_icon);
    hash = fnv1a(hash, t->tex_digit, sizeof(int32_t) * t->n_tex_digit);
    hash = fnv1a(hash, t->tex_alpha, t->n_tex_alpha);
    hash = fnv1a(hash, t->item_pos, sizeof(int32_t) * t->n_item_pos);
    hash = fnv1a(hash, t->daylight, sizeof(double) * t->n_daylight);
    hash = fnv1a(hash, t->vignette, sizeof(double) * t->n_vignette);
    hash = fnv1a(hash, t->unit255, sizeof(float) * t->n_unit255);
    uint64_t hash2 = 0x243f6a8885a308d3ull;
    hash2 = mix64(hash2, geo, sizeof(geo));
    hash2 = mix64(hash2, t->atlas, t->atlas_bytes);
    hash2 = mix64(hash2, t->tex_tile, sizeof(int32_t) * t->n_tex_tile);
    hash2 = mix64(hash2, t->tex_icon, sizeof(int32_t) * t->n_tex_icon);
    hash2 = mix64(hash2, t->tex_digit, sizeof(int32_t) * t->n_tex_digit);
    hash2 = mix64(hash2, t->tex_alpha, t->n_tex_alpha);
    hash2 = mix64(hash2, t->item_pos, sizeof(int32_t) * t->n_item_pos);
    hash2 = mix64(hash2, t->daylight, sizeof(double) * t->n_daylight);
    hash2 = mix64(hash2, t->vignette, sizeof(double) * t->n_vignette);
    hash2 = mix64(hash2, t->unit255, sizeof(float) * t->n_unit255);
    size_t bytes = (size_t)render_static_total_bytes(c);
    std::string key = std::to_string(dev) + "|" + std::to_string(hash) + "|" + std::to_string(hash2) + "|" + std::to_string(bytes);
    std::lock_guard<std::mutex> lock(g_blocks_mutex);
    auto it = g_blocks.find(key);
    if (it == g_blocks.end()) {
      void* blk = nullptr;
      hipError_t e = hipMalloc(&blk, bytes);
      if (e != hipSuccess) return hip_fail(h, "crafter_upload_tables: hipMalloc", e);
      hipLaunchKernelGGL(crafter_init_tables_kernel, dim3(1), dim3(kStepThreads), 0, 0, h->cfg, tb, (uint8_t*)blk);
      if (render_lit_sprite_steps(c) > 0)
        hipLaunchKernelGGL(crafter_init_lit_sprites_kernel, dim3(2 * render_lit_sprite_steps(c)), dim3(kStepThreads), 0, 0, h->cfg, tb, (uint8_t*)blk);
      e = hipDeviceSynchronize();
      if (e != hipSuccess) {
        (void)hipFree(blk);
        return hip_fail(h, "crafter_upload_tables: static render block", e);
      }
      it = g_blocks.emplace(key, SharedBlock{blk, 0}).first;
    }
    it->second.refs++;
    h->shared_block_key = key;
    tb.render_static = (const uint8_t*)it->second.ptr;
  }
  // The fused instances that run the compiled-in rules stage none: their layout is 280 bytes shorter (lds_layout with_rules
  // false) -- 26,872 B, which lets a SIXTH workgroup onto a CU (round 5; the step kernel's 61 VGPRs allow seven).  Same-box
  // A/B of the closed loop, CRAFTER_LDS_PAD=280 (five) against 0 (six): step kernel 60.9 -> 58.6 us, env-steps/s 62.07 -> 62.23 M
  // at 4096 envs, equal at 1024, no inline regeneration either way (profiles/r5_closed_occupancy_ab.txt): the kernel is
  // shorter, the world pool's kernels get their turn in the gaps instead of beside it.
  if (is_default_geometry(c) && h->default_rules)
    h->step_lds_bytes = lds_layout(c, 1, false, false).total + h->lds_pad;
  h->have_tables = true;
  return 0;
}

// Episodes without a time limit (Env(length=None), env.py:29,103): the table of _update_time's values (env.py:135-139) is
// finite and the host evaluates it; this hands over a longer one.  Launches already in their queues keep the table (and
// the size) they were launched with, so nothing is waited for; the old table is freed with the handle.
int crafter_extend_daylight(crafter_handle* h, const double* daylight, int32_t n) {
  if (!h || !daylight) return fail(h, "crafter_extend_daylight: null argument");
  if (!h->have_tables) return fail(h, "crafter_extend_daylight: no tables uploaded yet");
  if (n <= h->cfg.n_daylight) return fail(h, "crafter_extend_daylight: the new table must be longer than the current one");
  if (n >= (1 << 24)) return fail(h, "crafter_extend_daylight: more than 2^24 - 1 steps per episode");
  if (h->cfg.n_daylight < kLitSteps)   // (the renderer's static block is laid out for min(n_daylight, kLitSteps) lit steps)
    return fail(h, "crafter_extend_daylight: a handle created with fewer than " + std::to_string(kLitSteps) + " daylight steps cannot grow");
  const void* d = nullptr;
  if (upload(h, daylight, sizeof(double) * (size_t)n, &d)) return 1;
  h->tb.daylight = (const double*)d;
  h->cfg.n_daylight = n;
  return 0;
}

int crafter_bind_state(crafter_handle* h, const crafter_state_ptrs* state) {
  if (!h || !state) return fail(h, "crafter_bind_state: null argument");
  const StatePtrs& s = *state;
  if (!s.mat || !s.objmap || !s.objs || !s.mt || !s.rec || !s.chunk_order || !s.chunk_seen)
    return fail(h, "crafter_bind_state: null state buffer");
  if (h->cfg.want_semantic && !s.semantic) return fail(h, "crafter_bind_state: want_semantic without a buffer");
  if (h->cfg.auto_reset && !s.reset_q) return fail(h, "crafter_bind_state: auto_reset without a reset queue");
  if (h->pool && (!s.pool_mat || !s.pool_objs || !s.pool_mt || !s.pool_hdr || !s.pool_chunk_order || !s.gen_q || !s.gen_latest || !s.pool_perm || !s.pool_census))
    return fail(h, "crafter_bind_state: world pool enabled but pool buffers missing");
  uintptr_t bits = (uintptr_t)s.mat | (uintptr_t)s.objmap | (uintptr_t)s.objs | (uintptr_t)s.mt | (uintptr_t)s.rec;
  if (bits & 15) return fail(h, "crafter_bind_state: state buffers must be 16-byte aligned");
  h->st = s;
  h->have_state = true;
  return 0;
}

int32_t crafter_lds_bytes(const crafter_handle* h) { return h ? h->step_lds_bytes : -1; }

int32_t crafter_slot_map_derived(const crafter_handle* h) { return h ? lds_layout(h->cfg).maps_in_lds : -1; }

int32_t crafter_step_instance(const crafter_handle* h) {
  if (!h) return -1;
  int lm = lds_layout(h->cfg).maps_in_lds ? 1 : 0, geo = is_default_geometry(h->cfg) ? 1 : 0;
  int rul = (geo && h->have_tables && h->default_rules) ? 1 : 0;
  if (!lm && is_default_view(h->cfg) && h->have_tables && h->default_rules) return 8 + 1;   // crafter_step_kernel<0, 2, 1>
  return lm * 4 + geo * 2 + rul;
}

static int ready(crafter_handle* h, const char* who) {
  if (!h) return fail(nullptr, std::string(who) + ": null handle");
  if (!h->have_tables) return fail(h, std::string(who) + ": crafter_upload_tables not called");
  if (!h->have_state) return fail(h, std::string(who) + ": crafter_bind_state not called");
  return 0;
}

// World-pool scheduler, once per crafter_step (after the step's kernels are enqueued).  Any HIP failure here
// disables the pool for good (pool_failed): worlds of batches <= safe_seq stay valid, nothing newer is ever
// trusted, finished envs regenerate inline (gen_parity = -1 in StepCtl) -- slower, never wrong.
static void pool_fail(crafter_handle* h, const char* what, hipError_t e) {
  h->pool_failed = true;
  h->pool_err = std::string("world pool disabled (") + what + ": " + hipGetErrorString(e) + "); finished envs regenerate inline";
}

// Trust in batches polled_seq + 1 .. safe_seq rests on hipStreamWaitEvent calls enqueued on ONE stream (safe_stream).  The
// header only asks the caller to serialise the calls on a handle, not to keep to one stream: a call that arrives on
// another stream gets the same waits before any of its kernels is enqueued (a reset() under
// torch.cuda.stream(s) followed by step() on another stream trusted batches that stream had never waited for).
static void pool_adopt_stream(crafter_handle* h, hipStream_t stream) {
  if (!h->pool || h->pool_failed) return;
  if (h->have_safe_stream && h->safe_stream == stream) return;
  if (h->have_safe_stream)
    for (uint32_t s = (h->safe_seq > h->polled_seq + kGenRing ? h->safe_seq - kGenRing : h->polled_seq) + 1; s <= h->safe_seq; s++) {
      // (an event slot older than the ring has been recorded again by a later batch of the same side stream: waiting for
      // that one is waiting for more)
      hipError_t ew = hipStreamWaitEvent(stream, h->ev_gen[s % kGenRing], 0);
      if (ew != hipSuccess) {
        h->pool_failed = true;
        h->pool_err = std::string("world pool disabled (hipStreamWaitEvent(new launch stream): ") + hipGetErrorString(ew) + "); finished envs regenerate inline";
        h->safe_seq = h->polled_seq;
        break;
      }
    }
  h->safe_stream = stream;
  h->have_safe_stream = true;
}

// Every entry point that enqueues kernels takes the caller's stream.  The header promises that a caller who changes streams
// between calls need not order them himself: the new stream waits for whatever the handle still has in flight on the
// previous one (an event recorded there) -- whether or not the world pool runs (crafter_reset on stream A followed
// by crafter_step on stream B raced the reset kernel) -- and then for the pool batches the old stream had been ordered behind.
static int adopt_stream(crafter_handle* h, hipStream_t stream) {
  if (h->have_last_stream && h->last_stream != stream) {
    if (!h->ev_switch) {
      hipError_t ec = hipEventCreateWithFlags(&h->ev_switch, hipEventDisableTiming);
      if (ec != hipSuccess) return hip_fail(h, "stream change: hipEventCreate", ec);
    }
    hipError_t e = hipEventRecord(h->ev_switch, h->last_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream, h->ev_switch, 0);
    if (e != hipSuccess) {
      // The previous stream may be gone (a C caller may destroy a stream once its work is done -- ADVICE r5: the failed record
      // used to leave last_stream pointing at it, and every later call failed the same way).  Everything the handle enqueued
      // there is then ordered the blunt way, once, and the handle moves on to the new stream.
      (void)hipGetLastError();
      hipError_t es = hipDeviceSynchronize();
      if (es != hipSuccess) return hip_fail(h, "stream change: ordering the new stream behind the previous one", es);
    }
  }
  h->last_stream = stream;
  h->have_last_stream = true;
  pool_adopt_stream(h, stream);
  return 0;
}

// main_recorded: ev_main already marks the point of the launch stream the batch has to wait for (crafter_step_n: it is the
// stop event of the stretch's last kernel -- that kernel's own completion signal instead of a marker packet of its own
// behind it: every packet between two dependent kernels costs the launch stream 4-5 us here, profiles/r5_rollout_gaps.txt)
static void pool_schedule(crafter_handle* h, hipStream_t main, int steps = 1, bool main_recorded = false, bool behind_rollout = false) {
  // 1. trust: batches complete in launch order per stream but the two streams interleave, so poll in sequence order
  while (h->polled_seq < h->batches) {
    hipError_t q = hipEventQuery(h->ev_gen[(h->polled_seq + 1) % kGenRing]);
    if (q == hipSuccess) {
      h->polled_seq++;
      if (h->safe_seq < h->polled_seq) h->safe_seq = h->polled_seq;
    } else if (q == hipErrorNotReady) {
      break;
    } else {
      return pool_fail(h, "hipEventQuery", q);
    }
  }
  h->steps_since_gen += steps;
  if (h->steps_since_gen < h->gen_period) return;
  // 2. back-pressure: order the launch stream behind batch seq - kGenLag (no host wait).  This also covers the reuse
  //    of queue segment / event slot seq % kGenRing, last used by batch seq - kGenRing + 1 <= seq - kGenLag.
  uint32_t seq = h->batches + 1;
  if (seq > (uint32_t)h->gen_lag + 1 && h->safe_seq < seq - h->gen_lag) {
    uint32_t t = seq - h->gen_lag;
    hipError_t ew = hipStreamWaitEvent(main, h->ev_gen[t % kGenRing], 0);
    if (ew != hipSuccess) return pool_fail(h, "hipStreamWaitEvent(launch stream)", ew);
    h->safe_seq = t;   // steps enqueued from now on run after batch t
  }
  // 3. launch batch `seq` over the segment that has been collecting
  hipStream_t side = h->side[seq % kGenStreams];
  hipError_t e = main_recorded ? hipSuccess : hipEventRecord(h->ev_main, main);
  if (e != hipSuccess) return pool_fail(h, "hipEventRecord(launch stream)", e);
  e = hipStreamWaitEvent(side, h->ev_main, 0);
  if (e != hipSuccess) return pool_fail(h, "hipStreamWaitEvent(generation stream)", e);
  // (Consecutive batches run side by side on the two streams.  They never write the same pool entry: a request whose
  // entry still has an older generation on its way is put off on the device, request_generation / PoolHdr.pending.)
  int seg = h->gen_parity;
  int n = h->cfg.num_envs;
  // Wave priority of the batch's two one-wave-per-world kernels (seeding, ordered draws: a few hundred waves whose serial
  // chain is the batch's latency; the classification in between stays at 0).  Behind a rollout stretch 2: the launch
  // stream of an open loop runs far ahead of the host's polling, every stretch waits for batch seq - 3 on the device, and
  // on every second stretch that wait was a real one (profiles/r5_rollout_gaps.txt): open loop 74.5-74.9 -> 75.6-75.9 M.
  // Behind a closed-loop step 1 (round 6; 0 until then: priority 2 costs the 64x64 loop 0.5 %, 64.7 -> 64.4 M): the ordered draws of
  // a 256x256 world are a 7 ms chain, and BASELINE configs[3] -- bound by its generator -- runs at 15.05-15.12 M env-steps/s with it
  // against 14.25-14.59 M (profiles/r6_cfg4_prio.txt; 2 and 3: the same); the 64x64 loop's 20-step window gains 1.5 %, its
  // steady state loses 0.2 % (profiles/r6_headline_prio.txt).
  const int prio = h->gen_serial_prio >= 0 ? h->gen_serial_prio : (behind_rollout ? 2 : 1);
  const int cgrid = behind_rollout ? h->classify_grid : h->classify_grid_step;
  dim3 gs(n < kGenSerialGrid ? n : kGenSerialGrid), gc((long long)n * gen_classify_parts(h->cfg) < cgrid ? n * gen_classify_parts(h->cfg) : cgrid);
#ifdef CRAFTER_PROBES
  if (h->probe_free_gen) {   // timing probe: see crafter_gen_stamp_kernel
    if (h->probe_free_gen == 2) {   // ... behind workgroups that occupy what the three kernels occupy, for as long (profiles/r5_kernel_stats.csv)
      const int worlds = h->probe_worlds > 0 ? h->probe_worlds : 390;
      int ws = worlds < (int)gs.x ? worlds : (int)gs.x, wc = worlds * gen_classify_parts(h->cfg) < (int)gc.x ? worlds * gen_classify_parts(h->cfg) : (int)gc.x;
      const int l0 = h->probe_lds[0] >= 0 ? h->probe_lds[0] : kGenSeedLds, l1 = h->probe_lds[1] >= 0 ? h->probe_lds[1] : gen_classify_lds_bytes(h->cfg),
                l2 = h->probe_lds[2] >= 0 ? h->probe_lds[2] : h->gen_resolve_lds_bytes;
      if (h->probe_big) {
        hipLaunchKernelGGL(crafter_gen_occupy_kernel<0>, dim3(ws), dim3(kGenSeedThreads), l0, side, h->probe_us[0], worlds, (uint32_t*)nullptr);
        hipLaunchKernelGGL(crafter_gen_occupy_kernel<1>, dim3(wc), dim3(kGenClassifyThreads), l1, side, h->probe_us[1], worlds * gen_classify_parts(h->cfg), (uint32_t*)nullptr);
        hipLaunchKernelGGL(crafter_gen_occupy_kernel<1>, dim3(ws), dim3(kGenResolveThreads), l2, side, h->probe_us[2], worlds, (uint32_t*)nullptr);
      } else {
        hipLaunchKernelGGL(crafter_gen_occupy_kernel<0>, dim3(ws), dim3(kGenSeedThreads), l0, side, h->probe_us[0], worlds, (uint32_t*)nullptr);
        hipLaunchKernelGGL(crafter_gen_occupy_kernel<0>, dim3(wc), dim3(kGenClassifyThreads), l1, side, h->probe_us[1], worlds * gen_classify_parts(h->cfg), (uint32_t*)nullptr);
        hipLaunchKernelGGL(crafter_gen_occupy_kernel<0>, dim3(ws), dim3(kGenResolveThreads), l2, side, h->probe_us[2], worlds, (uint32_t*)nullptr);
      }
    }
    if (is_default_geometry(h->cfg))
      hipLaunchKernelGGL(crafter_gen_stamp_kernel<1>, dim3(64), dim3(kStepThreads), 0, side, h->cfg, h->tb, h->st, seg, seq);
    else
      hipLaunchKernelGGL(crafter_gen_stamp_kernel<0>, dim3(64), dim3(kStepThreads), 0, side, h->cfg, h->tb, h->st, seg, seq);
  } else
#endif
  if (is_default_geometry(h->cfg)) {
    hipLaunchKernelGGL(crafter_gen_seed_kernel<1>, gs, dim3(kGenSeedThreads), kGenSeedLds, side, h->cfg, h->tb, h->st, seg, prio);
    hipLaunchKernelGGL(crafter_gen_classify_kernel<1>, gc, dim3(kGenClassifyThreads), gen_classify_lds_bytes(h->cfg), side, h->cfg, h->tb, h->st, seg, h->gen_classify_prio);
    hipLaunchKernelGGL(crafter_gen_resolve_kernel<1>, gs, dim3(kGenResolveThreads), h->gen_resolve_lds_bytes, side, h->cfg, h->tb,
                       h->st, seg, seq, prio);
  } else {
    hipLaunchKernelGGL(crafter_gen_seed_kernel<0>, gs, dim3(kGenSeedThreads), kGenSeedLds, side, h->cfg, h->tb, h->st, seg, prio);
    hipLaunchKernelGGL(crafter_gen_classify_kernel<0>, gc, dim3(kGenClassifyThreads), gen_classify_lds_bytes(h->cfg), side, h->cfg, h->tb, h->st, seg, h->gen_classify_prio);
    hipLaunchKernelGGL(crafter_gen_resolve_kernel<0>, gs, dim3(kGenResolveThreads), h->gen_resolve_lds_bytes, side, h->cfg, h->tb,
                       h->st, seg, seq, prio);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return pool_fail(h, "generation kernel launch", e);
  // from here on the batch exists: the sequence number is consumed even if a later call fails (its event would then
  // never be trusted, because pool_failed stops the polling)
  h->batches = seq;
  h->steps_since_gen = 0;
  h->gen_parity = (seg + 1) % kGenRing;
  e = hipMemsetAsync(h->st.gen_q + (size_t)seg * gen_q_stride(h->cfg), 0, 16, side);
  if (e != hipSuccess) return pool_fail(h, "hipMemsetAsync(request segment)", e);
  e = hipEventRecord(h->ev_gen[seq % kGenRing], side);
  if (e != hipSuccess) return pool_fail(h, "hipEventRecord(generation stream)", e);
}

int crafter_reset(crafter_handle* h, const uint8_t* mask, uint8_t* obs, void* stream) {
  if (ready(h, "crafter_reset")) return 1;
  // The reset kernel generates the next episode's world into the env's pool entry itself (gen_one): it must not share
  // that entry with a generation batch still in flight on a side stream (an env reset in mid-episode k may have world
  // k + 2 in such a batch, and k + 2 lives in the entry the kernel is about to write: ADVICE r2).  So the launch stream
  // is ordered behind every launched batch first -- no host wait -- and all of them are trusted from here on.
  if (adopt_stream(h, (hipStream_t)stream)) return 1;
  if (h->pool && !h->pool_failed) {
    for (uint32_t s = h->safe_seq + 1; s <= h->batches; s++) {
      hipError_t ew = hipStreamWaitEvent((hipStream_t)stream, h->ev_gen[s % kGenRing], 0);
      if (ew != hipSuccess) {
        pool_fail(h, "hipStreamWaitEvent(reset)", ew);
        break;
      }
    }
    if (!h->pool_failed) h->safe_seq = h->batches;
  }
  hipLaunchKernelGGL(crafter_reset_kernel, dim3(h->cfg.num_envs), dim3(kResetThreads), h->reset_lds_bytes,
                     (hipStream_t)stream, h->cfg, h->tb, h->st, mask, (h->pool && !h->pool_failed) ? h->gen_parity : -1, obs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(h, "crafter_reset launch", e);
  return 0;
}

// A launch with start / stop events attached (timing mode: hipExtLaunchKernelGGL) or a plain one -- the plain launch is
// the cheaper call on the host, which is what bounds small batches (tools/host_overhead.py).
#define CRAFTER_LAUNCH(kernel, grid, block, lds, stream, start, stop, ...)                                          \
  do {                                                                                                              \
    if ((start) != nullptr || (stop) != nullptr)                                                                    \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, start, stop, 0, __VA_ARGS__);                         \
    else                                                                                                            \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                            \
  } while (0)

static int requeue_grid(const crafter_handle* h, const StepCtl& ctl) {
  // With the world pool running the queue is all but always empty (0 of 68,684 resets in the benchmark): a handful of
  // workgroups finds that out faster than 256 (each needs a slot next to the resident generation workgroups).  Without
  // the pool every reset comes through here.
  int full = h->cfg.num_envs < kRequeueGrid ? h->cfg.num_envs : kRequeueGrid;
  return (ctl.gen_parity >= 0 && full > h->requeue_grid) ? h->requeue_grid : full;
}
static void launch_requeue(crafter_handle* h, const StepCtl& ctl, uint8_t* obs, hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
  CRAFTER_LAUNCH(crafter_requeue_reset_kernel, dim3(requeue_grid(h, ctl)), dim3(kRequeueThreads), h->reset_lds_bytes, stream, start, stop,
                        h->cfg, h->tb, h->st, ctl.parity, ctl.gen_parity, obs);
}

static int need_noise_raw(crafter_handle* h, const char* who) {
  if (h->noise_raw || !h->noise_ahead) return 0;
  hipError_t ea = hipMalloc((void**)&h->noise_raw, (size_t)h->cfg.num_envs * kNoiseStates * MT_N * sizeof(uint32_t));
  if (ea != hipSuccess) return hip_fail(h, who, ea);
  h->owned.push_back(h->noise_raw);
  return 0;
}

// the scratch a night frame's pixels wait in when the kernel's layout keeps no buffer for them in LDS (split / pipelined
// step: the frame halves; big_layout: the step kernel of large worlds), allocated on first use
static int need_night_px(crafter_handle* h, const char* who) {
  if (h->night_px) return 0;
  hipError_t ea = hipMalloc((void**)&h->night_px, (size_t)h->cfg.num_envs * frame_night_px_words(h->cfg) * 4);
  if (ea != hipSuccess) return hip_fail(h, who, ea);
  h->owned.push_back(h->night_px);
  return 0;
}

int crafter_step(crafter_handle* h, const int32_t* actions, uint8_t* obs, float* reward, uint8_t* done,
                 void* stream) {
  if (ready(h, "crafter_step")) return 1;
  if (!actions || !reward || !done) return fail(h, "crafter_step: null argument");
  if (adopt_stream(h, (hipStream_t)stream)) return 1;
  StepCtl ctl;
  ctl.gen_parity = (h->pool && !h->pool_failed) ? h->gen_parity : -1;
  ctl.safe_seq = h->safe_seq;
  // timing mode: start / stop events attached to the kernels themselves (hipExtLaunchKernelGGL), i.e. the
  // execution time a profiler reports, without the dispatch latency a hipEventRecord bracket would include
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  dim3 grid_n(h->cfg.num_envs), block_s(kStepThreads);
  bool frames = h->cfg.render_obs != 0 && obs != nullptr;
  bool split = h->split < 0 ? !frames : h->split != 0;
  bool requeue = h->cfg.auto_reset != 0;
  bool pair = is_default_geometry(h->cfg) && h->default_rules && split;   // rules kernel (+ frame kernel) instead of the fused step kernel
  bool ordered = h->order && !pair;
  ctl.parity = (int)(h->steps++ & 1);   // (the reset_q halves alternate over the launches that use them)
  ctl.early_frame = h->early_frame < 0 ? (h->cfg.num_envs >= kEarlyMinEnvs ? 1 : 0) : h->early_frame;
  if (h->timing)
    for (int i = 0; i < 4; i++) {
      hipError_t ee = hipEventCreate(&ev[i]);
      if (ee != hipSuccess) return hip_fail(h, "crafter_step: hipEventCreate (timing mode)", ee);
    }
  if (frames && !pair) {   // a fused step kernel draws: its night frames take their noise from states generated ahead
    if (need_noise_raw(h, "crafter_step: noise scratch")) return 1;
    ctl.noise_raw = h->noise_raw;
  }
  if (ordered) {
    uint64_t k = h->ordered_launches++;
    ctl.order = k > 0 ? h->order + (size_t)(k & 1) * h->cfg.num_envs : nullptr;
    if (h->order_override) ctl.order = h->order_override;
    ctl.order_build = h->order + (size_t)((k + 1) & 1) * h->cfg.num_envs;
    ctl.next_step = h->next_step;
    grid_n = dim3(h->cfg.num_envs + 1);   // block 0 builds the next launch's order
  }
  // The regeneration kernel (envs that finished and found no world in the pool: all but never any) only has to sit between
  // the rules of this step and the rules of the next.  In the split step it runs BESIDE the frame kernel, on the handle's own
  // stream -- the envs it regenerates and draws are exactly those the frame kernel skips -- so its launch and the look at
  // the (empty) queue cost the launch stream nothing.
  bool beside = false;
  if (pair) {   // split step: rules at wave granularity, then the frames
    if (frames && need_night_px(h, "crafter_step: frame kernel scratch")) return 1;
    CRAFTER_LAUNCH(crafter_rules_kernel, grid_n, dim3(kRulesThreads), lane_layout(h->cfg).total, (hipStream_t)stream, ev[0],
                          frames ? nullptr : ev[1], h->cfg, h->tb, h->st, actions, obs, reward, done, ctl);
    if (frames && requeue && h->aux) {
      hipError_t ea = hipEventRecord(h->ev_rules, (hipStream_t)stream);
      if (ea == hipSuccess) ea = hipStreamWaitEvent(h->aux, h->ev_rules, 0);
      if (ea != hipSuccess) return hip_fail(h, "crafter_step: fork to the regeneration stream", ea);
      beside = true;
      launch_requeue(h, ctl, obs, h->aux, ev[2], ev[3]);
      ea = hipEventRecord(h->ev_requeue, h->aux);
      if (ea != hipSuccess) return hip_fail(h, "crafter_step: hipEventRecord(regeneration stream)", ea);
    }
    if (frames)
      CRAFTER_LAUNCH(crafter_frame_kernel, grid_n, block_s, frame_layout(h->cfg).total, (hipStream_t)stream, nullptr, ev[1],
                            h->cfg, h->tb, h->st, obs, h->night_px);
    if (beside) {
      hipError_t ea = hipStreamWaitEvent((hipStream_t)stream, h->ev_requeue, 0);
      if (ea != hipSuccess) return hip_fail(h, "crafter_step: join of the regeneration stream", ea);
    }
  } else if (is_default_geometry(h->cfg) && h->default_rules && !ordered && frames &&
             (h->wide < 0 ? h->cfg.num_envs <= kWideMaxEnvs : h->wide != 0)) {   // few envs: eight waves per env
    CRAFTER_LAUNCH(crafter_step_wide_kernel, grid_n, dim3(kWideThreads), h->step_lds_bytes, (hipStream_t)stream, ev[0], ev[1],
                          h->cfg, h->tb, h->st, actions, obs, reward, done, ctl);
  } else if (is_default_geometry(h->cfg) && h->default_rules && frames && ctl.early_frame) {   // ... in batches larger than the chip holds at once
    CRAFTER_LAUNCH(crafter_step_early_kernel, grid_n, block_s, h->step_lds_bytes, (hipStream_t)stream, ev[0], ev[1],
                          h->cfg, h->tb, h->st, actions, obs, reward, done, ctl);
  } else if (is_default_geometry(h->cfg) && h->default_rules) {   // crafter.Env() as everybody runs it
    CRAFTER_LAUNCH((crafter_step_kernel<1, 1, 1>), grid_n, block_s, h->step_lds_bytes, (hipStream_t)stream, ev[0], ev[1],
                          h->cfg, h->tb, h->st, actions, obs, reward, done, ctl);
  } else if (is_default_geometry(h->cfg)) {                 // implies LDS-resident maps
    CRAFTER_LAUNCH((crafter_step_kernel<1, 1, 0>), grid_n, block_s, h->step_lds_bytes, (hipStream_t)stream, ev[0], ev[1],
                          h->cfg, h->tb, h->st, actions, obs, reward, done, ctl);
  } else if (lds_layout(h->cfg).maps_in_lds) {
    CRAFTER_LAUNCH((crafter_step_kernel<1, 0, 0>), grid_n, block_s, h->lds_bytes, (hipStream_t)stream, ev[0], ev[1],
                          h->cfg, h->tb, h->st, actions, obs, reward, done, ctl);
  } else {
    if (frames && need_night_px(h, "crafter_step: night frame scratch")) return 1;
    ctl.night_px = h->night_px;
    if (is_default_view(h->cfg) && h->default_rules)   // BASELINE configs[3]: crafter.Env(area=(256, 256)) -- the default view and rules compiled in
      CRAFTER_LAUNCH((crafter_step_kernel<0, 2, 1>), grid_n, block_s, h->step_lds_bytes, (hipStream_t)stream, ev[0], ev[1],
                            h->cfg, h->tb, h->st, actions, obs, reward, done, ctl);
    else
      CRAFTER_LAUNCH((crafter_step_kernel<0, 0, 0>), grid_n, block_s, h->step_lds_bytes, (hipStream_t)stream, ev[0], ev[1],
                            h->cfg, h->tb, h->st, actions, obs, reward, done, ctl);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(h, "crafter_step launch", e);
  if (requeue && !beside) launch_requeue(h, ctl, obs, (hipStream_t)stream, ev[2], ev[3]);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(h, "crafter_step (auto-reset) launch", e);
  if (h->timing)
    for (int i = 0; i < 4; i++) h->events.push_back(ev[i]);
  if (h->pool && !h->pool_failed) pool_schedule(h, (hipStream_t)stream);
  return 0;
}

int crafter_set_timing(crafter_handle* h, int enable) {
  if (!h) return 1;
  h->timing = enable != 0;
  if (!h->timing && !h->events.empty()) {   // a caller that switches timing off without reading it: nothing is kept
    for (hipEvent_t ev : h->events) {
      if (!ev) continue;
      (void)hipEventSynchronize(ev);
      (void)hipEventDestroy(ev);
    }
    h->events.clear();
  }
  return 0;
}

int crafter_get_timing(crafter_handle* h, double* step_ms, double* reset_ms, int32_t* launches) {
  if (!h || !step_ms || !reset_ms || !launches) return fail(h, "crafter_get_timing: null argument");
  double a = 0, b = 0;
  int n = (int)h->events.size() / 4;
  for (int i = 0; i < n; i++) {
    hipEvent_t* ev = &h->events[4 * i];
    hipError_t e = hipEventSynchronize(ev[1]);
    if (e != hipSuccess) return hip_fail(h, "hipEventSynchronize", e);
    float x = 0, y = 0;
    (void)hipEventElapsedTime(&x, ev[0], ev[1]);
    a += x;
    if (h->cfg.auto_reset && ev[2] && ev[3] && hipEventSynchronize(ev[3]) == hipSuccess && hipEventElapsedTime(&y, ev[2], ev[3]) == hipSuccess) b += y;
    for (int k = 0; k < 4; k++)
      if (ev[k]) (void)hipEventDestroy(ev[k]);
  }
  (void)hipGetLastError();   // (an event pair that was never recorded -- a launch that failed half-way -- must not leave its error for the caller's next HIP call)
  h->events.clear();
  *step_ms = a;
  *reset_ms = b;
  *launches = n;
  return 0;
}

// T consecutive steps of every env in one launch per stretch between two generation batches (rollout_body): the open-loop
// form of crafter_step for policies that do not look at the observations (random, scripted, action repeat).  Outputs of
// step t at obs + t * num_envs * obs_bytes, reward + t * num_envs, done + t * num_envs; bit-identical to T calls of
// crafter_step with the same actions.
int crafter_step_n(crafter_handle* h, int32_t steps, const int32_t* actions, uint8_t* obs, float* reward, uint8_t* done,
                   void* stream) {
  if (ready(h, "crafter_step_n")) return 1;
  if (!actions || !reward || !done || steps < 1) return fail(h, "crafter_step_n: bad argument");
  if (adopt_stream(h, (hipStream_t)stream)) return 1;
  bool pooled = h->pool && !h->pool_failed;
  if (!h->stalled_at) {
    hipError_t ea = hipMalloc((void**)&h->stalled_at, (size_t)h->cfg.num_envs * sizeof(int32_t));
    if (ea != hipSuccess) return hip_fail(h, "crafter_step_n: hipMalloc", ea);
    h->owned.push_back(h->stalled_at);
  }
  const size_t n = (size_t)h->cfg.num_envs;
  const size_t obs_stride = n * (size_t)h->cfg.size_w * h->cfg.size_h * 3;
  dim3 grid_n(h->cfg.num_envs), block_s(kStepThreads);
  bool requeue = h->cfg.auto_reset != 0;
  for (int done_steps = 0; done_steps < steps;) {
    int left = steps - done_steps;
    // a stretch never crosses a generation batch (requests keep their segment).  Without the pool every env that finishes
    // an episode stops in the rollout kernel and runs the REST of the stretch in the regeneration kernel, on a small grid,
    // with the generic step instance: stretches stay short there too (one launch over all T steps put nearly
    // all the work on that grid once T exceeded an episode).
    int room = pooled ? h->gen_period - h->steps_since_gen : (left < kUnpooledStretch ? left : kUnpooledStretch);
    int T = left < room ? left : (room > 0 ? room : 1);
    const int32_t* a = actions + (size_t)done_steps * n;
    uint8_t* o = obs ? obs + (size_t)done_steps * obs_stride : nullptr;
    float* r = reward + (size_t)done_steps * n;
    uint8_t* d = done + (size_t)done_steps * n;
    StepCtl ctl;
    ctl.parity = (int)(h->steps++ & 1);
    ctl.gen_parity = pooled ? h->gen_parity : -1;
    ctl.safe_seq = h->safe_seq;
    RolloutArgs ra;
    ra.T = T; ra.obs_stride = obs_stride; ra.stalled_at = h->stalled_at;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (h->timing)
      for (int i = 0; i < 4; i++) {
        hipError_t ee = hipEventCreate(&ev[i]);
        if (ee != hipSuccess) return hip_fail(h, "crafter_step_n: hipEventCreate (timing mode)", ee);
      }
    int instance = (is_default_geometry(h->cfg) && h->default_rules) ? 7 : is_default_geometry(h->cfg) ? 6 : lds_layout(h->cfg).maps_in_lds ? 4 : 0;
    if (instance == 0 && is_default_view(h->cfg) && h->default_rules) instance = 9;   // (as crafter_step_kernel<0, 2, 1>)
    if (o && h->cfg.render_obs) {
      if (need_noise_raw(h, "crafter_step_n: noise scratch")) return 1;
      ctl.noise_raw = h->noise_raw;
    }
    if (h->order && h->rollout_order) {   // slow envs first, as in crafter_step (one order for both kinds of launch)
      uint64_t k = h->ordered_launches++;
      ctl.order = k > 0 ? h->order + (size_t)(k & 1) * h->cfg.num_envs : nullptr;
      if (h->order_override) ctl.order = h->order_override;
      ctl.order_build = h->order + (size_t)((k + 1) & 1) * h->cfg.num_envs;
      ctl.next_step = h->next_step;
    }
    if (instance == 0 || instance == 9) {   // big_layout (as crafter_step_kernel<0, 0, 0>)
      if (o && h->cfg.render_obs && need_night_px(h, "crafter_step_n: night frame scratch")) return 1;
      ctl.night_px = h->night_px;
    }
    // (the default instance keeps no staged rules in its resident layout)
    size_t rollout_lds = instance == 7 ? (size_t)lds_layout(h->cfg, 1, false, false).total + (size_t)h->rollout_lds_pad
                         : (instance == 6 || instance == 0 || instance == 9) ? (size_t)h->step_lds_bytes : (size_t)h->lds_bytes;
    launch_rollout(instance, h->cfg.num_envs, rollout_lds, (hipStream_t)stream, ev[0], ev[1], h->cfg, h->tb, h->st,
                   a, o, r, d, ctl, ra);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(h, "crafter_step_n launch", e);
    // (a generation batch follows this stretch: its side stream waits for the stretch's last kernel through that kernel's
    // own stop event)
    const bool batch_follows = pooled && requeue && !h->timing && h->fold_main_event && h->steps_since_gen + T >= h->gen_period;
    if (requeue)
      launch_requeue_rollout(requeue_grid(h, ctl), h->lds_bytes, (hipStream_t)stream, ev[2], batch_follows ? h->ev_main : ev[3], h->cfg, h->tb,
                             h->st, a, o, r, d, ctl, ra);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(h, "crafter_step_n (auto-reset) launch", e);
    if (h->timing)
      for (int i = 0; i < 4; i++) h->events.push_back(ev[i]);
    if (pooled) pool_schedule(h, (hipStream_t)stream, T, batch_follows, true);
    pooled = h->pool && !h->pool_failed;
    done_steps += T;
  }
  return 0;
}

// Diagnostics / experiments: the next crafter_step calls dispatch the envs in this order (device int32[num_envs], a
// permutation -- not checked; the caller keeps it alive) instead of the one the library builds; NULL: back to its own.
int crafter_debug_set_dispatch_order(crafter_handle* h, const int32_t* order) {
  if (ready(h, "crafter_debug_set_dispatch_order")) return 1;
  if (!h->order) return 2;
  if (order) {   // two workgroups stepping one env would corrupt it: checked here, once (a diagnostic call: the copy synchronises)
    std::vector<int32_t> host((size_t)h->cfg.num_envs);
    hipError_t e = hipMemcpy(host.data(), order, host.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(h, "crafter_debug_set_dispatch_order", e);
    std::vector<uint8_t> seen(host.size(), 0);
    for (int32_t v : host) {
      if (v < 0 || v >= h->cfg.num_envs || seen[(size_t)v]) return fail(h, "crafter_debug_set_dispatch_order: not a permutation of the env indices");
      seen[(size_t)v] = 1;
    }
  }
  h->order_override = order;
  return 0;
}

// Diagnostics: the dispatch order the next crafter_step will use (host int32[num_envs]; synchronises the device).
// Returns 2 when this handle keeps none (see crafter_handle::order).
int crafter_debug_dispatch_order(crafter_handle* h, int32_t* out) {
  if (ready(h, "crafter_debug_dispatch_order")) return 1;
  if (!out) return fail(h, "crafter_debug_dispatch_order: null argument");
  if (!h->order || h->ordered_launches == 0) return 2;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess)
    e = hipMemcpy(out, h->order + (size_t)(h->ordered_launches & 1) * h->cfg.num_envs, (size_t)h->cfg.num_envs * sizeof(int32_t),
                  hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(h, "crafter_debug_dispatch_order", e);
  return 0;
}

int crafter_render(crafter_handle* h, const uint8_t* mask, uint8_t* out, void* stream) {
  if (ready(h, "crafter_render")) return 1;
  if (!out) return fail(h, "crafter_render: null output");
  if (adopt_stream(h, (hipStream_t)stream)) return 1;
  hipLaunchKernelGGL(crafter_render_kernel, dim3(h->cfg.num_envs), dim3(kStepThreads), h->lds_bytes,
                     (hipStream_t)stream, h->cfg, h->tb, h->st, mask, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(h, "crafter_render launch", e);
  return 0;
}

#ifdef CRAFTER_PROBES
// probe builds: the generation kernels' phase clocks (env_kernels.hpp g_gen_probe), read and cleared; synchronises the device
int crafter_debug_gen_probe(uint64_t out[32]) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gen_probe), 32 * sizeof(uint64_t)) != hipSuccess) return 1;
  uint64_t zero[32] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gen_probe), zero, sizeof(zero)) == hipSuccess ? 0 : 1;
}
#endif

int crafter_pool_status(const crafter_handle* h, uint32_t* launched, uint32_t* trusted) {
  if (!h) return -1;
  if (launched) *launched = h->batches;
  if (trusted) *trusted = h->safe_seq;
  return !h->pool ? 0 : h->pool_failed ? 2 : 1;
}

const char* crafter_pool_error(const crafter_handle* h) { return h ? h->pool_err.c_str() : ""; }

int crafter_debug_eval(int mode, const uint8_t* perm, const double* x, const double* y, const double* z, double* out,
                       int64_t n, void* stream) {
  if (mode < 0 || mode > 3 || !x || !out || n < 0 || (mode == 0 && (!perm || !y || !z)))
    return fail(nullptr, "crafter_debug_eval: bad argument");
  if (n == 0) return 0;
  long long blocks = (n + kStepThreads - 1) / kStepThreads;
  hipLaunchKernelGGL(crafter_debug_eval_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(kStepThreads), 512 + kSimplexLdsBytes, (hipStream_t)stream,
                     perm, x, y, z, out, (long long)n, mode);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(nullptr, "crafter_debug_eval launch", e);
  return 0;
}

const char* crafter_last_error(const crafter_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Multi-GPU: the per-step exchange the north star names ("RCCL gather of obs / reward / done over xGMI"), enqueued from C.
// Through torch.distributed one step of an N > 1 run costs the host 49-72 us of Python (begin + step(out=) + all_gather
// launch + result) against 26-31 us of GPU at one GPU's share of configs[2] (512 envs): host-bound before a byte has
// crossed xGMI (round 4).  Here ONE call enqueues the step kernels -- writing straight into the packed send record -- and
// the all-gather of that record, on a stream of the exchange's own behind an event, so that it overlaps the next step.
// RCCL is bound at run time (dlopen: the library torch already loaded, or ROCm's): libcrafter_hip.so itself links no
// communication library, and a single-GPU user never touches this code.
#include <dlfcn.h>

namespace {
struct NcclId { char internal[128]; };
typedef void* NcclComm;
struct Rccl {
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // The copy the process has loaded ALREADY comes first (torch.distributed's "nccl" backend brings its own librccl: a second
    // copy next to it would be a second set of the library's global state -- ADVICE r5); only then a search by name / path.
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (lib) break;
    }
    if (!lib)
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
      }
    if (!lib) {
      r.err = std::string("RCCL not found (dlopen librccl.so): ") + (dlerror() ? dlerror() : "");
      return;
    }
    r.GetUniqueId = (int (*)(NcclId*))dlsym(lib, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(NcclComm*, int, NcclId, int))dlsym(lib, "ncclCommInitRank");
    r.CommDestroy = (int (*)(NcclComm))dlsym(lib, "ncclCommDestroy");
    r.AllGather = (int (*)(const void*, void*, size_t, int, NcclComm, hipStream_t))dlsym(lib, "ncclAllGather");
    r.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
    if (!r.ok) r.err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather";
  });
  return r;
}
constexpr int kNcclUint8 = 1;   // ncclUint8 (rccl.h ncclDataType_t)
constexpr int kExchangeMaxSlots = 8;
thread_local std::string g_exchange_error;
}  // namespace

struct crafter_exchange {
  NcclComm comm = nullptr;
  int rank = 0, world = 1, slots = 2;
  hipStream_t stream = nullptr;                       // the collectives run here, beside the launch stream's next step
  hipEvent_t ready[kExchangeMaxSlots] = {};           // the record of slot k is complete on the launch stream
  hipEvent_t done[kExchangeMaxSlots] = {};            // its all-gather is complete
  bool in_flight[kExchangeMaxSlots] = {};
  std::string err;
};

static int xfail(crafter_exchange* x, const std::string& msg) {
  if (x) x->err = msg; else g_exchange_error = msg;
  return 1;
}

extern "C" {

int crafter_exchange_unique_id(uint8_t id[128]) {
  Rccl& r = rccl();
  if (!r.ok) return xfail(nullptr, r.err);
  NcclId nid;
  int rc = r.GetUniqueId(&nid);
  if (rc != 0) return xfail(nullptr, std::string("ncclGetUniqueId: ") + r.GetErrorString(rc));
  memcpy(id, nid.internal, 128);
  return 0;
}

int crafter_exchange_create(const uint8_t id[128], int32_t rank, int32_t world, int32_t slots, crafter_exchange** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world || slots < 1 || slots > kExchangeMaxSlots)
    return xfail(nullptr, "crafter_exchange_create: bad argument");
  Rccl& r = rccl();
  if (!r.ok) return xfail(nullptr, r.err);
  crafter_exchange* x = new crafter_exchange();
  x->rank = rank; x->world = world; x->slots = slots;
  NcclId nid;
  memcpy(nid.internal, id, 128);
  int rc = r.CommInitRank(&x->comm, world, nid, rank);
  if (rc != 0) {
    delete x;
    return xfail(nullptr, std::string("ncclCommInitRank: ") + r.GetErrorString(rc));
  }
  bool ok = hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking) == hipSuccess;
  for (int k = 0; k < slots && ok; k++)
    ok = hipEventCreateWithFlags(&x->ready[k], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&x->done[k], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    crafter_exchange_destroy(x);
    return xfail(nullptr, "crafter_exchange_create: cannot create the exchange's stream / events");
  }
  *out = x;
  return 0;
}

void crafter_exchange_destroy(crafter_exchange* x) {
  if (!x) return;
  if (x->stream) (void)hipStreamSynchronize(x->stream);
  if (x->comm) (void)rccl().CommDestroy(x->comm);
  for (int k = 0; k < kExchangeMaxSlots; k++) {
    if (x->ready[k]) (void)hipEventDestroy(x->ready[k]);
    if (x->done[k]) (void)hipEventDestroy(x->done[k]);
  }
  if (x->stream) (void)hipStreamDestroy(x->stream);
  delete x;
}

int crafter_exchange_wait(crafter_exchange* x, int32_t slot, void* stream) {
  if (!x || slot < 0 || slot >= x->slots) return xfail(x, "crafter_exchange_wait: bad argument");
  if (!x->in_flight[slot]) return 0;
  hipError_t e = hipStreamWaitEvent((hipStream_t)stream, x->done[slot], 0);
  if (e != hipSuccess) return xfail(x, std::string("crafter_exchange_wait: ") + hipGetErrorString(e));
  return 0;
}

int crafter_step_exchange(crafter_handle* h, crafter_exchange* x, int32_t slot, const int32_t* actions, uint8_t* send, uint8_t* recv,
                          int64_t record_bytes, int64_t off_reward, int64_t off_done, int32_t with_obs, void* stream) {
  if (!x || slot < 0 || slot >= x->slots || !send || !recv || record_bytes <= 0 || off_reward < 0 || off_done <= off_reward ||
      off_done >= record_bytes || (off_reward & 3))
    return xfail(x, "crafter_step_exchange: bad argument");
  if (!h) return xfail(x, "crafter_step_exchange: null env handle");
  // the slot's previous gather read `send` and wrote `recv`: the kernels that overwrite the record come behind it
  if (crafter_exchange_wait(x, slot, stream)) return 1;
  if (crafter_step(h, actions, with_obs ? send : nullptr, (float*)(send + off_reward), send + off_done, stream)) {
    x->err = std::string("crafter_step: ") + crafter_last_error(h);
    return 1;
  }
  hipError_t e = hipEventRecord(x->ready[slot], (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(x->stream, x->ready[slot], 0);
  if (e != hipSuccess) return xfail(x, std::string("crafter_step_exchange: fork to the exchange stream: ") + hipGetErrorString(e));
  Rccl& r = rccl();
  int rc = r.AllGather(send, recv, (size_t)record_bytes, kNcclUint8, x->comm, x->stream);
  if (rc != 0) return xfail(x, std::string("ncclAllGather: ") + r.GetErrorString(rc));
  e = hipEventRecord(x->done[slot], x->stream);
  if (e != hipSuccess) return xfail(x, std::string("crafter_step_exchange: hipEventRecord: ") + hipGetErrorString(e));
  x->in_flight[slot] = true;
  return 0;
}

const char* crafter_exchange_error(const crafter_exchange* x) { return x ? x->err.c_str() : g_exchange_error.c_str(); }

}  // extern "C"
