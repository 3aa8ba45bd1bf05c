// TEST INFRASTRUCTURE -- CPU execution of the kernel bodies through WaveHost (see wave_host.hpp).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "wave_host.hpp"
#include "../../crafter_amd/csrc/env_kernels.hpp"

using namespace crafter;

extern "C" {

void hostsim_struct_sizes(int32_t* out) {
  out[0] = sizeof(Obj);
  out[1] = sizeof(EnvRec);
  out[2] = sizeof(Rules);
  out[3] = sizeof(Config);
  out[4] = sizeof(StatePtrs);
  out[5] = sizeof(TablePtrs);
}

int hostsim_lds_bytes(const Config* cfg) { return lds_layout(*cfg).total; }
int hostsim_slot_map_derived(const Config* cfg) { return lds_layout(*cfg).maps_in_lds; }
// LDS per workgroup of the step kernel's default instance, and of the two kernels of the split step
int hostsim_step_lds_bytes(const Config* cfg) { return lds_layout(*cfg, 1).total; }
int hostsim_rules_lds_bytes(const Config* cfg) { return lane_layout(*cfg).total; }
int hostsim_frame_lds_bytes(const Config* cfg) { return frame_layout(*cfg).total; }

// the renderer's static block (the library builds it on the device when the tables are uploaded)
long long hostsim_render_static_bytes(const Config* cfg) { return (long long)render_static_total_bytes(*cfg); }
void hostsim_build_static(const Config* cfg, const TablePtrs* tb, uint8_t* dst) {
  WaveHost w;
  Env<WaveHost> e(w, *cfg, *tb);
  RenderTarget rt = obs_target<WaveHost>(*cfg, *tb, nullptr, 0);
  Renderer<WaveHost> r(e, rt, dst, nullptr, nullptr);
  r.build_static(dst);
  r.build_lit_sprites(dst, 0, 1);
}

// The kernels' noise3 on its own: perm8[256] is the OpenSimplex permutation (oracle/noise.py builds the same one).
void hostsim_noise3(const uint8_t* perm8, const double* xs, const double* ys, const double* zs, double* out, int n) {
  uint8_t pg3[256];
  for (int i = 0; i < 256; i++) pg3[i] = (uint8_t)(perm8[i] % 24);
  static SimplexLds tab;
  simplex_fill_tables(&tab, [&](int n, auto body) { for (int i = 0; i < n; i++) body(i); });
  Simplex<WaveHost> sx{perm8, pg3, &tab};
  for (int i = 0; i < n; i++) out[i] = sx.noise3(xs[i], ys[i], zs[i]);
}

uint32_t hostsim_world_seed(uint64_t seed_lane, uint64_t episode) { return world_seed(seed_lane, episode); }

// the kernels' pinned exponential (worldgen.hpp exp_cr), compiled for the host: same operations as oracle/exp_cr.py
void hostsim_exp_cr(const double* x, double* out, int n) {
  for (int i = 0; i < n; i++) out[i] = exp_cr(x[i]);
}

// pool_mode: 0 = world pool off, 1 = pool on with generation right after every call (always trusted)
static void run_generation(const Config* cfg, const TablePtrs* tb, const StatePtrs* st, std::vector<uint8_t>& lds) {
  int32_t* q = st->gen_q;
  int count = q ? q[0] : 0;
  if (count > gen_q_capacity(*cfg)) count = gen_q_capacity(*cfg);
  for (int k = 0; k < count; k++) {
    // the product's three-kernel pipeline (seed -> classify -> resolve), each stage in freshly poisoned "LDS"
    int env = q[4 + 2 * k], episode = q[4 + 2 * k + 1];
    WaveHost w;
    memset(lds.data(), 0xCD, lds.size());
    gen_seed_body(w, lds.data(), env, episode, *cfg, *tb, *st);
    memset(lds.data(), 0xCD, lds.size());
    for (int part = 0, parts = gen_classify_parts(*cfg); part < parts; part++) {
      memset(lds.data(), 0xCD, lds.size());
      gen_classify_body(w, lds.data(), env, episode, part, parts, *cfg, *tb, *st);
    }
    memset(lds.data(), 0xCD, lds.size());
    gen_resolve_body(w, lds.data(), env, episode, 1u, *cfg, *tb, *st);
  }
  if (q) q[0] = 0;
}

int hostsim_reset(const Config* cfg, const TablePtrs* tb, const StatePtrs* st, const uint8_t* mask,
                  int pool_mode, uint8_t* obs) {
  std::vector<uint8_t> lds(lds_layout(*cfg).total + 64);
  for (int env = 0; env < cfg->num_envs; env++) {
    if (mask && !mask[env]) continue;
    memset(lds.data(), 0xCD, lds.size());
    WaveHost w;
    reset_body(w, lds.data(), env, *cfg, *tb, *st, obs, pool_mode ? 0 : -1);
  }
  if (pool_mode) run_generation(cfg, tb, st, lds);
  return 0;
}

// 1: steps run as the split pair rules half / frame half (crafter_rules_kernel + crafter_frame_kernel) where the library would
static int g_split = 0;
void hostsim_set_split(int on) { g_split = on; }
// 1 (default, as the library): the fused step generates a night frame's noise states ahead of the rules (noise_chain)
static int g_noise_ahead = 1;
void hostsim_set_noise_ahead(int on) { g_noise_ahead = on; }

int hostsim_step(const Config* cfg, const TablePtrs* tb, const StatePtrs* st, const int32_t* actions,
                 uint8_t* obs, float* reward, uint8_t* done, int pool_mode) {
  std::vector<uint8_t> lds(lds_layout(*cfg).total + frame_layout(*cfg).total + 64);
  // (the library splits the default instance only: default geometry AND the compiled-in rules)
  bool split = g_split && is_default_geometry(*cfg) && lds_layout(*cfg).maps_in_lds && lane_layout_ok(*cfg) &&
               memcmp(tb->rules, &kDefaultRules, sizeof(Rules)) == 0;
  StepCtl ctl;
  ctl.parity = 0;
  ctl.gen_parity = pool_mode ? 0 : -1;
  ctl.safe_seq = 0xffffffffu;
  ctl.early_frame = 1;   // (the library: batches larger than the chip holds at once; here always, so that the path is covered)
  // split step: the frame kernel's scratch (night pixels)
  static std::vector<uint32_t> night_px;
  night_px.resize((size_t)cfg->num_envs * frame_night_px_words(*cfg));
  static std::vector<uint32_t> noise_raw;   // the fused step's noise look-ahead scratch (noise_chain); g_noise_ahead 0: the in-frame pass
  noise_raw.resize((size_t)cfg->num_envs * kNoiseStates * MT_N);
  if (g_noise_ahead) ctl.noise_raw = noise_raw.data();
  bool frames = split && cfg->render_obs && obs;
  for (int env = 0; env < cfg->num_envs; env++) {
    memset(lds.data(), 0xCD, lds.size());
    WaveHost w;
    if (split) {
      step_body<WaveHost, -1, 1, LaneSlots, 1>(w, lds.data(), env, *cfg, *tb, *st, actions, obs, reward, done, ctl);
    } else if (is_default_geometry(*cfg))   // as crafter_step_kernel does: one-byte slot ids for crafter.Env()'s defaults
      step_body<WaveHost, -1, 0, uint8_t>(w, lds.data(), env, *cfg, *tb, *st, actions, obs, reward, done, ctl);
    else if (!lds_layout(*cfg).maps_in_lds) {   // crafter_step_kernel<0, 0, 0>: big_layout -- census in place, night pixels in global scratch
      StepCtl big = ctl;
      big.night_px = night_px.data();
      if (is_default_view(*cfg) && memcmp(tb->rules, &kDefaultRules, sizeof(Rules)) == 0)   // crafter_step_kernel<0, 2, 1>: the default rules compiled in
        step_body<WaveHost, 0, 1, FarSlot>(w, lds.data(), env, *cfg, *tb, *st, actions, obs, reward, done, big);
      else
        step_body<WaveHost, 0, 0, FarSlot>(w, lds.data(), env, *cfg, *tb, *st, actions, obs, reward, done, big);
    } else
      step_body<WaveHost, -1, 0, uint16_t>(w, lds.data(), env, *cfg, *tb, *st, actions, obs, reward, done, ctl);
  }
  if (frames) {   // the frame kernel
    for (int env = 0; env < cfg->num_envs; env++) {
      memset(lds.data(), 0xCD, lds.size());
      WaveHost wf;
      frame_body(wf, lds.data(), env, *cfg, *tb, *st, obs, night_px.data());
    }
  }
  if (cfg->auto_reset) {
    // same queue walk as crafter_requeue_reset_kernel
    int32_t* q = st->reset_q;
    int count = q ? q[0] : 0;
    for (int k = 0; k < count; k++) {
      memset(lds.data(), 0xCD, lds.size());
      WaveHost w;
      reset_body(w, lds.data(), q[4 + k], *cfg, *tb, *st, obs, ctl.gen_parity);
    }
    if (q) q[0] = 0;
    if (pool_mode) run_generation(cfg, tb, st, lds);
  }
  return 0;
}

// crafter_step_n as the library runs it: stretches of at most `stretch` steps (the generation period), every env through
// rollout_body, then the regeneration queue through requeue_rollout_body, then (pool on) the generation batch.
int hostsim_step_n(const Config* cfg, const TablePtrs* tb, const StatePtrs* st, int steps, const int32_t* actions,
                   uint8_t* obs, float* reward, uint8_t* done, int pool_mode, int stretch) {
  std::vector<uint8_t> lds(lds_layout(*cfg).total + 64);
  std::vector<int32_t> stalled_at((size_t)cfg->num_envs, -1);
  StepCtl ctl;
  ctl.parity = 0;
  ctl.gen_parity = pool_mode ? 0 : -1;
  ctl.safe_seq = 0xffffffffu;
  size_t n = (size_t)cfg->num_envs, obs_stride = n * (size_t)cfg->size_w * cfg->size_h * 3;
  for (int t0 = 0; t0 < steps; t0 += stretch) {
    int T = steps - t0 < stretch ? steps - t0 : stretch;
    const int32_t* a = actions + (size_t)t0 * n;
    uint8_t* o = obs ? obs + (size_t)t0 * obs_stride : nullptr;
    float* r = reward + (size_t)t0 * n;
    uint8_t* d = done + (size_t)t0 * n;
    for (int env = 0; env < cfg->num_envs; env++) {
      memset(lds.data(), 0xCD, lds.size());
      WaveHost w;
      if (is_default_geometry(*cfg))
        rollout_body<WaveHost, -1, 0, uint8_t>(w, lds.data(), env, *cfg, *tb, *st, a, o, r, d, ctl, T, obs_stride, stalled_at.data());
      else if (!lds_layout(*cfg).maps_in_lds) {   // crafter_rollout_kernel<0, 0, 0>: big_layout, the slot table in global memory
        static std::vector<uint32_t> night_px;
        night_px.resize((size_t)cfg->num_envs * frame_night_px_words(*cfg));
        StepCtl big = ctl;
        big.night_px = night_px.data();
        if (is_default_view(*cfg) && memcmp(tb->rules, &kDefaultRules, sizeof(Rules)) == 0)   // crafter_rollout_kernel<0, 2, 1>
          rollout_body<WaveHost, 0, 1, FarSlot>(w, lds.data(), env, *cfg, *tb, *st, a, o, r, d, big, T, obs_stride, stalled_at.data());
        else
          rollout_body<WaveHost, 0, 0, FarSlot>(w, lds.data(), env, *cfg, *tb, *st, a, o, r, d, big, T, obs_stride, stalled_at.data());
      } else
        rollout_body<WaveHost, -1, 0, uint16_t>(w, lds.data(), env, *cfg, *tb, *st, a, o, r, d, ctl, T, obs_stride, stalled_at.data());
    }
    if (cfg->auto_reset) {
      int32_t* q = st->reset_q;
      int count = q ? q[0] : 0;
      for (int k = 0; k < count; k++) {
        memset(lds.data(), 0xCD, lds.size());
        WaveHost w;
        requeue_rollout_body(w, lds.data(), q[4 + k], *cfg, *tb, *st, a, o, r, d, ctl, T, obs_stride, stalled_at.data());
      }
      if (q) q[0] = 0;
      if (pool_mode) run_generation(cfg, tb, st, lds);
    }
  }
  return 0;
}

int hostsim_render(const Config* cfg, const TablePtrs* tb, const StatePtrs* st, const uint8_t* mask, uint8_t* out) {
  std::vector<uint8_t> lds(lds_layout(*cfg).total + 64);
  for (int env = 0; env < cfg->num_envs; env++) {
    if (mask && !mask[env]) continue;
    memset(lds.data(), 0xCD, lds.size());
    WaveHost w;
    render_body(w, lds.data(), env, *cfg, *tb, *st, out);
  }
  return 0;
}

}  // extern "C"
