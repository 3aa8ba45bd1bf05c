/* TEST INFRASTRUCTURE: the whole hot path driven from plain C99 through include/crafter_hip.h alone -- what a binding
 * in the reference's position (or in Rust / Go) would do: fill crafter_config the way crafter.Env.__init__ does
 * (env.py:27-56), hand over the host tables, allocate and bind the state buffers, reset, step, read obs / reward / done.
 * No Python, no C++, no struct transcribed from elsewhere.  tests/test_gpu_c_boundary.py compiles it with
 * gcc -std=c99, feeds it the host tables (a flat file: the textures / daylight / vignette come from Pillow / numpy, as
 * in the reference) and an action tape, and compares what it writes with the oracle.
 *
 *   boundary_test <tables.bin> <tape.bin> <out.bin> <num_envs> <steps> <first_seed>      (linked against libcrafter_hip.so)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "crafter_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static void* read_blob(FILE* f, int64_t* bytes) {
  void* p;
  if (fread(bytes, sizeof(*bytes), 1, f) != 1) return NULL;
  p = malloc(*bytes > 0 ? (size_t)*bytes : 1);
  if (*bytes > 0 && fread(p, 1, (size_t)*bytes, f) != (size_t)*bytes) return NULL;
  return p;
}

static void* dev_zeros(size_t bytes) {
  void* p = NULL;
  if (hipMalloc(&p, bytes) != hipSuccess) return NULL;
  if (hipMemset(p, 0, bytes) != hipSuccess) return NULL;
  return p;
}

int main(int argc, char** argv) {
  if (argc != 7) {
    fprintf(stderr, "usage: %s tables tape out num_envs steps first_seed\n", argv[0]);
    return 1;
  }
  int n = atoi(argv[4]), steps = atoi(argv[5]);
  long first_seed = atol(argv[6]);

  int32_t sizes[6];
  crafter_struct_sizes(sizes);
  if (sizes[0] != (int)sizeof(crafter_obj) || sizes[1] != (int)sizeof(crafter_env_rec) || sizes[2] != (int)sizeof(crafter_rules) ||
      sizes[3] != (int)sizeof(crafter_config) || sizes[4] != (int)sizeof(crafter_state_ptrs)) {
    fprintf(stderr, "struct sizes of crafter_hip_types.h differ from the library's\n");
    return 1;
  }

  /* ---- host tables: rules, atlas, tex_tile, tex_icon, tex_digit, tex_alpha, item_pos, daylight, vignette, unit255 */
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int64_t nb[10];
  void* blob[10];
  for (int i = 0; i < 10; i++) {
    blob[i] = read_blob(f, &nb[i]);
    if (!blob[i]) { fprintf(stderr, "short tables file\n"); return 1; }
  }
  fclose(f);
  const crafter_rules* rules = (const crafter_rules*)blob[0];
  if (nb[0] != (int64_t)sizeof(crafter_rules)) { fprintf(stderr, "rules blob size\n"); return 1; }

  /* ---- crafter.Env(area=(64, 64), view=(9, 9), size=(64, 64), reward=True, length=10000) (env.py:27-46) */
  crafter_config c;
  memset(&c, 0, sizeof(c));
  c.num_envs = n;
  c.W = 64; c.H = 64;
  c.view_w = 9; c.view_h = 9;
  c.size_w = 64; c.size_h = 64;
  c.unit_x = c.size_w / c.view_w; c.unit_y = c.size_h / c.view_h;                       /* env.py:42 */
  int item_rows = (rules->n_items + c.view_w - 1) / c.view_w;                            /* env.py:43 */
  c.local_gw = c.view_w; c.local_gh = c.view_h - item_rows;
  c.item_gw = c.view_w; c.item_gh = item_rows;
  c.border_x = (c.size_w - c.unit_x * c.view_w) / 2; c.border_y = (c.size_h - c.unit_y * c.view_h) / 2;   /* env.py:127 */
  c.icon_w = (int)(0.8 * c.unit_x); c.icon_h = (int)(0.8 * c.unit_y);                   /* engine.py:239 */
  c.digit_w = (int)(0.6 * c.unit_x); c.digit_h = (int)(0.6 * c.unit_y);                 /* engine.py:246 */
  c.max_objects = 256;
  c.nchunk_x = (c.W + CRAFTER_CHUNK - 1) / CRAFTER_CHUNK; c.nchunk_y = (c.H + CRAFTER_CHUNK - 1) / CRAFTER_CHUNK;
  c.length = 10000;
  c.update_dist = 2 * (c.view_w > c.view_h ? c.view_w : c.view_h);                      /* env.py:88 */
  c.n_daylight = c.length + 2;
  c.auto_reset = 0; c.want_semantic = 0; c.render_obs = 1; c.reward = 1;
  c.gen_period = -1;
  if (nb[7] != (int64_t)sizeof(double) * c.n_daylight) { fprintf(stderr, "daylight table size\n"); return 1; }

  crafter_handle* h = NULL;
  if (crafter_create(&c, &h)) { fprintf(stderr, "crafter_create: %s\n", crafter_last_error(NULL)); return 3; }
  crafter_host_tables t;
  memset(&t, 0, sizeof(t));
  t.rules = rules;
  t.atlas = (const uint8_t*)blob[1];     t.atlas_bytes = (size_t)nb[1];
  t.tex_tile = (const int32_t*)blob[2];  t.n_tex_tile = (int32_t)(nb[2] / 4);
  t.tex_icon = (const int32_t*)blob[3];  t.n_tex_icon = (int32_t)(nb[3] / 4);
  t.tex_digit = (const int32_t*)blob[4]; t.n_tex_digit = (int32_t)(nb[4] / 4);
  t.tex_alpha = (const uint8_t*)blob[5]; t.n_tex_alpha = (int32_t)nb[5];
  t.item_pos = (const int32_t*)blob[6];  t.n_item_pos = (int32_t)(nb[6] / 4);
  t.daylight = (const double*)blob[7];   t.n_daylight = (int32_t)(nb[7] / 8);
  t.vignette = (const double*)blob[8];   t.n_vignette = (int32_t)(nb[8] / 8);
  t.unit255 = (const float*)blob[9];     t.n_unit255 = (int32_t)(nb[9] / 4);
  if (crafter_upload_tables(h, &t)) { fprintf(stderr, "crafter_upload_tables: %s\n", crafter_last_error(h)); return 3; }

  /* ---- state buffers (sizes: the comments of crafter_state_ptrs) */
  size_t cells = (size_t)c.W * c.H, nch = (size_t)c.nchunk_x * c.nchunk_y;
  crafter_state_ptrs s;
  memset(&s, 0, sizeof(s));
  s.mat = (uint8_t*)dev_zeros(n * cells);
  s.objmap = (uint16_t*)dev_zeros(n * cells * 2);
  s.objs = (crafter_obj*)dev_zeros((size_t)n * c.max_objects * sizeof(crafter_obj));
  s.mt = (uint32_t*)dev_zeros((size_t)n * CRAFTER_MT_N * 4);
  s.rec = (crafter_env_rec*)dev_zeros((size_t)n * sizeof(crafter_env_rec));
  s.chunk_order = (uint16_t*)dev_zeros(n * nch * 2);
  s.chunk_seen = (uint8_t*)dev_zeros(n * nch);
  s.census = (int32_t*)dev_zeros(n * nch * 5 * 4);
  s.terminal = (int32_t*)dev_zeros((size_t)n * (CRAFTER_MAX_ACH + 4) * 4);
  if (!s.mat || !s.objmap || !s.objs || !s.mt || !s.rec || !s.chunk_order || !s.chunk_seen || !s.census || !s.terminal) {
    fprintf(stderr, "hipMalloc failed\n");
    return 2;
  }
  crafter_env_rec* rec = (crafter_env_rec*)calloc((size_t)n, sizeof(crafter_env_rec));
  for (int i = 0; i < n; i++) {
    rec[i].seed_lane = (uint64_t)(first_seed + i);   /* CPython: hash(k) == k for 0 <= k < 2^61 - 1 (env.py:74) */
    rec[i].mt_pos = CRAFTER_MT_N;
    rec[i].nobj = 1;
  }
  CHECK_HIP(hipMemcpy(s.rec, rec, (size_t)n * sizeof(crafter_env_rec), hipMemcpyHostToDevice));
  if (crafter_bind_state(h, &s)) { fprintf(stderr, "crafter_bind_state: %s\n", crafter_last_error(h)); return 3; }

  size_t obs_bytes = (size_t)n * c.size_h * c.size_w * 3;
  uint8_t* d_obs = (uint8_t*)dev_zeros(obs_bytes);
  float* d_reward = (float*)dev_zeros((size_t)n * 4);
  uint8_t* d_done = (uint8_t*)dev_zeros((size_t)n);
  int32_t* d_act = (int32_t*)dev_zeros((size_t)n * 4);
  uint8_t* obs = (uint8_t*)malloc(obs_bytes);
  float* reward = (float*)malloc((size_t)n * 4);
  uint8_t* done = (uint8_t*)malloc((size_t)n);
  int32_t* tape = (int32_t*)malloc((size_t)steps * n * 4);
  f = fopen(argv[2], "rb");
  if (!f || fread(tape, 4, (size_t)steps * n, f) != (size_t)steps * n) { fprintf(stderr, "tape\n"); return 1; }
  fclose(f);
  FILE* out = fopen(argv[3], "wb");
  if (!out) { perror(argv[3]); return 1; }

  /* ---- Env.reset() then Env.step(a) per tape row, everything on the default stream */
  if (crafter_reset(h, NULL, d_obs, NULL)) { fprintf(stderr, "crafter_reset: %s\n", crafter_last_error(h)); return 3; }
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(obs, d_obs, obs_bytes, hipMemcpyDeviceToHost));
  fwrite(obs, 1, obs_bytes, out);
  int single = steps - steps / 2;   /* the first half one Env.step at a time, the rest in ONE crafter_step_n call */
  for (int k = 0; k < single; k++) {
    CHECK_HIP(hipMemcpy(d_act, tape + (size_t)k * n, (size_t)n * 4, hipMemcpyHostToDevice));
    if (crafter_step(h, d_act, d_obs, d_reward, d_done, NULL)) { fprintf(stderr, "crafter_step: %s\n", crafter_last_error(h)); return 3; }
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(obs, d_obs, obs_bytes, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(reward, d_reward, (size_t)n * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(done, d_done, (size_t)n, hipMemcpyDeviceToHost));
    fwrite(obs, 1, obs_bytes, out);
    fwrite(reward, 4, (size_t)n, out);
    fwrite(done, 1, (size_t)n, out);
  }
  if (steps > single) {   /* the loop `for a in tape: env.step(a)` as one call: [T][N] buffers, same outputs */
    int T = steps - single;
    uint8_t* d_obs_n = (uint8_t*)dev_zeros(obs_bytes * T);
    float* d_reward_n = (float*)dev_zeros((size_t)n * 4 * T);
    uint8_t* d_done_n = (uint8_t*)dev_zeros((size_t)n * T);
    int32_t* d_act_n = (int32_t*)dev_zeros((size_t)n * 4 * T);
    CHECK_HIP(hipMemcpy(d_act_n, tape + (size_t)single * n, (size_t)n * 4 * T, hipMemcpyHostToDevice));
    if (crafter_step_n(h, T, d_act_n, d_obs_n, d_reward_n, d_done_n, NULL)) { fprintf(stderr, "crafter_step_n: %s\n", crafter_last_error(h)); return 3; }
    CHECK_HIP(hipDeviceSynchronize());
    for (int k = 0; k < T; k++) {
      CHECK_HIP(hipMemcpy(obs, d_obs_n + (size_t)k * obs_bytes, obs_bytes, hipMemcpyDeviceToHost));
      CHECK_HIP(hipMemcpy(reward, d_reward_n + (size_t)k * n, (size_t)n * 4, hipMemcpyDeviceToHost));
      CHECK_HIP(hipMemcpy(done, d_done_n + (size_t)k * n, (size_t)n, hipMemcpyDeviceToHost));
      fwrite(obs, 1, obs_bytes, out);
      fwrite(reward, 4, (size_t)n, out);
      fwrite(done, 1, (size_t)n, out);
    }
  }
  /* info['inventory'] / info['achievements'] of the last step, and the sticky status, from the bound record */
  CHECK_HIP(hipMemcpy(rec, s.rec, (size_t)n * sizeof(crafter_env_rec), hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) {
    if (rec[i].status) { fprintf(stderr, "env %d: status %#x\n", i, rec[i].status); return 4; }
    fwrite(rec[i].inv, 4, CRAFTER_MAX_ITEMS, out);
    fwrite(rec[i].ach, 4, CRAFTER_MAX_ACH, out);
  }
  fclose(out);
  crafter_destroy(h);
  printf("ok: %d envs x %d steps through the C ABI (%d of them in one crafter_step_n call)\n", n, steps, steps - single);
  return 0;
}
