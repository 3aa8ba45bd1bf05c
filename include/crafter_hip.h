/* libcrafter_hip.so -- C ABI of the MI355X-native batched Crafter hot path.
 *
 * The reference (danijar/crafter) is pure Python and has no FFI; the boundary it offers for this
 * path is the Python surface of crafter/env.py:
 *     Env.__init__ (env.py:27-56)  Env.reset (env.py:70-81)  Env.step (env.py:83-118)
 *     Env.render (env.py:120-130)  info['semantic'] (engine.py:251-264)
 * Each entry point below names the reference interface it replaces.  The host side that binds
 * them (ctypes) and mirrors crafter.Env is crafter_amd/{lib,batched,env}.py; INTEGRATION.md shows
 * the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - "device pointer" = address in GPU memory owned by the CALLER (e.g. torch tensor data_ptr);
 *     the library never frees or reallocates caller memory; its own scratch lives in the handle;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only
 *     ENQUEUES work on that stream, there are no hidden synchronisations after crafter_create /
 *     crafter_upload_tables;
 *   - return 0 on success, non-zero on error with text in crafter_last_error(); nothing throws;
 *   - calls on one handle must be serialised by the caller.
 *
 * Struct layouts (crafter_config, crafter_rules, crafter_state_ptrs, crafter_obj, crafter_env_rec) are plain C99
 * in crafter_hip_types.h, included below: this header is self-contained for a C / Rust / Go / Java binding
 * (tests/c/boundary_test.c drives the whole path from C with nothing else).  The kernels' own C++ definitions
 * (crafter_amd/csrc/types.hpp) are static_asserted field by field against that file when the library is built;
 * crafter_amd/abi.py is the ctypes mirror, and crafter_struct_sizes lets any binding verify its own.
 */
#ifndef CRAFTER_HIP_H_
#define CRAFTER_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct crafter_handle crafter_handle;
#ifdef CRAFTER_HIP_INTERNAL   /* the library itself: the C names are its own C++ structs (checked against the C layouts) */
typedef struct crafter_config crafter_config;          /* crafter::Config    */
typedef struct crafter_rules crafter_rules;            /* crafter::Rules     */
typedef struct crafter_state_ptrs crafter_state_ptrs;  /* crafter::StatePtrs */
#else
#include "crafter_hip_types.h"
#endif

/* Host-side tables handed to crafter_upload_tables (all HOST pointers, copied by the library).
 * They carry everything the reference evaluates with numpy / Pillow / its yaml at run time:
 * data.yaml rules (constants.py:6-8), the resized textures (engine.py:131-142), daylight(step)
 * (env.py:135-139) and the night vignette (engine.py:213-218). */
typedef struct crafter_host_tables {
  const crafter_rules* rules;
  const uint8_t* atlas;      size_t atlas_bytes;   /* RGBA texels, [x][y] per texture          */
  const int32_t* tex_tile;   int32_t n_tex_tile;   /* byte offsets, types.hpp TEX_* slots      */
  const int32_t* tex_icon;   int32_t n_tex_icon;   /* per item                                 */
  const int32_t* tex_digit;  int32_t n_tex_digit;  /* '1'..'9' at [1..9], 'unknown' at [10]    */
  const uint8_t* tex_alpha;  int32_t n_tex_alpha;  /* 1 = source PNG had an alpha channel      */
  const int32_t* item_pos;   int32_t n_item_pos;   /* [items][4] icon x,y / digit x,y          */
  const double* daylight;    int32_t n_daylight;
  const double* vignette;    int32_t n_vignette;   /* local_w * local_h                        */
  const float* unit255;      int32_t n_unit255;    /* 256                                      */
} crafter_host_tables;

/* sizeof(Obj, EnvRec, Rules, Config, StatePtrs, TablePtrs) as compiled, for binding self-checks. */
void crafter_struct_sizes(int32_t out[6]);

/* ABI revision of this header. */
int32_t crafter_abi_version(void);

/* Replaces Env.__init__ (env.py:27-56) for a batch of cfg->num_envs environments. */
int crafter_create(const crafter_config* cfg, crafter_handle** out);
void crafter_destroy(crafter_handle* h);

/* Replaces the import-time loading of data.yaml / assets and the numpy evaluation of daylight and
 * vignette (constants.py:6-8, engine.py:122-129,213-218, env.py:135-139).  Synchronous copy. */
int crafter_upload_tables(crafter_handle* h, const crafter_host_tables* t);

/* Env(length=None) (env.py:29: no time limit; env.py:135-139 _update_time is evaluated by the host into the daylight table,
 * one value per step of an episode): replaces the handle's daylight table by a longer one -- daylight[0 .. n), n greater
 * than the current size, the first entries equal to the current table's -- so that an episode can run past the table it
 * started with.  A step beyond the table is clamped to its last entry and sets CRAFTER_ST_STEP_OVERFLOW.  Synchronous copy;
 * launches already enqueued keep the table they were launched with.  Handles created with fewer than 1024 daylight steps
 * cannot grow. */
int crafter_extend_daylight(crafter_handle* h, const double* daylight, int32_t n);

/* Registers the caller-owned device buffers holding the world state (sizes: crafter_amd/state.py). */
int crafter_bind_state(crafter_handle* h, const crafter_state_ptrs* state);

/* Bytes of LDS one environment's workgroup uses (diagnostics / occupancy planning). */
int32_t crafter_lds_bytes(const crafter_handle* h);

/* 1: the world maps are staged in LDS and the cell -> slot map (the reference's World._obj_map,
 * engine.py:32) is derived state, rebuilt from the slot table (World._objects, engine.py:33) at every
 * stage-in: crafter_state_ptrs.objmap is then never read or written.  0: large world, both maps live
 * in HBM and objmap is kept current. */
int32_t crafter_slot_map_derived(const crafter_handle* h);

/* Environment variables.  The library reads four, all DISPATCH OVERRIDES between kernels it ships (tests drive each kernel at
 * every batch size with them; none is needed in normal use), at crafter_create:
 *   CRAFTER_ORDER=0|1       dispatch order of the step launch (slow envs first) never / always   (default: batches > 1280 envs)
 *   CRAFTER_SPLIT=0|1       the default instance as one fused step kernel / as rules kernel + frame kernel (default: the pair
 *                           only when no frame is drawn)
 *   CRAFTER_STEP_WIDE=0|1   the default instance with 512 threads per env never / always          (default: batches <= 512 envs)
 *   CRAFTER_STEP_EARLY=0|1  the default instance as the kernel whose day frames begin before the rules end, never / always
 *                                                                                                  (default: batches >= 2048 envs)
 * Experiment knobs and timing probes exist only in builds with -DCRAFTER_PROBES (Diagnostics). */

/* Which instance of the step kernel crafter_step launches for this handle (diagnostics): bit 2 = maps staged in
 * LDS, bit 1 = the default geometry of crafter.Env() (env.py:27-46) compiled in, bit 0 = the uploaded rules equal
 * the compiled-in data.yaml (call after crafter_upload_tables).  7 = the fast path everybody should be on.  Bit 3 (9): a
 * world too large for LDS (maps and slot table stay in global memory) seen through the default view with the default rules,
 * both compiled in -- crafter_step_kernel<0, 2, 1>, BASELINE configs[3]. */
int32_t crafter_step_instance(const crafter_handle* h);

/* Replaces Env.reset (env.py:70-81) for every env whose mask byte is non-zero (mask == NULL: all).
 * mask: device uint8[num_envs].  obs: device uint8[num_envs][size_h][size_w][3] or NULL. */
int crafter_reset(crafter_handle* h, const uint8_t* mask, uint8_t* obs, void* stream);

/* Replaces Env.step (env.py:83-118) for all envs.
 * actions: device int32[num_envs]; obs as above; reward: device float[num_envs];
 * done: device uint8[num_envs].  With cfg->auto_reset, envs that finished are regenerated
 * (Env.reset) before the call's work completes and their obs is the new episode's first frame.
 * info[...] of the reference is read from the bound state buffers (EnvRec.inv/ach/..., semantic). */
int crafter_step(crafter_handle* h, const int32_t* actions, uint8_t* obs, float* reward, uint8_t* done,
                 void* stream);
/* Streams: a handle's calls are ordered by the stream they are issued on.  Changing the stream between two calls
 * (crafter_reset on one, crafter_step on another) is allowed: the library makes the new stream wait for the work the
 * handle still has in flight on the previous one and for the world pool's side streams, then carries on there.  Two
 * streams ALTERNATING every call therefore serialise; use one stream per handle.
 * With auto-reset and the world pool running, an env that finishes and finds no world in the pool (all but never one) is
 * regenerated by a second kernel behind the step launch, on the same stream: every output of the call is complete in
 * stream order. */

/* `steps` consecutive calls of crafter_step in one (Env.step, env.py:83-118, in a loop such as run_random.py:36-44) for
 * policies that choose their actions without looking at the observations (random, scripted, action repeat):
 * actions: device int32[steps][num_envs]; obs: device uint8[steps][num_envs][size_h][size_w][3] or NULL;
 * reward: device float[steps][num_envs]; done: device uint8[steps][num_envs].  Bit-identical to the loop; faster because
 * an env starts its step t + 1 without waiting for every other env's step t.  One launch covers the steps up to the world
 * pool's next generation batch, or 16 steps when the pool is off (so that an env that finished waits at most that long for the
 * launch boundary its inline regeneration happens at). */
int crafter_step_n(crafter_handle* h, int32_t steps, const int32_t* actions, uint8_t* obs, float* reward, uint8_t* done,
                   void* stream);

/* Multi-GPU (no reference counterpart: the reference is one env per process).  Envs shard by index, one process
 * per GPU; the one exchange of the path is the learner-side all-gather of every rank's packed (obs, reward, done) record
 * each step.  These entry points enqueue it from C -- through torch.distributed the same loop body costs the host more
 * than the step costs the GPU at 512 envs per rank.  RCCL is bound with dlopen at the first call (the librccl.so the
 * process has loaded, or ROCm's); a process that never calls them needs no RCCL.
 *   crafter_exchange_unique_id  rank 0 draws the communicator id (ncclGetUniqueId) and hands the 128 bytes to every rank
 *                               (any side channel: torch.distributed broadcast, MPI, a file)
 *   crafter_exchange_create     every rank, collectively (ncclCommInitRank); slots = records in flight (1..8, usually 2)
 *   crafter_step_exchange       crafter_step with its outputs INSIDE `send` -- obs at byte 0 (with_obs), reward at off_reward
 *                               (4-byte aligned), done at off_done -- then ncclAllGather(send -> recv[world][record_bytes]) on
 *                               the exchange's own stream, ordered behind the step's kernels by an event: it overlaps the
 *                               next step.  Before the kernels overwrite `send` the call makes `stream` wait for the slot's
 *                               previous gather.  send / recv: device memory, distinct per slot, caller-owned.
 *   crafter_exchange_wait       `stream` waits for the slot's gather (before anything reads recv)
 * All return 0 or 1 + crafter_exchange_error (thread-local text for the two calls without an exchange).
 * STATUS: exercised with ONE rank only so far (one GPU per test box; tests/test_gpu_dist.py).  A caller with more ranks should
 * check its first gather against a collective it trusts, as bench.py does (the first step's record against
 * torch.distributed.all_gather_into_tensor on every rank, falling back to crafter_amd.dist.StepExchange if any rank disagrees). */
typedef struct crafter_exchange crafter_exchange;
int crafter_exchange_unique_id(uint8_t id[128]);
int crafter_exchange_create(const uint8_t id[128], int32_t rank, int32_t world, int32_t slots, crafter_exchange** out);
void crafter_exchange_destroy(crafter_exchange* x);
int crafter_step_exchange(crafter_handle* h, crafter_exchange* x, int32_t slot, const int32_t* actions, uint8_t* send, uint8_t* recv,
                          int64_t record_bytes, int64_t off_reward, int64_t off_done, int32_t with_obs, void* stream);
int crafter_exchange_wait(crafter_exchange* x, int32_t slot, void* stream);
const char* crafter_exchange_error(const crafter_exchange* x);

/* Diagnostics (no reference counterpart): the order in which the next crafter_step dispatches the envs -- those whose next
 * step draws a night frame or balances the chunks first -- into host int32[num_envs]; synchronises the device.
 * Returns 2 when the handle keeps no order (few envs, no auto-reset, CRAFTER_ORDER=0). */
int crafter_debug_dispatch_order(crafter_handle* h, int32_t* out);
/* Experiments: dispatch in the caller's order (device int32[num_envs], must be a permutation and stay alive) until called
 * with NULL.  Returns 2 when the handle keeps no order. */
int crafter_debug_set_dispatch_order(crafter_handle* h, const int32_t* order);

/* Replaces Env.render() at the configured size (env.py:120-130) for masked envs (NULL: all):
 * re-draws the current frame into out (same layout as obs) and, exactly like the reference,
 * draws the night noise from each env's RNG again (engine.py:208-209). */
int crafter_render(crafter_handle* h, const uint8_t* mask, uint8_t* out, void* stream);

/* Measurement aid (no reference counterpart): when enabled, crafter_step attaches HIP start / stop events to
 * its two kernels (hipExtLaunchKernelGGL: the kernels' own execution time on the launch stream, what a
 * profiler reports).  crafter_get_timing waits for the recorded events, returns the SUM of step-kernel and
 * auto-reset-kernel durations in ms over `launches` calls and clears them. */
int crafter_set_timing(crafter_handle* h, int enable);
int crafter_get_timing(crafter_handle* h, double* step_ms, double* reset_ms, int32_t* launches);

/* World pool (no reference counterpart: Env.reset's worldgen, worldgen.py:10-18, runs ahead of time on side
 * streams so that an auto-reset adopts a finished world).  Returns 0 = pool off (no auto-reset / gen_period < 0),
 * 1 = running, 2 = disabled after a HIP error in its scheduler (text in crafter_pool_error; stepping stays
 * correct, finished envs regenerate inline), -1 = null handle.  launched / trusted (may be NULL): generation
 * batches launched so far / known complete. */
int crafter_pool_status(const crafter_handle* h, uint32_t* launched, uint32_t* trusted);
const char* crafter_pool_error(const crafter_handle* h);

/* Unit-test access (no reference counterpart) to the arithmetic the world generator evaluates on the device, so that it
 * can be compared bit for bit with the CPU oracle: mode 0: out[i] = noise3(x[i], y[i], z[i]) of the OpenSimplex instance
 * whose permutation is perm[256] (the third-party opensimplex package behind worldgen.py:11,84-87); mode 1:
 * out[i] = 1 / (1 + exp(-x[i])) (worldgen.py:27) with the library's pinned, correctly rounded exponential (np.exp is a
 * different function on different hosts: csrc/worldgen.hpp exp_cr); mode 2: out[i] = 4 - sqrt(x[i]) (worldgen.py:25);
 * mode 3: out[i] = exp_cr(x[i]).  All pointers are device pointers; perm / y / z may be NULL for modes 1 to 3.  Errors are reported through crafter_last_error(NULL). */
int crafter_debug_eval(int mode, const uint8_t* perm, const double* x, const double* y, const double* z, double* out,
                       int64_t n, void* stream);

/* Last error text of this handle (or of the failed crafter_create when h == NULL). */
const char* crafter_last_error(const crafter_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* CRAFTER_HIP_H_ */
