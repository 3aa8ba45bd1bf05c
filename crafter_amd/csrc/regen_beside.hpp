// Regeneration beside the step: the protocol between crafter_step_kernel and crafter_regen_server_kernel (crafter_hip.hip).
// No reference counterpart: Env.reset (env.py:70-81) of an env whose next world the pool does not hold runs CONCURRENTLY
// with the step launch instead of in a kernel behind it.  GPU only (waits between concurrently running kernels); the CPU
// harness keeps the queue-and-kernel form of the same bodies.
#pragma once
#include "env_kernels.hpp"

namespace crafter {

// Memory (crafter_hip.hip, zeroed at create): words[kRegenPushed] tickets handed out, [kRegenClosed] the sequence number of
// the newest launch whose env workgroups have all finished, [kRegenClaimed] tickets the server has taken, [kRegenServed]
// envs it has regenerated; ring[N] the (ticket + 1) << 32 | env of every ticket; counters[kRegenStripes], one 8-byte word
// per 128-byte line: low half = envs of this stripe handed over, high half = env workgroups of this stripe finished (both
// since create, mod 2^32; stripe = env mod 64).  Striped because 4096 workgroups counting themselves off on ONE word cost
// the launch 36 us (cross-XCD atomics on one line: profiles/r4v_regen_counter_ab.txt), and counted rather than flagged
// because block 0 reads 64 counters in one load where it needs sixteen rounds of loads for 4096 flags (r4w_regen_flags_ab.txt).
//
// The end of an env's workgroup.  Common case: nothing -- one fire-and-forget atomic went out before the frame (step_body).
// An env that needs a world the pool does not have:
// its state (store_env has run) is made visible device-wide, it takes a ticket, publishes (ticket, env), counts itself as
// handed over and only then as finished -- both halves of a stripe's word come from ONE 8-byte load, so whoever reads
// "all finished" reads hand-over counts that include every hand-over of the launch.
template <class W>
__device__ __forceinline__ void regen_handoff(W& w, const StepCtl& ctl, const Config& cfg, const StatePtrs& st, int env, bool handed) {
  uint32_t* stripe = ctl.regen_counters + (size_t)(env & (kRegenStripes - 1)) * kRegenStripeWords;
  if (handed) {
    __threadfence();
    w.sync();
    if (w.leader()) {
      uint32_t ticket = __hip_atomic_fetch_add(ctl.regen_words + kRegenPushed, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctl.regen_ring + ticket % (uint32_t)cfg.num_envs, ((uint64_t)(ticket + 1u) << 32) | (uint32_t)env,
                         __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (st.pool_stats) w.global_add(st.pool_stats + 1, 1);
      (void)__hip_atomic_fetch_add(stripe + 0, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      (void)__hip_atomic_fetch_add(stripe + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }   // (everybody else has counted itself off before its frame: step_body)
}

constexpr uint32_t kRegenSpinLimit = 1u << 23;   // block 0's polls (~2 us each): a protocol error ends as ST_PIPE_STALL on env 0, not as a hung device
constexpr uint32_t kRegenServeLimit = 1u << 28;  // the server's: it is in its queue the moment crafter_step returns, the launch it serves may sit
                                                 // behind minutes of the caller's own work on the launch stream (asleep, it costs nothing)

// Block 0 of the launch (its first wave, lane = stripe): leaves when every env workgroup has finished and everything handed
// over has been regenerated.  Polls are plain loads past the caches (relaxed, agent scope): no fence, no invalidation of the
// L2 the stepping workgroups live in.
template <class W>
__device__ inline void regen_close(W& w, const StepCtl& ctl, const Config& cfg, const StatePtrs& st) {
  if (!w.wave0()) return;
  const int lane = (int)w.tid();
  const uint32_t mine = (uint32_t)(cfg.num_envs / kRegenStripes + (lane < cfg.num_envs % kRegenStripes ? 1 : 0));
  const uint32_t want = ctl.regen_seq * mine;   // (the stripe's envs finish once per launch; sequence numbers start at 1)
  const uint64_t* word = (const uint64_t*)(ctl.regen_counters + (size_t)lane * kRegenStripeWords);
  bool closed = false;
  for (uint32_t polls = 0;; polls++) {
    uint64_t c = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t served = __hip_atomic_load(ctl.regen_words + kRegenServed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool all = __builtin_amdgcn_ballot_w64((uint32_t)(c >> 32) == want) == ~0ull;
    if (all) {
      uint32_t handed = (uint32_t)c;
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) handed += (uint32_t)__shfl_xor((int)handed, m);
      if (!closed && lane == 0) __hip_atomic_store(ctl.regen_words + kRegenClosed, ctl.regen_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the server may leave
      closed = true;
      if (served == handed) return;
    }
    if (all) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(4);
    if (polls >= kRegenSpinLimit) {
      if (lane == 0) st.rec[0].status |= ST_PIPE_STALL;
      return;
    }
  }
}

// One workgroup of the server: claims tickets while there are any, regenerates those envs (reset_body: world, first frame,
// the next requests to the pool) and counts them; leaves once the launch it serves is closed and no ticket is unclaimed
// (the counters read AFTER "closed" was seen: every hand-over of the launch precedes its closing).
template <class W>
__device__ inline void regen_serve(W& w, uint8_t* smem, const Config& cfg, const TablePtrs& tb, const StatePtrs& st, int gen_parity,
                                   uint8_t* obs, uint32_t* words, const uint64_t* ring, uint32_t seq, int* job) {
  bool closed = false;
  for (;;) {
    if (w.leader()) {
      int got = -1;
      for (uint32_t polls = 0; polls < kRegenServeLimit; polls++) {
        uint32_t pushed = __hip_atomic_load(words + kRegenPushed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t claimed = __hip_atomic_load(words + kRegenClaimed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)(pushed - claimed) > 0) {
          uint32_t expect = claimed;
          if (!__hip_atomic_compare_exchange_strong(words + kRegenClaimed, &expect, claimed + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT))
            continue;
          const uint64_t* slot = ring + claimed % (uint32_t)cfg.num_envs;
          uint64_t entry = 0;
          do {   // (ticket taken, entry not published yet: a few hundred nanoseconds)
            entry = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((uint32_t)(entry >> 32) != claimed + 1u && ++polls < kRegenServeLimit);
          got = (uint32_t)(entry >> 32) == claimed + 1u ? (int)(uint32_t)entry : -2;
          break;
        }
        if (closed) break;
        if ((int32_t)(__hip_atomic_load(words + kRegenClosed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) >= 0) {
          closed = true;   // one more look at the counters, issued after this value came back
          continue;
        }
        __builtin_amdgcn_s_sleep(32);
      }
      *job = got;
    }
    w.sync();
    int env = *job;
    w.sync();
    if (env < 0) {
      if (env == -2 && w.leader()) st.rec[0].status |= ST_PIPE_STALL;
      return;
    }
    __threadfence();   // every thread: what the env's workgroup stored, not a stale line of this CU's or XCD's caches
    reset_body(w, smem, env, cfg, tb, st, obs, gen_parity);
    __threadfence();
    w.sync();
    if (w.leader()) (void)__hip_atomic_fetch_add(words + kRegenServed, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace crafter
