// Regeneration beside the step: the protocol between crafter_step_kernel and crafter_regen_server_kernel (crafter_hip.hip).
// No reference counterpart: Env.reset (env.py:70-81) of an env whose next world the pool does not hold runs CONCURRENTLY
// with the step launch instead of in a kernel behind it.  GPU only (waits between concurrently running kernels); the CPU
// harness keeps the queue-and-kernel form of the same bodies.
#pragma once
#include "env_kernels.hpp"

namespace crafter {

// Memory (crafter_hip.hip, zeroed at create): words[kRegenPushed] tickets handed out, [kRegenClosed] the sequence number of
// the newest launch whose env workgroups have all finished, [kRegenClaimed] tickets the server has taken; ring[N] the
// (ticket + 1) << 32 | env of every ticket; flags[N] seq << 1 | handed-over, written by an env's workgroup as its last
// deed; marks[N] seq, written by the server once the env is regenerated.  No read-modify-write on the common path: 4096
// workgroups counting themselves off on ONE word cost the launch 36 us (cross-XCD atomics on one line: r4v_ab.txt);
// 4096 write-through stores to 4096 words cost nothing.
//
// The end of an env's workgroup.  Common case: one fire-and-forget store.  An env that needs a world the pool does not have:
// its state (store_env has run) is made visible device-wide, it takes a ticket and publishes (ticket, env), and only then
// raises its flag -- whoever has seen every flag of the launch reads a ticket counter that includes every hand-over.
template <class W>
__device__ __forceinline__ void regen_handoff(W& w, const StepCtl& ctl, const Config& cfg, const StatePtrs& st, int env, bool handed) {
  if (handed) {
    __threadfence();
    w.sync();
    if (w.leader()) {
      uint32_t ticket = __hip_atomic_fetch_add(ctl.regen_words + kRegenPushed, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctl.regen_ring + ticket % (uint32_t)cfg.num_envs, ((uint64_t)(ticket + 1u) << 32) | (uint32_t)env,
                         __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (st.pool_stats) w.global_add(st.pool_stats + 1, 1);
      __hip_atomic_store(ctl.regen_flags + env, (ctl.regen_seq << 1) | 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (w.leader()) {
    __hip_atomic_store(ctl.regen_flags + env, ctl.regen_seq << 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

constexpr uint32_t kRegenSpinLimit = 1u << 23;   // polls of ~1 us: a protocol error ends as ST_PIPE_STALL on env 0, not as a hung device
constexpr int kRegenBatch = 16;                  // flags one thread of block 0 has in flight per poll

// Block 0 of the launch (all its threads): leaves when every env workgroup has raised its flag and every env handed over
// carries the server's mark.  Polls are plain loads past the caches (relaxed, agent scope): no fence, no invalidation of
// the L2 the stepping workgroups live in.
template <class W>
__device__ inline void regen_close(W& w, const StepCtl& ctl, const Config& cfg, const StatePtrs& st, uint32_t* lds_word) {
  const uint32_t want = ctl.regen_seq;
  const int n = cfg.num_envs, tid = (int)w.tid();
  if (w.leader()) *lds_word = 0;
  w.sync();
  uint32_t polls = 0;
  bool handed = false, stalled = false;
  for (int base = tid; base < n && !stalled; base += W::kThreads * kRegenBatch) {
    for (;;) {
      uint32_t f[kRegenBatch];
#pragma unroll
      for (int k = 0; k < kRegenBatch; k++) {
        int i = base + W::kThreads * k;
        f[k] = i < n ? __hip_atomic_load(ctl.regen_flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want << 1;
      }
      bool all = true, any = false;
#pragma unroll
      for (int k = 0; k < kRegenBatch; k++) {
        all = all && (f[k] >> 1) == want;
        any = any || (f[k] & 1u) != 0;
      }
      if (all) {
        handed = handed || any;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      if (++polls >= kRegenSpinLimit) {
        stalled = true;
        break;
      }
    }
  }
  if (handed) *lds_word = 1;
  w.sync();
  if (w.leader()) __hip_atomic_store(ctl.regen_words + kRegenClosed, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the server may leave
  if (*lds_word != 0 && !stalled) {   // all but never: some env of this launch was handed over
    for (int i = tid; i < n && !stalled; i += W::kThreads) {
      if ((__hip_atomic_load(ctl.regen_flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) == 0) continue;
      while (__hip_atomic_load(ctl.regen_marks + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
        __builtin_amdgcn_s_sleep(32);
        if (++polls >= kRegenSpinLimit) {
          stalled = true;
          break;
        }
      }
    }
  }
  if (stalled) st.rec[0].status |= ST_PIPE_STALL;
}

// One workgroup of the server: claims tickets while there are any, regenerates those envs (reset_body: world, first frame,
// the next requests to the pool) and marks them; leaves once the launch it serves is closed and no ticket is unclaimed
// (the counters read AFTER "closed" was seen: every hand-over of the launch precedes its closing).
template <class W>
__device__ inline void regen_serve(W& w, uint8_t* smem, const Config& cfg, const TablePtrs& tb, const StatePtrs& st, int gen_parity,
                                   uint8_t* obs, uint32_t* words, const uint64_t* ring, uint32_t* marks, uint32_t seq, int* job) {
  bool closed = false;
  for (;;) {
    if (w.leader()) {
      int got = -1;
      for (uint32_t polls = 0; polls < kRegenSpinLimit; polls++) {
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
          } while ((uint32_t)(entry >> 32) != claimed + 1u && ++polls < kRegenSpinLimit);
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
    if (w.leader()) __hip_atomic_store(marks + env, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace crafter
