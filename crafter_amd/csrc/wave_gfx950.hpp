// Wave policy for gfx950 (CDNA4, 64-lane wavefronts): the lane-parallel primitives the kernel
// bodies (env_core.hpp, render.hpp, worldgen.hpp) are written against.
//
// One workgroup = one environment.  Wave 0 runs the wave-uniform rule code; `ballot`, `lanes`,
// `wave_for`, `lds_add` and `mt_twist` are only called from wave 0 and involve no workgroup
// barrier (a single wave executes its DS instructions in order, so a lane-0 LDS store is seen
// by the following broadcast read).  `block_for` / `sync` / `bcast_from_wave0` are called by
// every wave of the workgroup.
#pragma once
#include <hip/hip_runtime.h>

#include "mt19937.hpp"

namespace crafter {

// MT19937 regeneration, in place, by one wave.  new[i] needs old[i], old[i+1] and element
// i+397 (mod 624) which is OLD for i < 227 and NEW (= new[i-227]) afterwards; so the state is
// regenerated in three batches of <= 227 elements, each batch reading all of its inputs into
// registers before it stores anything, plus the wrap-around element 623.
// Out of line on purpose: the draw helpers are inlined at every call site of the rule code and
// each copy of this body costs ~1 KB of a 64 KB instruction cache (tools/code_size.py).
__device__ __forceinline__ static void mt_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// TEE 1: every new word also goes to tee[i] (global memory: the noise look-ahead leaves each regenerated state in the env's
// scratch, env_kernels.hpp noise_chain -- stores that nothing waits for, instead of a copy pass over the finished state)
template <int TEE>
__device__ __attribute__((noinline)) static void mt_twist_lds_impl(uint32_t* mt, uint32_t* tee) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_assume(__builtin_amdgcn_is_shared(mt));
#endif
  const int l = threadIdx.x & 63;
  uint32_t cur[4], nxt[4], far[4];
  // batch A: i in [0, 227), far = old[i + 397]
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int i = l + 64 * k;
    if (i < 227) {
      cur[k] = mt[i];
      nxt[k] = mt[i + 1];
      far[k] = mt[i + MT_M];
    }
  }
  mt_wave_sync();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int i = l + 64 * k;
    if (i < 227) {
      uint32_t v = mt_twist_word(cur[k], nxt[k], far[k]);
      mt[i] = v;
      if (TEE) tee[i] = v;
    }
  }
  mt_wave_sync();
  // batch B: i in [227, 454), far = new[i - 227]
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int i = 227 + l + 64 * k;
    if (i < 454) {
      cur[k] = mt[i];
      nxt[k] = mt[i + 1];
      far[k] = mt[i - 227];
    }
  }
  mt_wave_sync();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int i = 227 + l + 64 * k;
    if (i < 454) {
      uint32_t v = mt_twist_word(cur[k], nxt[k], far[k]);
      mt[i] = v;
      if (TEE) tee[i] = v;
    }
  }
  mt_wave_sync();
  // batch C: i in [454, 623), far = new[i - 227]
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int i = 454 + l + 64 * k;
    if (i < 623) {
      cur[k] = mt[i];
      nxt[k] = mt[i + 1];
      far[k] = mt[i - 227];
    }
  }
  mt_wave_sync();
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int i = 454 + l + 64 * k;
    if (i < 623) {
      uint32_t v = mt_twist_word(cur[k], nxt[k], far[k]);
      mt[i] = v;
      if (TEE) tee[i] = v;
    }
  }
  mt_wave_sync();
  // element 623: nxt = new[0], far = new[396]
  uint32_t last = mt_twist_word(mt[623], mt[0], mt[396]);
  mt_wave_sync();
  if (l == 0) {
    mt[623] = last;
    if (TEE) tee[623] = last;
  }
  mt_wave_sync();
}
__device__ __forceinline__ static void mt_twist_lds(uint32_t* mt) { mt_twist_lds_impl<0>(mt, nullptr); }


// The regeneration for the noise look-ahead's chain (eleven in a row by one wave while the rule wave works): ONE LDS read
// phase and one write phase per regeneration instead of four of each.  Elements are dealt to the lanes batch by batch --
// lane l, turn k holds i = l + 64 k of batch A [0, 227), i = 227 + l + 64 k of batch B [227, 454), i = 454 + l + 64 k of
// batch C [454, 623) -- so that the NEW word a batch needs from the batch before (index i - 227) is the very value the
// same lane computed in the same turn: the far operands never leave the registers, every OLD word (cur, nxt, batch A's
// far) is read before anything is written, and new[623] takes new[0] and new[396] by v_readlane.
__device__ __attribute__((noinline)) static void mt_twist_chain(uint32_t* mt, uint32_t* tee) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_assume(__builtin_amdgcn_is_shared(mt));
#endif
  const int l = threadIdx.x & 63;
  uint32_t ac[4], an[4], af[4], bc[4], bn[4], cc[3], cn[3];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int i = l + 64 * k;
    int ia = i < 227 ? i : 226;
    ac[k] = mt[ia];
    an[k] = mt[ia + 1];
    af[k] = mt[ia + MT_M];
    int ib = 227 + ia;   // (i < 227 <=> 227 + i < 454)
    bc[k] = mt[ib];
    bn[k] = mt[ib + 1];
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int i = 454 + l + 64 * k;
    int ic = i < 623 ? i : 622;
    cc[k] = mt[ic];
    cn[k] = mt[ic + 1];
  }
  uint32_t old_last = mt[623];
  mt_wave_sync();
  uint32_t a[4], b[4], c[3];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    a[k] = mt_twist_word(ac[k], an[k], af[k]);
    b[k] = mt_twist_word(bc[k], bn[k], a[k]);   // far = new[i - 227]: this lane's batch-A word of the same turn
  }
#pragma unroll
  for (int k = 0; k < 3; k++) c[k] = mt_twist_word(cc[k], cn[k], b[k]);   // far = new[227 + l + 64 k]: this lane's batch-B word
  // new[623] = T(old[623], new[0], new[396]): new[0] = a[0] of lane 0, new[396] = b at 396 - 227 = 169 = lane 41, turn 2
  uint32_t last = mt_twist_word(old_last, (uint32_t)__builtin_amdgcn_readlane((int)a[0], 0), (uint32_t)__builtin_amdgcn_readlane((int)b[2], 41));
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int i = l + 64 * k;
    if (i < 227) {
      mt[i] = a[k];
      mt[227 + i] = b[k];
      tee[i] = a[k];
      tee[227 + i] = b[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int i = 454 + l + 64 * k;
    if (i < 623) {
      mt[i] = c[k];
      tee[i] = c[k];
    }
  }
  if (l == 0) {
    mt[623] = last;
    tee[623] = last;
  }
  mt_wave_sync();
}

// NT = workgroup size, a compile-time constant: with a run-time blockDim the compiler versions every
// block_for loop (stride-1 special cases) and the step kernel no longer fits the instruction cache.
// FRESH 1 (rollout kernel): the thread index is read through a member that refresh() makes a new value as far as the
// optimiser can tell -- see rollout_body (env_kernels.hpp); every other kernel reads the hardware register directly.
template <int NT, int FRESH = 0, int EARLY = 0>   // EARLY 1: the step kernel instance whose frames start before the rules end (kEarlyFrame)
struct WaveGfx950 {
  static_assert(NT % 64 == 0 && NT >= 64, "whole waves");   // NT == 64: single-wave workgroups (world-pool seeding / resolution), never renders
  uint32_t* scratch;  // one LDS dword for workgroup broadcasts
  // Early frame (render.hpp early_frame): the waves behind the first one draw the material half of a day frame while the rule
  // wave is still in the object loop.  One kernel instance only (crafter_step_early_kernel, batches of at least 2048 envs):
  // the code costs every kernel that carries it ~1 % (nine more VGPRs, 8 KB more code) whether it runs or not, and where all
  // envs are resident at once the launch ends with its night frames, which gain nothing (same-box A/Bs, round 6, with the
  // code in every instance: 4096 envs +1.0 %, 1024 envs -2.2 %, 512 envs -1.5 %; as its own instance: 1536 envs -1.0 %,
  // 2048 / 3072 envs +0.3 %, 4096 envs +0.1 ... +1.0 %, 8192 envs +1.2 %).
  static constexpr bool kEarlyFrame = EARLY != 0 && NT > 64;
  static constexpr bool kConcurrentWaves = true;          // the workgroup's waves really run side by side (the CPU harness plays them one after the other)
  static constexpr int kDrawingWaves = NT > 64 ? NT / 64 - 1 : 1;
  // waits until the LDS word *p holds at least `value` (another wave of the workgroup stores / adds to it)
  __device__ __forceinline__ void spin_until(const uint32_t* p, uint32_t value) const {
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < value) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  uint32_t tid_ = threadIdx.x;
  __device__ __forceinline__ uint32_t tx() const {
    if constexpr (FRESH != 0) {
      __builtin_assume(tid_ < (uint32_t)NT);
      return tid_;
    } else {
      return threadIdx.x;
    }
  }
  __device__ __forceinline__ void refresh() {
    if constexpr (FRESH != 0) asm volatile("" : "+v"(tid_));
  }
  // promise that p points into LDS (lets InferAddressSpaces turn flat accesses into ds_*)
  __device__ __forceinline__ static void assume_lds(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the builtin only exists in the device pass of hipcc
    __builtin_assume(__builtin_amdgcn_is_shared(p));
#else
    (void)p;
#endif
  }

  // IEEE-754 correctly rounded float division, whatever the compiler's fast-division defaults are
  __device__ __forceinline__ static float fdiv(float a, float b) { return __fdiv_rn(a, b); }
  // v is the same in every lane of the wave (the compiler cannot always prove it): keep it in an SGPR so that
  // branches on it are scalar branches instead of exec-mask regions
  __device__ __forceinline__ static int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
  // a * b for operands known to fit 24 bits: full-rate v_mul_u32_u24 (v_mul_lo_u32 is quarter rate)
  __device__ __forceinline__ static int mul24(int a, int b) { return __mul24(a, b); }
  // (a * b) >> 32 for operands known to fit 24 bits: one full-rate v_mul_hi_u32_u24 (the compiler only picks it when it
  // can prove the ranges; v_mul_hi_u32 is quarter rate)
  __device__ __forceinline__ static uint32_t mulhi24(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  }

  // v, as far as the optimiser can tell a new wave-uniform value: what is computed from it inside a loop stays inside the
  // loop (a rollout's steps must not share hoisted addresses and table values: they cost the step kernel its registers)
  __device__ __forceinline__ static int opaque(int v) {
    asm volatile("" : "+s"(v));
    return v;
  }
  __device__ __forceinline__ int tid() const { return tx(); }
  __device__ __forceinline__ int nthreads() const { return NT; }
  // barrier-free per-thread code that depends on the workgroup's shape: f(thread index), for each of kThreads threads
  static constexpr int kThreads = NT;
  template <class F>
  __device__ __forceinline__ void each_thread(F f) const { f((int)tx()); }
  // per-thread values that live from one each_thread pass to the next: T v[kThreadSlots], indexed by thread_slot(tid)
  // (registers here; the CPU harness keeps one element per virtual thread)
  static constexpr int kThreadSlots = 1;
  __device__ __forceinline__ static int thread_slot(int) { return 0; }
  __device__ __forceinline__ int lane() const { return tx() & 63; }
  __device__ __forceinline__ bool leader() const { return tx() == 0; }
  __device__ __forceinline__ bool wave0() const { return tx() < 64; }
  // wave k of the workgroup (ballot / lanes work in any wave); lets independent wave-level jobs run side by side
  __device__ __forceinline__ bool wave_is(int k) const { return (int)(tx() >> 6) == k; }
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  // A barrier for data exchanged through LDS ONLY (per-frame tables, MT19937 state buffers): what the waves' global loads
  // and stores are doing is none of its business.  sync() carries a workgroup-scope release fence, and where global stores
  // or loads are in flight that is a wait for every one of them -- a night frame that keeps its pixels in global scratch
  // paid a store round trip per epoch, the inventory texels fetched ahead of the frame tables were waited for at the
  // tables' first barrier (r4c: frame group 53 k clocks per night frame).  A wave's DS instructions execute in order, so
  // "all my LDS accesses are done" is lgkmcnt(0).
  __device__ __forceinline__ void sync_lds() const { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
  // an opaque point in the instruction stream: code on either side is not merged across it (env_core.hpp mat_at_uniform)
  __device__ __forceinline__ static void keep_apart() { asm volatile("" ::: "memory"); }
  // orders this wave's LDS traffic for the compiler; the hardware already keeps it in order
  __device__ __forceinline__ void wsync() const {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // 64-bit mask of pred(base + lane) over the lanes with base + lane < n
  template <class F>
  __device__ __forceinline__ uint64_t ballot(int base, int n, F pred) const {
    int i = base + lane();
    bool p = false;
    if (i < n) p = pred(i);
    return __ballot(p);
  }
  // f(i, lane) on lane = i - base for base <= i < min(base + 64, n); all lanes in lock-step
  template <class F>
  __device__ __forceinline__ void lanes(int base, int n, F f) const {
    int i = base + lane();
    if (i < n) f(i, lane());
  }
  template <class F>
  __device__ __forceinline__ void wave_for(int n, F f) const {
#pragma clang loop unroll(disable)
    for (int i = lane(); i < n; i += 64) f(i);
  }
  template <class F>
  __device__ __forceinline__ void block_for(int n, F f) const {
#pragma clang loop unroll(disable)
    for (int i = tx(); i < n; i += NT) f(i);
  }
  // Per-lane scratch registers that survive between primitives, so a multi-step lane-parallel
  // round (speculate -> ballot -> commit) never has to bounce its lane state through LDS.
  uint32_t lv[20];  // (10 .. 19: only the scan and the despawn pass of the instance whose slot table stays in global memory -- env_core.hpp FarSlot) 8, 9: the balance pass's census look-ahead; 7: the balance pass's pair flags; 0, 1, 3: round state of lane-parallel algorithms; 2: tempered RNG look-ahead (env_core.hpp); 4, 5, 6: the
                    // balance pass's hit list (env_core.hpp balance): lane h = the h-th (chunk, class) pair that draws a spawn / despawn
  template <class F>
  __device__ __forceinline__ void lane_set(int slot, int base, int n, F f) {
    int i = base + lane();
    uint32_t v = 0;
    if (i < n) v = f(i, lane());
    lv[slot] = v;
  }
  __device__ __forceinline__ uint32_t lane_get(int slot, int /*lane*/) const { return lv[slot]; }      // own lane, inside lambdas
  // ... by every lane of the wave, unpredicated (f may fetch across lanes: lane_fetch)
  template <class F>
  __device__ __forceinline__ void lane_set2_all(int slot_lo, int slot_hi, F f) {
    uint64_t v = f((int)lane());
    lv[slot_lo] = (uint32_t)v;
    lv[slot_hi] = (uint32_t)(v >> 32);
  }
  // register `slot` of lane `from` (per-lane; 0 <= from < 64), asked for by lane l: one ds_bpermute_b32 (the LDS crossbar, no LDS memory)
  __device__ __forceinline__ uint32_t lane_fetch(int slot, int from, int /*l*/) const {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)lv[slot]);
  }
  // acc + the number of set bits of the wave-uniform mask m below lane l (the caller's own lane): v_mbcnt_lo / v_mbcnt_hi
  __device__ __forceinline__ int count_below(uint64_t m, int /*l*/, int acc) const {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)acc));
  }
  // four registers s0 .. s0 + 3 at once: f returns a 16-byte vector (one dwordx4 load)
  template <int S0, class F>
  __device__ __forceinline__ void lane_set4(int base, int n, F f) {
    int i = base + lane();
    typedef uint32_t v4 __attribute__((vector_size(16)));
    v4 v = {0u, 0u, 0u, 0u};
    if (i < n) v = f(i, lane());
    lv[S0] = v[0];
    lv[S0 + 1] = v[1];
    lv[S0 + 2] = v[2];
    lv[S0 + 3] = v[3];
  }
  // K registers R0 .. R0 + K - 1: register R0 + k of a lane = map(load(i), i) for i = base + 64 k + lane.  The K loads are
  // UNPREDICATED (indices clamped to n - 1) and all issued before the first map: one memory round trip for K x 64 elements.
  // A lane whose index is beyond n - 1 gets `beyond`.
  template <int R0, int K, class F, class G>
  __device__ __forceinline__ void lane_gather_map(int base, int n, uint32_t beyond, F load, G map) {
    decltype(load(0)) v[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      int i = base + 64 * k + (int)lane();
      v[k] = load(i < n ? i : n - 1);
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      int i = base + 64 * k + (int)lane();
      lv[R0 + k] = i < n ? (uint32_t)map(v[k], i) : beyond;
    }
  }
  // two registers at once: f returns 64 bits, the low half goes to slot_lo, the high half to slot_hi
  template <class F>
  __device__ __forceinline__ void lane_set2(int slot_lo, int slot_hi, int base, int n, F f) {
    int i = base + lane();
    uint64_t v = 0;
    if (i < n) v = f(i, lane());
    lv[slot_lo] = (uint32_t)v;
    lv[slot_hi] = (uint32_t)(v >> 32);
  }
  // f(i) for i = lane, lane + 64, lane + 128 into registers 0, 1, 3 -- UNPREDICATED (indices clamped to n - 1), so that when f
  // loads from memory the three loads are in flight together instead of one round trip after the other
  template <class F>
  __device__ __forceinline__ void lane_gather3(int n, F f) {
    int l = lane();
    int i0 = l < n ? l : n - 1, i1 = l + 64 < n ? l + 64 : n - 1, i2 = l + 128 < n ? l + 128 : n - 1;
    uint32_t a = f(i0), b = f(i1), c = f(i2);
    lv[0] = a;
    lv[1] = b;
    lv[3] = c;
  }
  // ballot of (register `slot` == value) over the lanes with base + lane < n
  __device__ __forceinline__ uint64_t lane_match(int slot, int base, int n, uint32_t value) const {
    return __ballot(base + lane() < n && lv[slot] == value);
  }
  // lane l's register := v, for wave-uniform l and v (v_writelane: serial scalar code builds a lane register word by word)
  __device__ __forceinline__ void lane_put(int slot, int l, uint32_t v) {
    lv[slot] = ((int)(tx() & 63) == l) ? v : lv[slot];   // v_cmp + v_cndmask with scalar operands
  }
  // wave-uniform 64-bit value the compiler cannot prove uniform: keep it in an SGPR pair
  __device__ __forceinline__ static uint64_t uni64(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  }
  __device__ __forceinline__ uint32_t lane_read(int slot, int l) const { return __builtin_amdgcn_readlane(lv[slot], l); }  // uniform l

  // Occupancy registers (env_core.hpp, LaneSlots): lane l of group g holds the packed position x | y << 16 of object slot
  // 64 g + l, 0xFFFFFFFF when the slot holds nothing.  "Which object stands on this cell" is then a compare + ballot over
  // the registers (a few instructions, no memory) instead of a cell -> slot map in LDS: 4 KB less per environment.
  static constexpr int kOccGroups = 4;   // 256 slots
  uint32_t occ[kOccGroups];
  template <class F>
  __device__ __forceinline__ void occ_fill(int n, F f) {   // f(i) for i < n, nothing beyond
#pragma unroll
    for (int g = 0; g < kOccGroups; g++) {
      int i = 64 * g + lane();
      uint32_t v = 0xFFFFFFFFu;
      if (i < n) v = f(i);
      occ[g] = v;
    }
  }
  // slot whose register equals key among slots < n (n wave-uniform), -1 if none
  __device__ __forceinline__ int occ_find(uint32_t key, int n) const {
#pragma unroll
    for (int g = 0; g < kOccGroups; g++) {
      if (uni(n) > 64 * g) {
        uint64_t m = __ballot(occ[g] == key);
        if (m) return 64 * g + __builtin_ctzll(m);
      }
    }
    return -1;
  }
  __device__ __forceinline__ void occ_put(int slot, uint32_t key) {   // wave-uniform slot, key
#pragma unroll
    for (int g = 0; g < kOccGroups; g++) occ[g] = (64 * g + (int)(tx() & 63) == slot) ? key : occ[g];
  }
  // f(g, ballot of pred(packed position) over the registers of group g) for every group that holds a slot < n.  The
  // groups are walked by an unrolled loop: a register array indexed by a run-time value would be put in scratch memory,
  // and the whole wave object with it.
  template <class P, class F>
  __device__ __forceinline__ void occ_groups(int n, P pred, F f) const {
#pragma unroll
    for (int g = 0; g < kOccGroups; g++)
      if (uni(n) > 64 * g) f(g, __ballot(pred(occ[g])));
  }
  // a word another wave of this workgroup has stored to global memory (after drain_stores + a barrier): straight from L2
  __device__ __forceinline__ static uint32_t load_fresh(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // Words that CONCURRENT kernels exchange (the world pool's header words: the step kernel on the launch stream, the
  // generation kernels on side streams, on other XCDs): relaxed agent-scope atomics, so that every access is one
  // indivisible access of the whole word at the device's coherence point and never a cached, torn or elided one.
  // (Leader-only and rare: free.  The protocol on top of them is one writer per word at a time -- env_kernels.hpp.)
  template <class T>
  __device__ __forceinline__ static T agent_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  template <class T>
  __device__ __forceinline__ static void agent_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  // every store this wave has issued has reached L2 (orders two passes of stores to the same addresses by different lanes)
  __device__ __forceinline__ static void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __device__ __forceinline__ uint64_t lane_ballot(int slot, uint32_t mask) const { return __ballot((lv[slot] & mask) != 0); }

  // index of the k-th (0-based) set bit of a wave-uniform mask: each lane ranks itself among the set
  // bits below it, the lane whose rank is k answers (instead of k serial clear-lowest-bit steps)
  __device__ __forceinline__ int kth_set(uint64_t m, int k) const {
    int l = lane();
    bool mine = ((m >> l) & 1ull) && __builtin_popcountll(m & ((1ull << l) - 1ull)) == k;
    return __builtin_ctzll(__ballot(mine));
  }
  __device__ __forceinline__ void lds_add(int32_t* p, int v) const { atomicAdd(p, v); }
  __device__ __forceinline__ void lds_or(uint32_t* p, uint32_t v) const { atomicOr(p, v); }
  __device__ __forceinline__ uint32_t lds_inc(uint32_t* p) const { return atomicAdd(p, 1u); }   // returns the old value
  __device__ __forceinline__ uint32_t lds_fetch_add(uint32_t* p, uint32_t v) const { return atomicAdd(p, v); }
  __device__ __forceinline__ int wave_index() const { return (int)(tx() >> 6); }
  __device__ __forceinline__ static constexpr int num_waves() { return NT / 64; }
  // producer / consumer split of a workgroup: wave 0 produces, the other waves consume (a
  // single-wave workgroup does both, one after the other)
  __device__ __forceinline__ bool producer() const { return tx() < 64; }
  // A lane's share of a <= 312-item epoch as (first index, stride); false if the lane only produces.
  static constexpr int kEpochSlots = NT > 64 ? (312 + NT - 64 - 1) / (NT > 64 ? NT - 64 : 1) : 312;   // pixels of one epoch per consumer lane
  __device__ __forceinline__ bool consumer_slot(bool split, int& first, int& stride) const {
    if (split) {
      first = (int)tx() - 64;
      stride = NT - 64;
      return tx() >= 64;
    }
    first = tx();
    stride = NT;
    return true;
  }
  template <class F>
  __device__ __forceinline__ void consumer_for(int n, F f) const {
    if (tx() >= 64)
#pragma clang loop unroll(disable)
      for (int i = tx() - 64; i < n; i += NT - 64) f(i);
  }
  // f() in the waves behind the first one only, as a wave-level (scalar) branch: the first wave does not so much as wait
  // for what f's condition looks at
  template <class F>
  __device__ __forceinline__ void consumers(F f) const {
    if (uni((int)(tx() >= 64))) f();
  }
  __device__ __forceinline__ int global_add(int32_t* p, int v) const { return atomicAdd(p, v); }
  // wave issue priority (0..3): the latency-critical step kernel outranks background generation
  // waves that share its SIMDs
  __device__ __forceinline__ static void set_priority_high() { __builtin_amdgcn_s_setprio(3); }
  __device__ __forceinline__ static void set_priority_mid() { __builtin_amdgcn_s_setprio(1); }
  // (a launch parameter: s_setprio takes an immediate)
  __device__ __forceinline__ static void set_priority(int p) {
    if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p >= 3) __builtin_amdgcn_s_setprio(3);
  }
  __device__ __forceinline__ uint64_t clock() const { return __builtin_readcyclecounter(); }

  __device__ __forceinline__ uint32_t bcast_from_wave0(uint32_t v) const {
    if (tx() == 0) *scratch = v;
    sync();
    uint32_t r = *scratch;
    sync();
    return r;
  }

  // dst = regenerated src (out of place; src stays readable for the other waves meanwhile)
  // Every batch reads all its words first (clamped indices, unconditional) and only then writes: written as
  // read-compute-write per element the compiler keeps the reads behind the previous element's store (the buffers may
  // alias as far as it knows) and the wave pays one LDS round trip per 64 words instead of one per batch.
  __device__ __forceinline__ void mt_twist_from(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) const {
    const int l = lane();
    uint32_t cur[4], nxt[4], far[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {   // i in [0, 227): far element is old
      int i = l + 64 * k;
      int ii = i < 227 ? i : 226;
      cur[k] = src[ii];
      nxt[k] = src[ii + 1];
      far[k] = src[ii + MT_M];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int i = l + 64 * k;
      if (i < 227) dst[i] = mt_twist_word(cur[k], nxt[k], far[k]);
    }
    wsync();
#pragma unroll
    for (int k = 0; k < 4; k++) {   // i in [227, 454): far = new[i - 227]
      int i = 227 + l + 64 * k;
      int ii = i < 454 ? i : 453;
      cur[k] = src[ii];
      nxt[k] = src[ii + 1];
      far[k] = dst[ii - 227];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int i = 227 + l + 64 * k;
      if (i < 454) dst[i] = mt_twist_word(cur[k], nxt[k], far[k]);
    }
    wsync();
#pragma unroll
    for (int k = 0; k < 3; k++) {   // i in [454, 623)
      int i = 454 + l + 64 * k;
      int ii = i < 623 ? i : 622;
      cur[k] = src[ii];
      nxt[k] = src[ii + 1];
      far[k] = dst[ii - 227];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      int i = 454 + l + 64 * k;
      if (i < 623) dst[i] = mt_twist_word(cur[k], nxt[k], far[k]);
    }
    wsync();
    uint32_t last = mt_twist_word(src[623], dst[0], dst[396]);
    if (l == 0) dst[623] = last;
    wsync();
  }

  __device__ __forceinline__ void mt_twist(uint32_t* mt) const { mt_twist_lds(mt); }
  // ... the same, every new word also stored to tee[0 .. 623]
  __device__ __forceinline__ void mt_twist_tee(uint32_t* mt, uint32_t* tee) const { mt_twist_chain(mt, tee); }
};

}  // namespace crafter
