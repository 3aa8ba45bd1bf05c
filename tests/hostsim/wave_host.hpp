// TEST INFRASTRUCTURE -- not part of the product, never loaded by crafter_amd.
//
// Serial stand-in for the gfx950 wave policy (crafter_amd/csrc/wave_gfx950.hpp) so that the
// per-environment kernel bodies (env_kernels.hpp) can be executed and debugged on a machine
// without a GPU: one "wave" = one host thread that plays the 64 lanes one after another.
// Compiled by tests/hostsim/build.py with  g++ -D__device__= -D__host__= -ffp-contract=off.
#pragma once
#include <stdint.h>

struct uint4 {
  uint32_t x, y, z, w;
};

namespace crafter {

struct WaveHost {
  uint32_t* scratch = nullptr;
  static constexpr bool kEarlyFrame = true;
  static constexpr bool kConcurrentWaves = false;   // one host thread plays the waves one after the other: what a device wave waits for has happened already
  static constexpr int kDrawingWaves = 1;
  void spin_until(const uint32_t*, uint32_t) const {}
  static void assume_lds(const void*) {}
  static float fdiv(float a, float b) { return a / b; }
  static int mul24(int a, int b) { return a * b; }
  static uint32_t mulhi24(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
  static int uni(int v) { return v; }
  static int opaque(int v) { return v; }
  void refresh() {}
  int tid() const { return 0; }
  int nthreads() const { return 1; }
  // per-thread code written against the device workgroup's shape runs once per virtual thread
  static constexpr int kThreads = 256;
  template <class F>
  void each_thread(F f) const {
    for (int t = 0; t < kThreads; t++) f(t);
  }
  static constexpr int kThreadSlots = kThreads;
  static int thread_slot(int tid) { return tid; }
  int lane() const { return 0; }
  bool leader() const { return true; }
  bool wave0() const { return true; }
  bool wave_is(int) const { return true; }
  // compiler barriers: the device versions order memory, and the kernel bodies rely on that between passes that
  // access the same LDS bytes through different types
  void sync() const { asm volatile("" ::: "memory"); }
  void wsync() const { asm volatile("" ::: "memory"); }
  void sync_lds() const { asm volatile("" ::: "memory"); }
  template <class F>
  uint64_t ballot(int base, int n, F pred) const {
    uint64_t m = 0;
    for (int lane = 0; lane < 64; lane++) {
      int i = base + lane;
      if (i < n && pred(i)) m |= 1ull << lane;
    }
    return m;
  }
  template <class F>
  void lanes(int base, int n, F f) const {
    for (int lane = 0; lane < 64; lane++) {
      int i = base + lane;
      if (i < n) f(i, lane);
    }
  }
  template <class F>
  void wave_for(int n, F f) const {
    for (int i = 0; i < n; i++) f(i);
  }
  template <class F>
  void block_for(int n, F f) const {
    for (int i = 0; i < n; i++) f(i);
  }
  uint32_t lv[20][64] = {};
  template <int S0, class F>
  void lane_set4(int base, int n, F f) {
    for (int lane = 0; lane < 64; lane++) {
      int i = base + lane;
      for (int k = 0; k < 4; k++) lv[S0 + k][lane] = 0u;
      if (i < n) {
        auto v = f(i, lane);
        for (int k = 0; k < 4; k++) lv[S0 + k][lane] = v[k];
      }
    }
  }
  template <int R0, int K, class F, class G>
  void lane_gather_map(int base, int n, uint32_t beyond, F load, G map) {
    for (int k = 0; k < K; k++)
      for (int lane = 0; lane < 64; lane++) {
        int i = base + 64 * k + lane;
        lv[R0 + k][lane] = i < n ? (uint32_t)map(load(i), i) : beyond;
      }
  }
  template <class F>
  void lane_set(int slot, int base, int n, F f) {
    for (int lane = 0; lane < 64; lane++) {
      int i = base + lane;
      lv[slot][lane] = (i < n) ? (uint32_t)f(i, lane) : 0u;
    }
  }
  uint32_t lane_get(int slot, int lane) const { return lv[slot][lane]; }
  template <class F>
  void lane_set2_all(int slot_lo, int slot_hi, F f) {
    for (int lane = 0; lane < 64; lane++) {
      uint64_t v = (uint64_t)f(lane);
      lv[slot_lo][lane] = (uint32_t)v;
      lv[slot_hi][lane] = (uint32_t)(v >> 32);
    }
  }
  uint32_t lane_fetch(int slot, int from, int) const { return lv[slot][from & 63]; }
  int count_below(uint64_t m, int l, int acc) const { return acc + __builtin_popcountll(m & ((1ull << l) - 1ull)); }
  template <class F>
  void lane_set2(int slot_lo, int slot_hi, int base, int n, F f) {
    for (int lane = 0; lane < 64; lane++) {
      int i = base + lane;
      uint64_t v = (i < n) ? (uint64_t)f(i, lane) : 0ull;
      lv[slot_lo][lane] = (uint32_t)v;
      lv[slot_hi][lane] = (uint32_t)(v >> 32);
    }
  }
  template <class F>
  void lane_gather3(int n, F f) {
    for (int lane = 0; lane < 64; lane++) {
      lv[0][lane] = (uint32_t)f(lane < n ? lane : n - 1);
      lv[1][lane] = (uint32_t)f(lane + 64 < n ? lane + 64 : n - 1);
      lv[3][lane] = (uint32_t)f(lane + 128 < n ? lane + 128 : n - 1);
    }
  }
  uint64_t lane_match(int slot, int base, int n, uint32_t value) const {
    uint64_t m = 0;
    for (int lane = 0; lane < 64; lane++)
      if (base + lane < n && lv[slot][lane] == value) m |= 1ull << lane;
    return m;
  }
  uint32_t lane_read(int slot, int l) const { return lv[slot][l]; }
  void lane_put(int slot, int l, uint32_t v) { lv[slot][l] = v; }
  static uint64_t uni64(uint64_t v) { return v; }
  static constexpr int kOccGroups = 4;
  uint32_t occ[kOccGroups * 64];
  template <class F>
  void occ_fill(int n, F f) {
    for (int i = 0; i < kOccGroups * 64; i++) occ[i] = i < n ? (uint32_t)f(i) : 0xFFFFFFFFu;
  }
  int occ_find(uint32_t key, int n) const {
    for (int i = 0; i < kOccGroups * 64 && i < (n + 63) / 64 * 64; i++)   // whole groups, like the device
      if (occ[i] == key) return i;
    return -1;
  }
  template <class P, class F>
  void occ_groups(int n, P pred, F f) const {
    for (int g = 0; g < kOccGroups && 64 * g < n; g++) {
      uint64_t m = 0;
      for (int lane = 0; lane < 64; lane++)
        if (pred(occ[64 * g + lane])) m |= 1ull << lane;
      f(g, m);
    }
  }
  static void drain_stores() {}
  static void keep_apart() {}
  template <class T>
  static T agent_load(const T* p) { return *p; }
  template <class T>
  static void agent_store(T* p, T v) { *p = v; }
  static uint32_t load_fresh(const uint32_t* p) { return *p; }
  void occ_put(int slot, uint32_t key) {
    if (slot >= 0 && slot < kOccGroups * 64) occ[slot] = key;
  }
  uint64_t lane_ballot(int slot, uint32_t mask) const {
    uint64_t m = 0;
    for (int lane = 0; lane < 64; lane++)
      if (lv[slot][lane] & mask) m |= 1ull << lane;
    return m;
  }
  int kth_set(uint64_t m, int k) const {
    for (int i = 0; i < k; i++) m &= m - 1;
    return __builtin_ctzll(m);
  }
  void lds_add(int32_t* p, int v) const { *p += v; }
  void lds_or(uint32_t* p, uint32_t v) const { *p |= v; }
  uint32_t lds_inc(uint32_t* p) const { return (*p)++; }
  uint32_t lds_fetch_add(uint32_t* p, uint32_t v) const { uint32_t old = *p; *p += v; return old; }
  int wave_index() const { return 0; }
  static constexpr int num_waves() { return 1; }
  template <class F>
  void consumers(F f) const { f(); }
  bool producer() const { return true; }
  static constexpr int kEpochSlots = 312;
  bool consumer_slot(bool, int& first, int& stride) const {
    first = 0;
    stride = 1;
    return true;
  }
  template <class F>
  void consumer_for(int n, F f) const {
    for (int i = 0; i < n; i++) f(i);
  }
  void mt_twist_from(const uint32_t* src, uint32_t* dst) const {
    const int N = 624, M = 397;
    for (int i = 0; i < N; i++) {
      uint32_t nxt = (i + 1 < N) ? src[i + 1] : dst[0];
      uint32_t far = (i + M < N) ? src[i + M] : dst[i + M - N];
      uint32_t y = (src[i] & 0x80000000u) | (nxt & 0x7fffffffu);
      dst[i] = far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
  }
  int global_add(int32_t* p, int v) const { int old = *p; *p += v; return old; }
  static void set_priority_high() {}
  static void set_priority_mid() {}
  uint64_t clock() const { return 0; }
  uint32_t bcast_from_wave0(uint32_t v) const { return v; }
  void mt_twist_tee(uint32_t* mt, uint32_t* tee) const {
    mt_twist(mt);
    for (int i = 0; i < 624; i++) tee[i] = mt[i];
  }
  // textbook in-place twist (genrand_int32's regeneration loop)
  void mt_twist(uint32_t* mt) const {
    const int N = 624, M = 397;
    for (int i = 0; i < N; i++) {
      uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % N] & 0x7fffffffu);
      mt[i] = mt[(i + M) % N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
  }
};

}  // namespace crafter
