// The dispatch order of a step launch (StepCtl::order): shared by the step kernel (crafter_hip.hip) and the pipelined step
// kernel (crafter_pipe.hip).  Device code only (one 256-thread workgroup).
#pragma once
#include <hip/hip_runtime.h>

#include "env_kernels.hpp"

namespace crafter {

constexpr int kOrderThreads = 256;   // = the step kernels' workgroup

// The dispatch order of the step launch after this one (StepCtl::order_build): the envs whose step will draw a night frame or
// balance the chunks -- about a quarter of them, twice as long as a plain day step -- from the front, the others from the
// back.  next_step[env] = the step number the env executes in the launch now running (left there by the launch before), so
// the launch after this one runs step next_step[env] + 1 unless the env resets in between (then it is misfiled: harmless).
// One workgroup: per thread a bit mask of its envs (env = thread + k * 256), a block-wide exclusive sum, one store per env.
// Launches of several steps per env (crafter_step_n's rollout kernel): the launch now running executes `ahead` steps of every
// env, the one after it `horizon`: an env is slow there if its steps reach into the night (a night lasts 125 steps, so the
// two ends of the stretch tell; every stretch of ten steps or more has its balance step).
__device__ __forceinline__ void build_order(const Config& cfg, const TablePtrs& tb, int32_t* __restrict__ order,
                                   const int32_t* __restrict__ next_step, uint32_t* lds, int ahead = 1, int horizon = 1) {
  const int n = cfg.num_envs, tid = (int)threadIdx.x;
  constexpr int NT = kOrderThreads;
  uint64_t slow_bits = 0;   // bit k: env tid + k * NT is slow (n <= 64 * NT, the caller's condition)
  int n_slow = 0, n_mine = 0;
  for (int k = 0, env = tid; env < n; env += NT, k++) {
    int s = next_step[env] + ahead;
    if (s < 0) s = 0;
    if (s >= cfg.n_daylight) s = cfg.n_daylight - 1;
    bool slow;
    if (horizon <= 1) {
      slow = (s % 10 == 0) || tb.daylight[s] < 0.5;
    } else {
      int s2 = s + horizon - 1;
      if (s2 >= cfg.n_daylight) s2 = cfg.n_daylight - 1;
      slow = tb.daylight[s] < 0.5 || tb.daylight[s2] < 0.5;
    }
    slow_bits |= (uint64_t)slow << k;
    n_slow += slow;
    n_mine++;
  }
  // exclusive sums over the threads of the (slow, fast) counts, both packed into one word: Hillis-Steele in LDS
  uint32_t v = (uint32_t)n_slow | ((uint32_t)(n_mine - n_slow) << 16);
  lds[tid] = v;
  __syncthreads();
  for (int d = 1; d < NT; d <<= 1) {
    uint32_t add = tid >= d ? lds[tid - d] : 0u;
    __syncthreads();
    lds[tid] += add;
    __syncthreads();
  }
  uint32_t before = lds[tid] - v;
  int at_slow = (int)(before & 0xFFFFu), at_fast = (int)(before >> 16);
  for (int k = 0, env = tid; env < n; env += NT, k++) {
    if ((slow_bits >> k) & 1ull)
      order[at_slow++] = env;
    else
      order[n - 1 - at_fast++] = env;
  }
}


}  // namespace crafter
