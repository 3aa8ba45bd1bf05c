// The kernels of crafter_step_n (open-loop rollouts): see crafter_rollout.hpp for why they are a translation unit of
// their own, env_kernels.hpp rollout_body / requeue_rollout_body for what they do.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "crafter_rollout.hpp"
#include "dispatch_order.hpp"
#include "wave_gfx950.hpp"

namespace crafter {
namespace {

constexpr int kStepThreads = 256;      // = crafter_hip.hip (checked by the launchers' callers through cfg.step_threads)
constexpr int kRequeueThreads = 256;

// T steps of env blockIdx.x in one launch
// (the default geometry's instances: six waves per SIMD -- 80 VGPRs -- are what their 26.9 KB of LDS allow per CU; left to
// itself the register allocator takes 83 and loses the sixth workgroup.  The generic instances need what they need.)
#define CRAFTER_ROLLOUT_BOUNDS __launch_bounds__(kStepThreads, GEO == 1 ? 6 : LM == 0 ? (GEO == 2 ? 6 : 4) : 1)   // (GEO 2: the default view on a world of any size, crafter_hip.hip)
template <int LM, int GEO, int RUL>
__global__ void CRAFTER_ROLLOUT_BOUNDS
crafter_rollout_kernel(Config cfg_in, TablePtrs tb, StatePtrs st, const int32_t* __restrict__ actions,
                       uint8_t* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done,
                       StepCtl ctl, RolloutArgs ra) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  WaveGfx950<kStepThreads, 1> w;
  const Config cfg = GEO == 1 ? with_default_geometry(cfg_in) : GEO == 2 ? with_default_view(cfg_in) : cfg_in;
  int env = (int)blockIdx.x;
  if (ctl.order_build) {   // dispatch order, as crafter_step_kernel keeps it: block 0 sorts for the launch after this one -- the envs
                           // whose next stretch reaches into the night first: their sixteen steps take half as long again
    if (env == 0) {
      build_order(cfg, tb, ctl.order_build, ctl.next_step, (uint32_t*)smem, ra.T, ra.T);
      return;
    }
    env -= 1;
    if (ctl.order) env = ctl.order[env];
  }
  if constexpr (GEO == 1)   // max_objects == 256: one-byte slot ids, as in crafter_step_kernel
    rollout_body<WaveGfx950<kStepThreads, 1>, LM, RUL, uint8_t>(w, smem, env, cfg, tb, st, actions, obs, reward, done, ctl,
                                                                ra.T, ra.obs_stride, ra.stalled_at);
  else
    rollout_body<WaveGfx950<kStepThreads, 1>, LM, RUL, typename StepSlot<LM>::type>(w, smem, env, cfg, tb, st, actions, obs, reward, done, ctl,
                                                                 ra.T, ra.obs_stride, ra.stalled_at);
}

// ... and the regeneration kernel behind it: the envs that stopped for want of a world (all but never any).  Bounded to four
// waves per SIMD: with the 256 VGPRs the allocator takes when left alone a workgroup needs half a CU's register file, and
// while a generation batch is resident (always: the batch of the stretch before runs beside this one) it waited 50 us on
// average for that much to come free -- to find its queue empty (profiles/r5_rollout_profile.json: min 4.4 us).
__global__ void __launch_bounds__(kRequeueThreads, 4)
crafter_requeue_rollout_kernel(Config cfg, TablePtrs tb, StatePtrs st, const int32_t* __restrict__ actions, uint8_t* __restrict__ obs,
                               float* __restrict__ reward, uint8_t* __restrict__ done, StepCtl ctl, RolloutArgs ra) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int32_t* q = st.reset_q + (size_t)ctl.parity * (cfg.num_envs + 4);
  int count = q[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) st.reset_q[(size_t)(1 - ctl.parity) * (cfg.num_envs + 4)] = 0;
  for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
    WaveGfx950<kRequeueThreads> w;
    requeue_rollout_body(w, smem, q[4 + k], cfg, tb, st, actions, obs, reward, done, ctl, ra.T, ra.obs_stride, ra.stalled_at);
    __syncthreads();
  }
}

#define CRAFTER_LAUNCH(kernel, grid, block, lds, stream, start, stop, ...)                                          \
  do {                                                                                                              \
    if ((start) != nullptr || (stop) != nullptr)                                                                    \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, start, stop, 0, __VA_ARGS__);                         \
    else                                                                                                            \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                            \
  } while (0)

}  // namespace

void launch_rollout(int instance, int num_envs, size_t lds, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const Config& cfg,
                    const TablePtrs& tb, const StatePtrs& st, const int32_t* actions, uint8_t* obs, float* reward, uint8_t* done,
                    const StepCtl& ctl, const RolloutArgs& ra) {
  dim3 grid(num_envs + (ctl.order_build ? 1 : 0)), block(kStepThreads);
  if (instance == 7)
    CRAFTER_LAUNCH((crafter_rollout_kernel<1, 1, 1>), grid, block, lds, stream, start, stop, cfg, tb, st, actions, obs, reward, done, ctl, ra);
  else if (instance == 6)
    CRAFTER_LAUNCH((crafter_rollout_kernel<1, 1, 0>), grid, block, lds, stream, start, stop, cfg, tb, st, actions, obs, reward, done, ctl, ra);
  else if (instance == 9)
    CRAFTER_LAUNCH((crafter_rollout_kernel<0, 2, 1>), grid, block, lds, stream, start, stop, cfg, tb, st, actions, obs, reward, done, ctl, ra);
  else if (instance == 4)
    CRAFTER_LAUNCH((crafter_rollout_kernel<1, 0, 0>), grid, block, lds, stream, start, stop, cfg, tb, st, actions, obs, reward, done, ctl, ra);
  else
    CRAFTER_LAUNCH((crafter_rollout_kernel<0, 0, 0>), grid, block, lds, stream, start, stop, cfg, tb, st, actions, obs, reward, done, ctl, ra);
}

void launch_requeue_rollout(int grid, size_t lds, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const Config& cfg,
                            const TablePtrs& tb, const StatePtrs& st, const int32_t* actions, uint8_t* obs, float* reward,
                            uint8_t* done, const StepCtl& ctl, const RolloutArgs& ra) {
  CRAFTER_LAUNCH(crafter_requeue_rollout_kernel, dim3(grid), dim3(kRequeueThreads), lds, stream, start, stop, cfg, tb, st, actions, obs,
                 reward, done, ctl, ra);
}

hipError_t rollout_allow_lds(int bytes) {
  const void* big[] = {(const void*)crafter_rollout_kernel<0, 0, 0>, (const void*)crafter_rollout_kernel<0, 2, 1>, (const void*)crafter_rollout_kernel<1, 0, 0>,
                       (const void*)crafter_requeue_rollout_kernel};
  for (const void* f : big) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace crafter
