// crafter_step_n's kernels live in a translation unit of their own (crafter_rollout.hip) because they are compiled with
// one more flag: -mllvm -disable-machine-licm.  A rollout is a loop around the step; the machine-level loop-invariant code
// motion pass moves every constant and address a step materialises out of that loop and keeps it in registers across the
// steps (234 VGPRs against the step kernel's 71, two workgroups per CU instead of five); without it the loop costs 9 VGPRs.
// The step kernel itself must not be compiled that way (its inner loops want the pass), hence the second unit.
#pragma once
#include <hip/hip_runtime.h>

#include "env_kernels.hpp"

namespace crafter {

struct RolloutArgs {
  int T;
  size_t obs_stride;      // bytes between the observations of consecutive steps
  int32_t* stalled_at;    // [N] the step an env stopped at for want of a world (valid for the envs in the regeneration queue)
};

// instance: bit 2 = maps in LDS, bit 1 = default geometry, bit 0 = default rules (as crafter_step_instance reports it); 9 = maps and
// slot table in global memory, default view and default rules compiled in (crafter_rollout_kernel<0, 2, 1>)
void launch_rollout(int instance, int num_envs, size_t lds, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const Config& cfg,
                    const TablePtrs& tb, const StatePtrs& st, const int32_t* actions, uint8_t* obs, float* reward, uint8_t* done,
                    const StepCtl& ctl, const RolloutArgs& ra);
void launch_requeue_rollout(int grid, size_t lds, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const Config& cfg,
                            const TablePtrs& tb, const StatePtrs& st, const int32_t* actions, uint8_t* obs, float* reward,
                            uint8_t* done, const StepCtl& ctl, const RolloutArgs& ra);
// large worlds: lets the generic instances take `bytes` of dynamic LDS
hipError_t rollout_allow_lds(int bytes);

}  // namespace crafter
