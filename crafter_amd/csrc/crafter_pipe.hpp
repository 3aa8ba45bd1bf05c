// The pipelined step kernel of the default instance (crafter.Env() defaults, frames drawn): ONE workgroup = a rule wave and
// a frame group of three waves, persistent over the envs at its positions of the dispatch order.  The rule wave runs the
// rule half of a step (the split step's rule body: lane-register occupancy, a window of the material map, 9.7 KB) and hands
// what the frame depends on -- the 192-byte frame record, at night the MT19937 state -- to the frame group THROUGH LDS; while
// the group draws env k's frame (the split step's frame body) the rule wave is already running env k + 1's rules.  In the
// fused step kernel three of a workgroup's four waves wait while the first runs the rules, and the rule wave waits while
// the frame is drawn; here neither does, and nothing crosses a workgroup or an XCD: the hand-off is an LDS counter polled
// with s_sleep (the cross-stream / cross-XCD forms of the same idea failed in round 3: DESIGN.md 5).
// A translation unit of its own for the reason crafter_rollout.hpp gives: the kernel is two loops around step bodies.
#pragma once
#include <hip/hip_runtime.h>

#include "env_kernels.hpp"

namespace crafter {

struct PipeArgs {
  uint32_t* night_px;   // [N][frame_night_px_words] the frame group's scratch: a night frame's pixels in noise-stream order
  int workgroups;       // pipeline workgroups of the launch (the grid without the block that builds the dispatch order)
  int32_t* tickets;     // the handle's ticket counter (env_kernels.hpp rules_pipe_loop), or null: static walks
  uint32_t ticket_base; // first ticket of this launch
};

constexpr int kPipeThreads = 256;       // rule wave + three frame waves
constexpr int kPipeFrameThreads = 192;
__host__ __device__ inline int pipe_lds_bytes(const Config& c) { return lane_layout(c).total + frame_layout(c, false).total + 2 * pipe_slot_bytes() + 16; }

void launch_pipe(int grid, size_t lds, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const Config& cfg, const TablePtrs& tb,
                 const StatePtrs& st, const int32_t* actions, uint8_t* obs, float* reward, uint8_t* done, const StepCtl& ctl,
                 const PipeArgs& pa);

}  // namespace crafter
