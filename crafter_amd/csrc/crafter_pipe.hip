// crafter_pipe_kernel: see crafter_pipe.hpp (what and why), env_kernels.hpp rules_pipe_loop / frame_pipe_loop (the two halves).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "crafter_pipe.hpp"
#include "dispatch_order.hpp"
#include "wave_gfx950.hpp"

namespace crafter {
namespace {

__global__ void __launch_bounds__(kPipeThreads, 5)   // five waves per SIMD = five workgroups per CU: at most 96 VGPRs
crafter_pipe_kernel(Config cfg_in, TablePtrs tb, StatePtrs st, const int32_t* __restrict__ actions, uint8_t* __restrict__ obs,
                    float* __restrict__ reward, uint8_t* __restrict__ done, StepCtl ctl, PipeArgs pa) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const Config cfg = with_default_geometry(cfg_in);
  int b = (int)blockIdx.x;
  if (ctl.order_build) {   // block 0 sorts the envs for the launch after this one (dispatch_order.hpp), as in crafter_step_kernel
    if (b == 0) {
      build_order(cfg, tb, ctl.order_build, ctl.next_step, (uint32_t*)smem);
      return;
    }
    b -= 1;
  }
  uint8_t* frame_base = smem + lane_layout(cfg).total;
  const FrameLayout F = frame_layout(cfg, false);
  uint8_t* slots = frame_base + F.total;
  uint32_t* pctl = (uint32_t*)(slots + 2 * pipe_slot_bytes());
  if (threadIdx.x < 4) pctl[threadIdx.x] = 0;
  __syncthreads();   // the workgroup's only s_barrier: from here on its two halves run different code
  PipeLink link;
  link.slots = slots;
  link.ctl = pctl;
  link.published = 0;
  if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) < 64) {
    WaveGfx950<64, 1, 1> w;
    rules_pipe_loop(w, smem, link, b, pa.workgroups, cfg, tb, st, actions, obs, reward, done, ctl, pa.tickets, pa.ticket_base);
  } else {
    WaveGfx950<kPipeFrameThreads, 1, 2> w;
    w.bar = pctl + 3;
    frame_pipe_loop(w, frame_base, link, cfg, tb, st, obs, pa.night_px);
  }
}

}  // namespace

void launch_pipe(int grid, size_t lds, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const Config& cfg, const TablePtrs& tb,
                 const StatePtrs& st, const int32_t* actions, uint8_t* obs, float* reward, uint8_t* done, const StepCtl& ctl,
                 const PipeArgs& pa) {
  if (start != nullptr || stop != nullptr)
    hipExtLaunchKernelGGL(crafter_pipe_kernel, dim3(grid), dim3(kPipeThreads), lds, stream, start, stop, 0, cfg, tb, st, actions, obs, reward,
                          done, ctl, pa);
  else
    hipLaunchKernelGGL(crafter_pipe_kernel, dim3(grid), dim3(kPipeThreads), lds, stream, cfg, tb, st, actions, obs, reward, done, ctl, pa);
}

}  // namespace crafter
