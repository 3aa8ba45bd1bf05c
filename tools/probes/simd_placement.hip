// GPU box probe: on which SIMD of its CU does each wave of a 256-thread workgroup land?  (the step kernel gives wave 0 the
// serial rules and waves 1..3 the staging / noise chain: if every workgroup's wave 0 shares one SIMD, that SIMD is the bottleneck)
// build: hipcc --offload-arch=gfx950 -O2 -o simd_placement simd_placement.hip ; run: ./simd_placement [lds_bytes] [grid]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
  extern __shared__ unsigned char smem[];
  unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
  unsigned x = threadIdx.x;
  for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;   // stay resident for a while
  if (x == 12345u) smem[0] = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = id;
}
int main(int argc, char** argv) {
  int lds = argc > 1 ? atoi(argv[1]) : 26872, grid = argc > 2 ? atoi(argv[2]) : 4096;
  unsigned* d; hipMalloc(&d, grid * 4 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, d, 20000);
  std::vector<unsigned> h(grid * 4);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  long hist[4][4] = {};   // [wave of the workgroup][simd]
  long pattern[256] = {};
  for (int b = 0; b < grid; b++) {
    int key = 0;
    for (int wv = 0; wv < 4; wv++) { int s = (h[b * 4 + wv] >> 4) & 3; hist[wv][s]++; key = key * 4 + s; }
    pattern[key]++;
  }
  printf("lds %d grid %d\n", lds, grid);
  for (int wv = 0; wv < 4; wv++) printf("wave %d: simd0 %ld simd1 %ld simd2 %ld simd3 %ld\n", wv, hist[wv][0], hist[wv][1], hist[wv][2], hist[wv][3]);
  for (int k = 0; k < 256; k++) if (pattern[k]) printf("pattern %d%d%d%d: %ld\n", k >> 6, (k >> 4) & 3, (k >> 2) & 3, k & 3, pattern[k]);
  printf("first blocks:"); for (int b = 0; b < 12; b++) printf(" [%u%u%u%u cu%u se%u]", (h[b*4]>>4)&3, (h[b*4+1]>>4)&3, (h[b*4+2]>>4)&3, (h[b*4+3]>>4)&3, (h[b*4]>>8)&15, (h[b*4]>>13)&7); printf("\n");
  return 0;
}
