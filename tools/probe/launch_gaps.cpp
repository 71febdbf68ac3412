// GPU-side cost of a chain of small dependent kernels behind a long one (the shape of a write batch: folding kernel, then ~14 kernels of a
// growth round): (a) plain launches on one stream, queued while the long kernel runs; (b) the same chain captured in a hipGraph.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/launch_gaps.cpp -o /tmp/launch_gaps && /tmp/launch_gaps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_long(uint32_t* d, uint32_t n) { uint32_t a = threadIdx.x; for (uint32_t i = 0; i < n; i++) a = a * 1664525u + 1013904223u; d[blockIdx.x * blockDim.x + threadIdx.x] = a; }
__global__ void k_small(uint32_t* d, uint32_t v) { d[blockIdx.x * blockDim.x + threadIdx.x] += v; }
int main() {
  hipStream_t s; OK(hipStreamCreate(&s));
  uint32_t* d; OK(hipMalloc(&d, 1 << 24));
  hipEvent_t e0, e1, e2; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1)); OK(hipEventCreate(&e2));
  const int CH = 14, R = 50;
  for (int grid : {64, 4096}) {
    float t_long = 0, t_chain = 0;
    for (int r = 0; r < R; r++) {
      OK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_long, dim3(1024), dim3(256), 0, s, d, 20000u);
      OK(hipEventRecord(e1, s));
      for (int k = 0; k < CH; k++) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, d, (uint32_t)k);
      OK(hipEventRecord(e2, s));
      OK(hipStreamSynchronize(s));
      float a, b; OK(hipEventElapsedTime(&a, e0, e1)); OK(hipEventElapsedTime(&b, e1, e2));
      if (r >= 5) { t_long += a; t_chain += b; }
    }
    printf("grid %5d: long kernel %.1f us; %d small kernels behind it, plain launches: %.1f us = %.2f us each\n", grid, t_long / (R - 5) * 1e3, CH,
           t_chain / (R - 5) * 1e3, t_chain / (R - 5) * 1e3 / CH);
    // the same in a graph
    hipGraph_t g; hipGraphExec_t ge;
    OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < CH; k++) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, d, (uint32_t)k);
    OK(hipStreamEndCapture(s, &g));
    OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    t_chain = 0;
    for (int r = 0; r < R; r++) {
      hipLaunchKernelGGL(k_long, dim3(1024), dim3(256), 0, s, d, 20000u);
      OK(hipEventRecord(e1, s));
      OK(hipGraphLaunch(ge, s));
      OK(hipEventRecord(e2, s));
      OK(hipStreamSynchronize(s));
      float b; OK(hipEventElapsedTime(&b, e1, e2));
      if (r >= 5) t_chain += b;
    }
    printf("grid %5d: the same chain as a hipGraph: %.1f us = %.2f us each\n", grid, t_chain / (R - 5) * 1e3, t_chain / (R - 5) * 1e3 / CH);
    OK(hipGraphExecDestroy(ge)); OK(hipGraphDestroy(g));
  }
  return 0;
}
