// GPU probe: cost of a grid-wide barrier on MI355X (256 workgroups x 1024 lanes), by flavour.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/gridsync.cpp -o gridsync && ./gridsync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned* bar, unsigned iters, unsigned* data) {
  auto grid = cooperative_groups::this_grid();
  unsigned target = 0;
  for (unsigned i = 0; i < iters; i++) {
    if (MODE == 0) { grid.sync(); continue; }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 2 || MODE == 4) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      target += gridDim.x;
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      if (MODE == 3 || MODE == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  if (data && threadIdx.x == 0) data[blockIdx.x] = target;
}

template <int MODE>
int run(const char* name, unsigned* bar, unsigned* data, int blocks) {
  unsigned iters = 200;
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; rep++) {
    OK(hipMemset(bar, 0, 4));
    void* args[] = {&bar, &iters, &data};
    OK(hipEventRecord(e0, 0));
    OK(hipLaunchCooperativeKernel((const void*)&k<MODE>, dim3(blocks), dim3(1024), args, 0, 0));
    OK(hipEventRecord(e1, 0));
    OK(hipDeviceSynchronize());
    float ms; OK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("%-44s %7.2f us per barrier\n", name, ms * 1e3 / iters);
  }
  return 0;
}

int main() {
  unsigned *bar, *data; OK(hipMalloc(&bar, 4)); OK(hipMalloc(&data, 4096));
  int cus; OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  printf("%d CUs\n", cus);
  run<0>("cooperative_groups grid.sync()", bar, data, cus);
  run<1>("atomic counter, no cache maintenance", bar, data, cus);
  run<2>("atomic counter + release fence (wbl2)", bar, data, cus);
  run<3>("atomic counter + acquire fence (inv)", bar, data, cus);
  run<4>("atomic counter + release + acquire", bar, data, cus);
  return 0;
}
