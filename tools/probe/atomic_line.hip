// Do returning device-scope atomics of ONE wave instruction that fall into one 128-byte line cost one serialised step or one per
// lane?  (dense ids: the 16 hottest cells of the hottest row share a line, and every tile of k_apply_agg adds to each of them)
//   hipcc --offload-arch=gfx950 -O2 tools/probe/atomic_line.hip -o libsmatrix_amd/lib/ab/atomic_line && libsmatrix_amd/lib/ab/atomic_line
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// mode 0: lane 0 of every wave adds to word 1 (same address)
// mode 1: lanes 0..15 of every wave add to the value words of 16 adjacent cells (one instruction, one line)
// mode 2: the same 16 adds as 16 instructions of lane 0
// mode 3: lanes 0..15 add to 16 cells in 16 DIFFERENT lines (4 KB apart)
// mode 4: wave w adds to cell (w & 15) of the line: one lane per wave, 16 addresses in one line
// mode 5: wave w adds to cell (w & 15) in 16 different lines
__global__ void k(uint32_t* buf, uint32_t* sink, int mode) {
  const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t r = 0;
  if (mode == 0) { if (lane == 0) r = atomicAdd(&buf[1], 1u); }
  else if (mode == 1) { if (lane < 16) r = atomicAdd(&buf[lane * 2 + 1], 1u); }
  else if (mode == 2) { if (lane == 0) for (int q = 0; q < 16; q++) r += atomicAdd(&buf[q * 2 + 1], 1u); }
  else if (mode == 3) { if (lane < 16) r = atomicAdd(&buf[lane * 1024 + 1], 1u); }
  else if (mode == 4) { if (lane == 0) r = atomicAdd(&buf[(wave & 15u) * 2 + 1], 1u); }
  else if (mode == 5) { if (lane == 0) r = atomicAdd(&buf[(wave & 15u) * 1024 + 1], 1u); }
  if (r == 0xFFFFFFFFu) sink[0] = r;
}
int main() {
  uint32_t *buf, *sink; OK(hipMalloc(&buf, 1 << 20)); OK(hipMalloc(&sink, 64));
  hipEvent_t a, b; OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
  const int waves = 1 << 17;
  const char* name[] = {"1 lane, same word", "16 lanes, 16 cells of one line (one instruction)", "lane 0, 16 instructions on one line",
                        "16 lanes, 16 lines (one instruction)", "1 lane per wave, 16 cells of one line", "1 lane per wave, 16 lines"};
  for (int mode = 0; mode < 6; mode++) for (int rep = 0; rep < 2; rep++) {
    OK(hipMemset(buf, 0, 1 << 20));
    OK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(waves / 4), dim3(256), 0, 0, buf, sink, mode);
    OK(hipEventRecord(b)); OK(hipEventSynchronize(b));
    float ms; OK(hipEventElapsedTime(&ms, a, b));
    const double adds = (double)waves * (mode == 1 || mode == 2 || mode == 3 ? 16 : 1);
    if (rep) printf("%-52s %8.3f ms  %7.1f ns per wave  %6.2f ns per add\n", name[mode], ms, ms * 1e6 / waves, ms * 1e6 / adds);
  }
  return 0;
}
