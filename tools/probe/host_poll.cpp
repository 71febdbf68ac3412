// How long after a kernel has written its verdict does the host know?  (a) hipMemcpyAsync D2H + hipStreamSynchronize -- what ctl_read
// does; (b) the kernel stores to coherent pinned host memory, the host spins on a sequence word.  Round-trip of launch + wait, 2000 reps.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/host_poll.cpp -o /tmp/host_poll && /tmp/host_poll
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <immintrin.h>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Blk { uint32_t w[38]; };
__global__ void k_work(uint32_t* d, uint32_t v) { d[threadIdx.x] = v + threadIdx.x; }
__global__ void k_publish(const Blk* d, Blk* h, uint32_t* seq, uint32_t v) {
  if (threadIdx.x < 38) __hip_atomic_store(&h->w[threadIdx.x], d->w[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(seq, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main() {
  hipStream_t s; OK(hipStreamCreate(&s));
  Blk* d; OK(hipMalloc(&d, sizeof(Blk)));
  Blk* h_plain; OK(hipHostMalloc(&h_plain, sizeof(Blk)));
  Blk* h_coh; OK(hipHostMalloc(&h_coh, sizeof(Blk) + 64, hipHostMallocCoherent));
  uint32_t* seq = reinterpret_cast<uint32_t*>(h_coh + 1);
  *seq = 0;
  const int R = 2000;
  for (int mode = 0; mode < 2; mode++) {
    for (int warm = 0; warm < 2; warm++) {
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 1; r <= R; r++) {
        hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, (uint32_t*)d, (uint32_t)r);
        if (mode == 0) {
          OK(hipMemcpyAsync(h_plain, d, sizeof(Blk), hipMemcpyDeviceToHost, s));
          OK(hipStreamSynchronize(s));
          if (h_plain->w[5] != (uint32_t)r + 5) { printf("bad copy\n"); return 1; }
        } else {
          const uint32_t v = (uint32_t)(warm * R + r);
          hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, d, h_coh, seq, v);
          while (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != v) _mm_pause();
          if (h_coh->w[5] != (uint32_t)r + 5) { printf("bad publish %u\n", h_coh->w[5]); return 1; }
        }
      }
      auto t1 = std::chrono::steady_clock::now();
      if (warm) printf("%s: %.2f us per launch + wait\n", mode == 0 ? "memcpy D2H + stream sync " : "publish kernel + host spin",
                       std::chrono::duration<double, std::micro>(t1 - t0).count() / R);
    }
  }
  OK(hipStreamSynchronize(s));
  return 0;
}
