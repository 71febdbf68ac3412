// What does giving device memory back cost?  hipFree (synchronises the device), hipFreeAsync on hipMalloc'ed memory, and the stream-ordered
// pair hipMallocAsync / hipFreeAsync, each beside a kernel that keeps the device busy.   hipcc --offload-arch=gfx950 -O2 free_cost.cpp -o free_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define OK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); } } while (0)
__global__ void spin(unsigned long long* p, unsigned long long n) { unsigned long long a = 0; for (unsigned long long i = 0; i < n; i++) a += i * i; if (a == 42) *p = a; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long* sink; OK(hipMalloc(&sink, 8));
  const size_t sizes[] = {1 << 20, 64 << 20, 256 << 20};
  for (size_t sz : sizes) {
    for (int busy = 0; busy < 2; busy++) {
      void *a, *b, *c;
      OK(hipMalloc(&a, sz)); OK(hipMalloc(&b, sz));
      double t0 = now(); hipError_t e = hipMallocAsync(&c, sz, s); double t_ma = now() - t0;
      if (e != hipSuccess) { printf("hipMallocAsync: %s\n", hipGetErrorString(e)); c = nullptr; }
      if (busy) hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, sink, 3000000ull);      // ~ms of device work in flight
      t0 = now(); OK(hipFree(a)); double t_free = now() - t0;
      if (busy) hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, sink, 3000000ull);
      t0 = now(); e = hipFreeAsync(b, s); double t_fa = now() - t0;
      if (e != hipSuccess) { printf("hipFreeAsync(hipMalloc'ed): %s\n", hipGetErrorString(e)); OK(hipFree(b)); }
      double t_fa2 = -1;
      if (c) { t0 = now(); OK(hipFreeAsync(c, s)); t_fa2 = now() - t0; }
      OK(hipStreamSynchronize(s));
      printf("%4zu MB, device %s: hipFree %.3f ms | hipFreeAsync(hipMalloc'ed) %.3f ms | hipMallocAsync %.3f ms, its hipFreeAsync %.3f ms\n", sz >> 20,
             busy ? "BUSY" : "idle", t_free * 1e3, t_fa * 1e3, t_ma * 1e3, t_fa2 * 1e3);
    }
  }
  return 0;
}
