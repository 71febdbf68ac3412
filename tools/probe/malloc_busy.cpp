// Does hipMalloc / hipMallocAsync / hipFree wait for a running kernel?   hipcc --offload-arch=gfx950 -O2 malloc_busy.cpp -o malloc_busy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define OK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); } } while (0)
__global__ void spin(unsigned long long* p, unsigned long long n) { unsigned long long a = 0; for (unsigned long long i = 0; i < n; i++) a += i * i + (a >> 3); if (a == 42) *p = a; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long* sink; OK(hipMalloc(&sink, 8));
  void* warm; OK(hipMallocAsync(&warm, 4096, s)); OK(hipFreeAsync(warm, s)); OK(hipStreamSynchronize(s));
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, sink, 4000000ull);
    OK(hipStreamSynchronize(s));
    const double t_kernel = now() - t0;
    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, sink, 4000000ull);
    void *a, *b;
    t0 = now(); OK(hipMalloc(&a, 160 << 20)); const double t_m = now() - t0;
    t0 = now(); OK(hipMallocAsync(&b, 160 << 20, s)); const double t_ma = now() - t0;
    t0 = now(); OK(hipStreamSynchronize(s)); const double t_rest = now() - t0;
    printf("kernel alone %.2f ms | while it runs: hipMalloc(160 MB) %.3f ms, hipMallocAsync(160 MB) %.3f ms, then the kernel needed %.2f ms more\n",
           t_kernel * 1e3, t_m * 1e3, t_ma * 1e3, t_rest * 1e3);
    OK(hipFree(a)); OK(hipFreeAsync(b, s)); OK(hipStreamSynchronize(s));
  }
  return 0;
}
