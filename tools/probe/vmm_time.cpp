// times VMM chunk creation + map + set-access + memset for several chunk sizes (and plain hipMalloc)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc ad = {}; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite;
  void* p = nullptr; size_t R = (size_t)64 << 30;
  hipMemAddressReserve(&p, R, 2 << 20, nullptr, 0);
  char* base = (char*)p; size_t mapped = 0;
  size_t sizes[] = {64u << 20, 256u << 20, 512u << 20, (size_t)1 << 30, (size_t)2 << 30, (size_t)1 << 30, 512u << 20, (size_t)4 << 30};
  for (size_t s : sizes) {
    hipMemGenericAllocationHandle_t h;
    double t0 = now(); hipMemCreate(&h, s, &prop, 0);
    double t1 = now(); hipMemMap(base + mapped, s, 0, h, 0);
    double t2 = now(); hipMemSetAccess(base, mapped + s, &ad, 1);
    double t3 = now(); hipMemset(base + mapped, 0, s); hipDeviceSynchronize();
    double t4 = now();
    printf("chunk %5zu MiB at %6zu MiB: create %7.2f  map %6.2f  setaccess(whole) %7.2f  memset %6.2f ms\n", s >> 20, mapped >> 20, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    mapped += s;
  }
  for (size_t s : {(size_t)1 << 30, (size_t)4 << 30}) {
    void* q; double t0 = now(); hipMalloc(&q, s); double t1 = now(); hipMemset(q, 0, s); hipDeviceSynchronize(); double t2 = now();
    printf("hipMalloc %zu MiB: %.2f ms, memset %.2f ms\n", s >> 20, t1 - t0, t2 - t1);
  }
  return 0;
}
