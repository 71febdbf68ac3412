#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); printf("%-70s -> %s\n", #x, hipGetErrorString(e)); }while(0)
int main(){ setvbuf(stdout,NULL,_IONBF,0);
  hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gmin=0, grec=0;
  CK(hipMemGetAllocationGranularity(&gmin,&prop,hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec,&prop,hipMemAllocationGranularityRecommended));
  printf("gran min=%zu rec=%zu\n", gmin, grec);
  size_t f,t; CK(hipMemGetInfo(&f,&t)); printf("free=%zu total=%zu\n", f,t);
  void* p=nullptr; size_t R = (size_t)64<<30;
  CK(hipMemAddressReserve(&p, R, grec, nullptr, 0)); printf("p=%p\n", p);
  char* base=(char*)p; size_t mapped=0;
  size_t sizes[] = {4u<<20, 2u<<20, 6u<<20, 64u<<20, 1u<<30};
  hipMemAccessDesc ad = {}; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite;
  for (size_t s : sizes) {
    hipMemGenericAllocationHandle_t h;
    printf("--- chunk %zu at off %zu\n", s, mapped);
    CK(hipMemCreate(&h, s, &prop, 0));
    CK(hipMemMap(base+mapped, s, 0, h, 0));
    hipError_t ea = hipMemSetAccess(base+mapped, s, &ad, 1); printf("setaccess(sub) -> %s\n", hipGetErrorString(ea));
    if (ea != hipSuccess) { ea = hipMemSetAccess(base, mapped+s, &ad, 1); printf("setaccess(whole from base) -> %s\n", hipGetErrorString(ea)); }
    if (ea == hipSuccess) { hipError_t e = hipMemset(base+mapped, 0, s); printf("memset -> %s\n", hipGetErrorString(e)); hipDeviceSynchronize(); }
    mapped += s;
  }
  // alt: set access over whole range from base
  CK(hipMemSetAccess(base, mapped, &ad, 1));
  return 0;
}
