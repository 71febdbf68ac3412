// GPU probe: random-touch rates of MI355X by access flavour and scope (round 2).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/atomic_scope.cpp -o /tmp/atomic_scope && /tmp/atomic_scope
// Question behind it: do workgroup-scope (no sc1) integer atomics execute in the XCD's L2 -- which would make
// an XCD-affine incr kernel worth building -- or at the memory side like agent-scope ones?  Also: plain
// read-modify-write and plain scattered 4-byte stores (what a globally de-duplicated batch would need),
// and the block -> XCC map.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint64_t splitmix_at(uint64_t seed, uint64_t j) {
  uint64_t z = seed + (j + 1) * 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

__device__ inline uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

// MODE 0 read8 | 1 atomic ret agent | 2 atomic noret agent | 3 atomic ret workgroup | 4 atomic noret workgroup
//      5 plain RMW (load, add, store 4 B) | 6 plain 4-B store | 7 atomic ret wavefront scope | 8 CAS64 agent
//      9 atomic ret agent, XCD-affine (word index forced into the XCD's eighth of the buffer)
//     10 atomic ret workgroup, XCD-affine
template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t* buf, uint64_t words, uint64_t touches, uint64_t seed, unsigned long long* sink) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t xcc = xcc_id() & 7u;
  uint64_t acc = 0;
  for (uint64_t i = t; i < touches; i += 4 * stride) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint64_t j = i + q * stride;
      if (j >= touches) break;
      uint64_t w = splitmix_at(seed, j) % words;
      if (MODE == 9 || MODE == 10) w = (w & ~7ull) | 0;   // placeholder, replaced below
      if (MODE == 9 || MODE == 10) { const uint64_t per = words / 8; w = (splitmix_at(seed, j) % per) + per * xcc; }
      uint32_t* p = reinterpret_cast<uint32_t*>(&buf[w]);
      if (MODE == 0) acc += buf[w];
      else if (MODE == 1 || MODE == 9) acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 2) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 3 || MODE == 10) acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 4) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 5) { uint32_t v = *p; *p = v + 1u; acc += v; }
      else if (MODE == 6) *p = (uint32_t)j;
      else if (MODE == 7) acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      else if (MODE == 8) acc += atomicCAS(reinterpret_cast<unsigned long long*>(&buf[w]), 0ull, (unsigned long long)j + 1);
    }
  }
  if (acc == 0x1234567deadbeefULL) *sink = acc;
}

__global__ void k_xcc_map(uint32_t* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// correctness of workgroup-scope atomics with adders on all XCDs: every lane adds 1 to word (j % words)
template <int SCOPE>
__global__ __launch_bounds__(256) void k_count(uint32_t* buf, uint32_t words, uint64_t touches) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = t; j < touches; j += stride) {
    uint32_t* p = &buf[splitmix_at(7, j) % words];
    if (SCOPE == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
__global__ void k_sum(const uint32_t* buf, uint32_t words, unsigned long long* out) {
  unsigned long long s = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) s += buf[i];
  atomicAdd(out, s);
}

template <int MODE>
void run(const char* name, uint64_t* buf, uint64_t bytes, uint64_t touches, unsigned long long* sink) {
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    OK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE>), dim3(256 * 8 * 4), dim3(256), 0, 0, buf, bytes / 8, touches, 99 + rep, sink);
    OK(hipEventRecord(e1, 0));
    OK(hipDeviceSynchronize());
    float ms; OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("  %-46s %8.3f ms  %7.2f G touches/s\n", name, best, touches / (best * 1e-3) / 1e9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  unsigned long long* sink; OK(hipMalloc(&sink, 8));
  uint32_t* map; OK(hipMalloc(&map, 64 * 4));
  hipLaunchKernelGGL(k_xcc_map, dim3(64), dim3(64), 0, 0, map);
  uint32_t h[64]; OK(hipMemcpy(h, map, sizeof h, hipMemcpyDeviceToHost));
  printf("xcc of blocks 0..63:"); for (int i = 0; i < 64; i++) printf(" %u", h[i]); printf("\n");

  // are workgroup-scope adds from all XCDs lost?  (hot: 1024 words; cold: 64 M words)
  for (uint32_t words : {1024u, 1u << 26}) {
    uint32_t* cb; OK(hipMalloc(&cb, (size_t)words * 4));
    unsigned long long* tot; OK(hipMalloc(&tot, 8));
    for (int scope = 0; scope < 2; scope++) {
      OK(hipMemset(cb, 0, (size_t)words * 4)); OK(hipMemset(tot, 0, 8));
      const uint64_t touches = 1ull << 26;
      if (scope == 0) hipLaunchKernelGGL((k_count<0>), dim3(8192), dim3(256), 0, 0, cb, words, touches);
      else hipLaunchKernelGGL((k_count<1>), dim3(8192), dim3(256), 0, 0, cb, words, touches);
      hipLaunchKernelGGL(k_sum, dim3(1024), dim3(256), 0, 0, cb, words, tot);
      unsigned long long ht; OK(hipMemcpy(&ht, tot, 8, hipMemcpyDeviceToHost));
      printf("count check words=%u scope=%s: sum=%llu of %llu %s\n", words, scope ? "workgroup" : "agent", ht,
             (unsigned long long)touches, ht == touches ? "OK" : "LOST");
    }
    OK(hipFree(cb)); OK(hipFree(tot));
  }

  for (uint64_t gib_x4 : {1ull /*256 MiB*/, 16ull /*4 GiB*/, 128ull /*32 GiB*/}) {
    const uint64_t bytes = gib_x4 << 28;
    uint64_t* buf; OK(hipMalloc(&buf, bytes));
    for (uint64_t off = 0; off < bytes; off += 1ull << 30) OK(hipMemset((char*)buf + off, 0, bytes - off < (1ull << 30) ? bytes - off : (1ull << 30)));
    OK(hipDeviceSynchronize());
    const uint64_t touches = 1ull << 27;
    printf("buffer %.2f GiB, %llu touches\n", bytes / 1073741824.0, (unsigned long long)touches);
    run<0>("read8", buf, bytes, touches, sink);
    run<1>("atomicAdd u32 returning, agent", buf, bytes, touches, sink);
    run<2>("atomicAdd u32 no return, agent", buf, bytes, touches, sink);
    run<3>("atomicAdd u32 returning, workgroup scope", buf, bytes, touches, sink);
    run<4>("atomicAdd u32 no return, workgroup scope", buf, bytes, touches, sink);
    run<7>("atomicAdd u32 returning, wavefront scope", buf, bytes, touches, sink);
    run<9>("atomicAdd ret agent, XCD-affine eighths", buf, bytes, touches, sink);
    run<10>("atomicAdd ret workgroup, XCD-affine eighths", buf, bytes, touches, sink);
    run<8>("atomicCAS u64 returning, agent", buf, bytes, touches, sink);
    run<5>("plain load + store 4 B (RMW)", buf, bytes, touches, sink);
    run<6>("plain store 4 B", buf, bytes, touches, sink);
    OK(hipFree(buf));
  }
  return 0;
}
