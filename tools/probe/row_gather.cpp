// GPU probe: random gathers of 2 KiB rows (the CF shape of config 3: 256 cells x 8 B) out of a 27 GB buffer, one wave per
// row, K rows in flight per wave, with and without a dependent index load in front (the directory lookup of k_getrow).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/row_gather.cpp -o /tmp/row_gather && /tmp/row_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ inline uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bU; h ^= h >> 13; h *= 0xc2b2ae35U; h ^= h >> 16; return h; }

// MODE 0: row index computed (no dependent load); 1: row index read from a random 16-byte "directory" slot first;
//      2: the request id is loaded first as well (xs[r] -> directory slot -> cells: k_getrow's chain)
template <int K, int MODE>
__global__ __launch_bounds__(256) void k(const uint4* buf, uint32_t nrows, const uint4* dir, uint32_t dmask, uint32_t n, unsigned long long* sink) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, nwaves = (gridDim.x * blockDim.x) >> 6;
  uint64_t acc = 0;
  for (uint32_t r0 = wave; r0 < n; r0 += K * nwaves) {
    uint32_t row[K];
    uint4 d[K];
#pragma unroll
    for (int k2 = 0; k2 < K; k2++) {
      const uint32_t r = r0 + k2 * nwaves;
      row[k2] = fmix32(r * 2654435761u + 12345u) % nrows;
      if (MODE == 1) d[k2] = dir[fmix32(r) & dmask];
      if (MODE == 2) d[k2] = dir[fmix32(reinterpret_cast<const uint32_t*>(sink + 1)[r]) & dmask];
    }
    uint4 c[K][2];
#pragma unroll
    for (int k2 = 0; k2 < K; k2++) {
      if (MODE >= 1) row[k2] = (row[k2] + (d[k2].x & 1u)) % nrows;      // depends on the loaded slot
      c[k2][0] = buf[(size_t)row[k2] * 128 + lane];
      c[k2][1] = buf[(size_t)row[k2] * 128 + 64 + lane];
    }
#pragma unroll
    for (int k2 = 0; k2 < K; k2++) acc += c[k2][0].x + c[k2][0].w + c[k2][1].y + c[k2][1].z;
  }
  if (acc == 0x1234567deadbeefULL) *sink = acc;
}

// the whole of k_getrow's row: id -> directory slot -> 256 cells -> offsets -> ballot compaction -> 8-byte stores at the
// pairs' ranks -> count.  Cells are "non-empty" by a hash of their position (45 %); STORE 0: the stores are left out
template <int STORE>
__global__ __launch_bounds__(256) void k_full(const uint4* buf, uint32_t nrows, const uint4* dir, uint32_t dmask, uint32_t n, const uint32_t* xs,
                                              const uint64_t* offsets, uint64_t* ret, uint32_t* counts) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint64_t lt = (1ull << lane) - 1;
  __shared__ uint64_t l_pairs[4][2][256];              // STORE 2: the row's pairs are compacted here first, then written 16 B per lane
  for (uint32_t r0 = wave; r0 < n; r0 += 2 * nwaves) {
    uint32_t row[2], r[2];
    uint4 d[2];
    for (int k2 = 0; k2 < 2; k2++) {
      r[k2] = min(r0 + k2 * nwaves, n - 1);
      d[k2] = dir[fmix32(xs[r[k2]]) & dmask];
    }
    uint4 c[2][2];
    for (int k2 = 0; k2 < 2; k2++) {
      row[k2] = (fmix32(r[k2] * 2654435761u + 12345u) + (d[k2].x & 1u)) % nrows;
      c[k2][0] = buf[(size_t)row[k2] * 128 + lane];
      c[k2][1] = buf[(size_t)row[k2] * 128 + 64 + lane];
    }
    for (int k2 = 0; k2 < 2; k2++) {
      const uint64_t off = offsets[r[k2]];
      const uint32_t cap = (uint32_t)(offsets[r[k2] + 1] - off);
      uint32_t written = 0;
      for (int j = 0; j < 2; j++) {
        const uint32_t p = row[k2] * 256 + j * 128 + 2 * lane;
        const bool ne0 = (fmix32(p) % 100) < 45 + (c[k2][j].x & 0), ne1 = (fmix32(p + 1) % 100) < 45 + (c[k2][j].z & 0);
        const uint64_t m0 = __ballot(ne0), m1 = __ballot(ne1);
        uint32_t rank = written + (uint32_t)__popcll(m0 & lt) + (uint32_t)__popcll(m1 & lt);
        if (STORE == 1 && ne0 && rank < cap) ret[off + rank] = ((uint64_t)c[k2][j].y << 32) | c[k2][j].x;
        if (STORE == 2 && ne0) l_pairs[threadIdx.x >> 6][k2][rank] = ((uint64_t)c[k2][j].y << 32) | c[k2][j].x;
        rank += ne0;
        if (STORE == 1 && ne1 && rank < cap) ret[off + rank] = ((uint64_t)c[k2][j].w << 32) | c[k2][j].z;
        if (STORE == 2 && ne1) l_pairs[threadIdx.x >> 6][k2][rank] = ((uint64_t)c[k2][j].w << 32) | c[k2][j].z;
        written += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
      }
      if (STORE == 2) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t nw = min(written, cap);
        // head pair if the destination is not 16-byte aligned, then 16 B per lane, then the tail pair
        const uint32_t head = (uint32_t)(off & 1u) & (nw ? 1u : 0u);
        if (lane == 0 && head) ret[off] = l_pairs[threadIdx.x >> 6][k2][0];
        const uint32_t pairs2 = (nw - head) / 2;
        for (uint32_t q = lane; q < pairs2; q += 64) {
          const uint64_t a = l_pairs[threadIdx.x >> 6][k2][head + 2 * q], b = l_pairs[threadIdx.x >> 6][k2][head + 2 * q + 1];
          *reinterpret_cast<uint4*>(&ret[off + head + 2 * q]) = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
        }
        if (lane == 0 && ((nw - head) & 1u)) ret[off + nw - 1] = l_pairs[threadIdx.x >> 6][k2][nw - 1];
        __builtin_amdgcn_wave_barrier();
      }
      if (lane == 0) counts[r[k2]] = written;
    }
  }
}

template <int K, int MODE>
void run(const char* name, const uint4* buf, uint32_t nrows, const uint4* dir, uint32_t dmask, uint32_t n, unsigned long long* sink) {
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 3; rep++) {
    OK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<K, MODE>), dim3(16384), dim3(256), 0, 0, buf, nrows, dir, dmask, n, sink);
    OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1));
    float ms; OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("  %-52s %8.3f ms  %6.2f TB/s\n", name, best, (double)n * 2048 / (best * 1e-3) / 1e12);
}

int main() {
  const uint32_t nrows = 13000000, n = 13000000;
  uint4* buf; OK(hipMalloc(&buf, (size_t)nrows * 2048));
  OK(hipMemset(buf, 1, (size_t)nrows * 2048));
  const uint32_t dslots = 1u << 25;
  uint4* dir; OK(hipMalloc(&dir, (size_t)dslots * 16)); OK(hipMemset(dir, 0, (size_t)dslots * 16));
  unsigned long long* sink; OK(hipMalloc(&sink, 8 + (size_t)n * 4)); OK(hipMemset(sink, 3, 8 + (size_t)n * 4));
  printf("random gathers of 13 M rows x 2 KiB out of %.1f GB (hipMalloc), one wave per row:\n", nrows * 2048.0 / 1e9);
  run<1, 0>("1 row in flight, computed index", buf, nrows, dir, dslots - 1, n, sink);
  run<2, 0>("2 rows in flight, computed index", buf, nrows, dir, dslots - 1, n, sink);
  run<4, 0>("4 rows in flight, computed index", buf, nrows, dir, dslots - 1, n, sink);
  run<8, 0>("8 rows in flight, computed index", buf, nrows, dir, dslots - 1, n, sink);
  run<2, 1>("2 rows in flight, index behind a random 16-B load", buf, nrows, dir, dslots - 1, n, sink);
  run<4, 1>("4 rows in flight, index behind a random 16-B load", buf, nrows, dir, dslots - 1, n, sink);
  run<8, 1>("8 rows in flight, index behind a random 16-B load", buf, nrows, dir, dslots - 1, n, sink);
  run<2, 2>("2 rows in flight, id load -> 16-B load -> row", buf, nrows, dir, dslots - 1, n, sink);
  // the same rows in memory mapped like the library's arena: one VA reservation, 1 GiB physical chunks (hipMemCreate / hipMemMap)
  {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; OK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t want = ((size_t)nrows * 2048 + ((size_t)1 << 30) - 1) >> 30 << 30;
    void* va = nullptr; OK(hipMemAddressReserve(&va, want, (size_t)2 << 20, nullptr, 0));
    hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t off = 0; off < want; off += (size_t)1 << 30) {
      hipMemGenericAllocationHandle_t h; OK(hipMemCreate(&h, (size_t)1 << 30, &prop, 0));
      OK(hipMemMap((char*)va + off, (size_t)1 << 30, 0, h, 0));
    }
    OK(hipMemSetAccess(va, want, &ad, 1));
    OK(hipMemset(va, 1, want));
    printf("the same out of a VMM mapping (granularity reported %zu, 1 GiB chunks):\n", gran);
    run<2, 0>("2 rows in flight, computed index", (const uint4*)va, nrows, dir, dslots - 1, n, sink);
    run<2, 1>("2 rows in flight, index behind a random 16-B load", (const uint4*)va, nrows, dir, dslots - 1, n, sink);
  }
  {
    uint64_t* offsets; OK(hipMalloc(&offsets, ((size_t)n + 1) * 8));
    std::vector<uint64_t> ho(n + 1);
    for (uint32_t i = 0; i <= n; i++) ho[i] = (uint64_t)i * 161;                 // room for 161 pairs per row (~115 are written): odd -> half the rows start 8 B off a 16-byte boundary
    OK(hipMemcpy(offsets, ho.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
    uint64_t* ret; OK(hipMalloc(&ret, (size_t)n * 161 * 8));
    uint32_t* counts; OK(hipMalloc(&counts, (size_t)n * 4));
    uint32_t* xs = reinterpret_cast<uint32_t*>(sink + 1);
    printf("k_getrow's whole row (id -> slot -> 256 cells -> offsets -> compaction -> pairs), hipMalloc buffer:\n");
    for (int st = 0; st < 3; st++) {
      hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
      float best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        OK(hipEventRecord(e0));
        if (st == 2) hipLaunchKernelGGL((k_full<2>), dim3(16384), dim3(256), 0, 0, buf, nrows, dir, dslots - 1, n, xs, offsets, ret, counts);
        else if (st) hipLaunchKernelGGL((k_full<1>), dim3(16384), dim3(256), 0, 0, buf, nrows, dir, dslots - 1, n, xs, offsets, ret, counts);
        else hipLaunchKernelGGL((k_full<0>), dim3(16384), dim3(256), 0, 0, buf, nrows, dir, dslots - 1, n, xs, offsets, ret, counts);
        OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1));
        float ms; OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("  %-52s %8.3f ms\n", st == 2 ? "pairs staged in LDS, 16-byte stores" : st ? "with the pair stores (~115 x 8 B per row)" : "without the stores", best);
    }
  }
  return 0;
}
