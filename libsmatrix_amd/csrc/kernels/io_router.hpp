// kernels/io_router.hpp -- persistence kernels (dirty rows, packing), debug helpers, the partition / gather kernels of the multi-GPU router.
// A fragment of smx_kernels.hpp (round 5: the 4 500-line header split by concern, no kernel changed): included there, in order,
// INSIDE namespace smx; not a header of its own.

// ---- persistence: dirty rows (src/smatrix.c:418-425 rmap_sync_defer, :929-960 the IO thread's queue) --------
// k_dirty_collect: every directory slot marked META_DIRTY is copied to `out` and unmarked (one list reservation per
// workgroup).  With all != 0 every row is taken (first write of a file, compaction).
// budget (bytes of row cells; ~0: none): the flush that snapshots its rows on the device takes only so much at a time.
// count[0] = rows listed, count[1] = "more are waiting", count[2..3] = bytes reserved so far (one 64-bit word).
// A workgroup reserves its rows' bytes with one add on that word; a share that STARTS beyond the budget is left as
// it is -- rows stay marked, count[1] is set -- so one call takes the budget plus at most one workgroup's rows.
__global__ __launch_bounds__(256) void k_dirty_collect(DirSlot* dir, uint32_t dir_size, uint32_t all, DirSlot* out,
                                                       uint32_t cap, uint32_t* count, unsigned long long budget) {
  __shared__ uint32_t l_n, l_base, l_ok;
  __shared__ unsigned long long l_bytes;
  for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < dir_size; i0 += gridDim.x * blockDim.x) {     // block-uniform
    if (threadIdx.x == 0) { l_n = 0; l_bytes = 0; l_ok = 1; }
    __syncthreads();
    const uint32_t i = i0 + threadIdx.x;
    DirSlot d = {0, 0, 0, 0};
    bool take = false;
    if (i < dir_size) {
      d = dir[i];
      take = (d.meta & META_USED) && d.base != 0 && (all || (d.meta & META_DIRTY));
    }
    uint32_t rank = 0;
    if (take) {
      rank = atomicAdd(&l_n, 1u);
      if (budget != ~0ull) atomicAdd(&l_bytes, 16ull + (8ull << meta_lg(d.meta)));
    }
    __syncthreads();
    if (threadIdx.x == 0 && l_n) {
      if (budget != ~0ull) {
        const unsigned long long before = atomicAdd(reinterpret_cast<unsigned long long*>(count + 2), l_bytes);   // (count + 2 is 8-byte aligned)
        if (before >= budget) { l_ok = 0; count[1] = 1; }
      }
      if (l_ok) l_base = atomicAdd(count, l_n);
    }
    __syncthreads();
    if (take && l_ok) {
      if (d.meta & META_DIRTY) dir[i].meta = d.meta & ~META_DIRTY;
      const uint32_t at = l_base + rank;
      if (at < cap) { d.meta &= ~META_DIRTY; out[at] = d; }
    }
    __syncthreads();
  }
}

// k_pack_rows: row tables -> a staging buffer laid out like the FILE (RMAP block = 8 x 0x23, u64 n_slots, the
// raw cells: src/smatrix.c:57-70), so that a window of it goes out with one pwrite.  One wave per row, 16-byte
// moves; big rows add their sub-counter sums nowhere (the file holds cells only).
struct PackRow {
  uint32_t base;       // arena unit of the row's cells
  uint32_t lg;         // log2(cells)
  uint64_t out;        // byte offset in the staging buffer of the block's first byte (header if with_head)
  uint32_t with_head;  // 1: header + cells (a new block), 0: cells only (rewrite in place)
  uint32_t pad;
};
__global__ __launch_bounds__(256) void k_pack_rows(uint32_t n, const PackRow* rows, const uint8_t* arena, uint8_t* stage) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t r = wave; r < n; r += nwaves) {
    const PackRow w = rows[r];
    const uint4* src = reinterpret_cast<const uint4*>(arena + (uint64_t)w.base * UNIT_BYTES);
    uint8_t* dst = stage + w.out;
    if (w.with_head) {
      if (lane == 0) {
        uint64_t* h = reinterpret_cast<uint64_t*>(dst);
        h[0] = 0x2323232323232323ull;
        h[1] = 1ull << w.lg;
      }
      dst += 16;
    }
    uint4* d4 = reinterpret_cast<uint4*>(dst);                     // 16-byte aligned: offsets are multiples of 8 + 16
    const uint32_t n16 = (8u << w.lg) / 16u;
    for (uint32_t i = lane; i < n16; i += 64) d4[i] = src[i];
  }
}

// ---- debug / export helpers -------------------------------------------------------
__global__ void k_row_info(DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t x, uint32_t* out4) {
  uint4 s;
  DirSlot* d = dir_find(dir, dmask, x, &s);
  out4[0] = d ? 1 : 0;
  out4[1] = d ? 1u << meta_lg(s.x) : 0;
  out4[2] = d ? s.w : 0;
  out4[3] = d ? s.z : 0;
  if (d && s.z && meta_lg(s.x) >= BIG_LG) out4[2] += subs_sum(row_subs(arena, s.z, meta_lg(s.x)));
}

// ---- row-hash sharding over the GPUs of a node (include/smatrix_shard.h) --------------------
// owner(x) = floor(fmix32(x ^ salt) * nshards / 2^32): the HIGH bits of a differently salted mix,
// so that the rows of one shard still spread over all low-bit buckets of its local directory.
__host__ __device__ inline uint32_t shard_mix(uint32_t h) {
  h ^= 0x9E3779B9u;
  h ^= h >> 16; h *= 0x85ebca6bU; h ^= h >> 13; h *= 0xc2b2ae35U; h ^= h >> 16;
  return h;
}
__host__ __device__ inline uint32_t shard_of(uint32_t x, uint32_t nshards) {
  return (uint32_t)(((uint64_t)shard_mix(x) * nshards) >> 32);
}

constexpr uint32_t MAX_SHARDS = 64;

// Placement (libsmatrix_amd/sharded.py plans it, include/smatrix_shard.h states the layout):
//   cuts  : nshards - 1 ascending cut points of the 32-bit hash space; shard r owns the rows with
//           cuts[r-1] <= shard_mix(x) < cuts[r]  (cuts[-1] = 0, cuts[nshards-1] = 2^32).  NULL = equal ranges.
//   place : the few hot rows that are placed one by one: open addressing over `slots` (a power of two
//           <= PLACE_MAX_SLOTS) entries {x, owner + 1}; slot of x = fmix32(x) & (slots - 1), linear
//           probing, owner + 1 == 0 marks an empty slot.
constexpr uint32_t PLACE_MAX_SLOTS = 1024;
struct PlaceLds {
  uint2 tab[PLACE_MAX_SLOTS];
  uint32_t cuts[MAX_SHARDS];
};
__device__ inline void place_stage(PlaceLds& l, const uint2* place, uint32_t slots, const uint32_t* cuts, uint32_t nshards) {
  for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) l.tab[i] = place[i];
  if (cuts && threadIdx.x < nshards - 1u) l.cuts[threadIdx.x] = cuts[threadIdx.x];
}
__device__ inline uint32_t owner_of(uint32_t x, uint32_t nshards, const PlaceLds& l, uint32_t slots, bool have_cuts) {
  if (slots) {
    for (uint32_t i = fmix32(x) & (slots - 1u);; i = (i + 1u) & (slots - 1u)) {
      const uint2 e = l.tab[i];
      if (e.y == 0) break;
      if (e.x == x) return e.y - 1u;
    }
  }
  if (!have_cuts) return shard_of(x, nshards);
  const uint32_t h = shard_mix(x);
  uint32_t lo = 0, hi = nshards - 1u;             // owner = number of cut points <= h
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (l.cuts[mid] <= h) lo = mid + 1u; else hi = mid;
  }
  return lo;
}

// pass 1: per-shard op counts (LDS histogram per workgroup, one global atomic per shard per WG)
__global__ __launch_bounds__(256) void k_part_count(uint32_t n, const uint32_t* __restrict__ xs,
                                                    uint32_t nshards, unsigned long long* counts,
                                                    const uint2* place, uint32_t place_slots, const uint32_t* cuts) {
  __shared__ uint32_t h[MAX_SHARDS];
  __shared__ PlaceLds l_place;
  if (threadIdx.x < MAX_SHARDS) h[threadIdx.x] = 0;
  place_stage(l_place, place, place_slots, cuts, nshards);
  __syncthreads();
  // one LDS atomic per distinct owner and WAVE (ballots): with a handful of shards every lane of a wave names one of a
  // few counters, and 64 same-address LDS atomics serialise (round 3, 2^24 ops, one shard: 63 us before)
  const uint32_t lane = __lane_id();
  for (uint64_t i064 = (uint64_t)blockIdx.x * blockDim.x; i064 < n; i064 += (uint64_t)gridDim.x * blockDim.x) {        // block-uniform
    const uint32_t i = (uint32_t)i064 + threadIdx.x;
    const bool live = i < n;
    const uint32_t o = live ? owner_of(xs[i], nshards, l_place, place_slots, cuts != nullptr) : 0u;
    uint64_t todo = __ballot(live);
    while (todo) {
      const uint32_t leader = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
      const uint32_t o0 = (uint32_t)__shfl((int)o, (int)leader);
      const uint64_t m = __ballot(live && o == o0);
      if (lane == leader) atomicAdd(&h[o0], (uint32_t)__popcll(m));
      todo &= ~m;
    }
  }
  __syncthreads();
  if (threadIdx.x < nshards && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// between the passes: counts -> exclusive offsets (the scatter's cursors), on the device so that the host
// waits once per partition instead of twice.  work[0..63] = counts (kept for the host), work[64..127] = cursors
__global__ void k_part_offsets(unsigned long long* work, uint32_t nshards) {
  unsigned long long run = 0;
  for (uint32_t i = 0; i < nshards; i++) {
    work[MAX_SHARDS + i] = run;
    run += work[i];
  }
}

// pass 2: scatter into shard-contiguous order.  cursors[] start at the exclusive offsets; a
// workgroup reserves its range per shard with one global atomic, lanes rank inside it in LDS.
// perm[i] = position of op i in the partitioned arrays (used to route results back).
constexpr uint32_t PART_OPT = 8;
__global__ __launch_bounds__(256) void k_part_scatter(
    uint32_t n, const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
    const uint32_t* __restrict__ vs, uint32_t nshards, unsigned long long* cursors,
    uint32_t* __restrict__ perm, uint32_t* __restrict__ xo, uint32_t* __restrict__ yo,
    uint32_t* __restrict__ vo, uint32_t* __restrict__ packed, const uint2* place, uint32_t place_slots,
    const uint32_t* cuts) {
  __shared__ uint32_t cnt[MAX_SHARDS];
  __shared__ unsigned long long base[MAX_SHARDS];
  __shared__ PlaceLds l_place;
  if (threadIdx.x < MAX_SHARDS) cnt[threadIdx.x] = 0;
  place_stage(l_place, place, place_slots, cuts, nshards);
  __syncthreads();
  const uint32_t tile0 = blockIdx.x * 256 * PART_OPT;
  uint32_t sh[PART_OPT], rk[PART_OPT], X[PART_OPT];
#pragma unroll
  for (uint32_t k = 0; k < PART_OPT; k++) {
    uint32_t i = tile0 + k * 256 + threadIdx.x;
    sh[k] = ~0u;
    const bool live = i < n;
    if (live) {
      X[k] = xs[i];
      sh[k] = owner_of(X[k], nshards, l_place, place_slots, cuts != nullptr);
    }
    // ranks inside the tile: one LDS atomic per distinct owner and wave, lanes rank themselves by ballot
    uint64_t todo = __ballot(live);
    while (todo) {
      const uint32_t leader = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
      const uint32_t o0 = (uint32_t)__shfl((int)sh[k], (int)leader);
      const uint64_t m = __ballot(live && sh[k] == o0);
      uint32_t base0 = 0;
      if (__lane_id() == leader) base0 = atomicAdd(&cnt[o0], (uint32_t)__popcll(m));
      base0 = (uint32_t)__shfl((int)base0, (int)leader);
      if (live && sh[k] == o0) rk[k] = base0 + (uint32_t)__popcll(m & ((1ull << __lane_id()) - 1ull));
      todo &= ~m;
    }
  }
  __syncthreads();
  if (threadIdx.x < nshards && cnt[threadIdx.x])
    base[threadIdx.x] = atomicAdd(&cursors[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < PART_OPT; k++) {
    uint32_t i = tile0 + k * 256 + threadIdx.x;
    if (sh[k] == ~0u) continue;
    uint32_t dst = (uint32_t)(base[sh[k]] + rk[k]);
    perm[i] = dst;
    if (packed) {                       // one {x,y[,v]} record per op: ONE collective moves it
      const uint32_t w = vs ? 3u : 2u;
      packed[(uint64_t)dst * w] = X[k];
      packed[(uint64_t)dst * w + 1] = ys[i];
      if (vs) packed[(uint64_t)dst * w + 2] = vs[i];
    } else {
      xo[dst] = X[k];
      yo[dst] = ys[i];
      if (vs) vo[dst] = vs[i];
    }
  }
}

// rows of this shard whose hash owner is another shard (the placement table is rebuilt from them when
// sharded files are reopened)
__global__ __launch_bounds__(256) void k_displaced_rows(const DirSlot* dir, uint32_t dir_size, uint32_t rank,
                                                        uint32_t nshards, uint32_t* out, uint32_t cap, uint32_t* count) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < dir_size; i += gridDim.x * blockDim.x) {
    const DirSlot d = dir[i];
    if ((d.meta & META_USED) && shard_of(d.x, nshards) != rank) {   // (equal ranges: files written without a placement)
      const uint32_t k = atomicAdd(count, 1u);
      if (k < cap) out[k] = d.x;
    }
  }
}

// records {x,y[,v]} -> separate arrays (what the op kernels read)
__global__ __launch_bounds__(256) void k_unpack(uint32_t n, uint32_t width, const uint32_t* __restrict__ packed,
                                                uint32_t* __restrict__ x, uint32_t* __restrict__ y,
                                                uint32_t* __restrict__ v) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  x[i] = packed[(uint64_t)i * width];
  y[i] = packed[(uint64_t)i * width + 1];
  if (width == 3) v[i] = packed[(uint64_t)i * width + 2];
}

__global__ __launch_bounds__(256) void k_gather(uint32_t n, const uint32_t* __restrict__ src,
                                                const uint32_t* __restrict__ perm,
                                                uint32_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[perm[i]];
}

__global__ __launch_bounds__(256) void k_gather2(uint32_t n, const uint32_t* __restrict__ src, const uint32_t* __restrict__ src2,
                                                 const uint32_t* __restrict__ perm, uint32_t* __restrict__ out, uint32_t* __restrict__ out2) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const uint32_t p = perm[i]; out[i] = src[p]; out2[i] = src2[p]; }
}
