// kernels/rows.hpp -- rowlen / getrow and the fused CF-recommender read / write paths.
// A fragment of smx_kernels.hpp (round 5: the 4 500-line header split by concern, no kernel changed): included there, in order,
// INSIDE namespace smx; not a header of its own.

// ---- rowlen / getrow ------------------------------------------------------------

// src/smatrix.c:212-223: rmap->used, 0 for an absent row
__global__ __launch_bounds__(256) void k_rowlen(DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n,
                                                const uint32_t* __restrict__ xs,
                                                uint32_t* __restrict__ out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint4 s;
  DirSlot* d = dir_find(dir, dmask, xs[t], &s);
  uint32_t len = d ? s.w : 0;
  if (d && s.z && meta_lg(s.x) >= BIG_LG) len += subs_sum(row_subs(arena, s.z, meta_lg(s.x)));
  out[t] = len;
}

// src/smatrix.c:189-210: the row's table is scanned in slot order and the non-empty cells are
// compacted (ballot + prefix popcount keeps slot order).  Row r may receive at most
// offsets[r+1]-offsets[r] pairs; counts[r] = pairs written.
//   k_getrow      one wave per row, 128 cells (1 KiB) per step with 16-byte loads; rows of more
//                 than GETROW_WAVE_MAX cells are only noted down in `big`
//   k_getrow_big  one 1024-lane workgroup per noted row -- per 32768-cell SEGMENT of a giant one --, 2048 cells per step
constexpr uint32_t GETROW_WAVE_MAX = 8192;

__device__ inline uint32_t getrow_cap(const uint64_t* offsets, uint32_t r) {
  const uint64_t c = offsets[r + 1] - offsets[r];
  return c > 0xffffffffull ? 0xffffffffu : (uint32_t)c;
}

template <int AHEAD = 2, bool XCD = true, int DBG = 0>     // DBG: measurement variants only (1: no pair stores, 2: no cell loads)
__global__ __launch_bounds__(256) void k_getrow(DirSlot* dir, uint32_t dmask, uint8_t* arena,
                                                uint32_t n, const uint32_t* __restrict__ xs,
                                                const uint64_t* __restrict__ offsets,
                                                uint64_t* __restrict__ ret,
                                                uint32_t* __restrict__ counts, uint32_t* big) {
  // Workgroups are dealt to the 8 XCDs round robin (MI355X_MICROARCH.md), each XCD with its own L2.  Consecutive rows of
  // the request write consecutive output ranges whose ends share cache lines: numbered naively, the four rows of
  // workgroup b and those of b + 1 meet in a line that two L2s each hold half of, and both halves reach memory as
  // partial-line writes.  So workgroups are RENUMBERED: XCD x takes the virtual workgroups [x * G/8, (x+1) * G/8), a
  // contiguous range of rows per sweep, and neighbours' partial lines merge in its L2.
  const uint32_t G = gridDim.x;
  const uint32_t vb = XCD && (G & 7u) == 0 ? (blockIdx.x & 7u) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  uint32_t wave = (vb * blockDim.x + threadIdx.x) >> 6;
  uint32_t lane = threadIdx.x & 63;
  uint32_t nwaves = (G * blockDim.x) >> 6;
  const uint64_t lt = (1ull << lane) - 1;
  // A row is a chain of dependent accesses (id -> directory slot -> cells -> pairs out) and a CF-shaped
  // row is only 1-2 KiB: a wave that walks one row at a time spends its life waiting.  Two rows are in
  // flight per wave instead: both directory slots are requested together, then ALL cells of both rows up to 512 per
  // row (four 1 KiB steps each: every load of a CF row is in flight before the first one is consumed -- round 2 fetched
  // the second KiB of a 256-cell row only after the first had been compacted).
  struct Row {
    bool live, scan;
    uint32_t r, size, cap, written;
    uint64_t off;
    const uint4* cells;
  };
  auto step = [&](Row& w, const uint4 c) {           // compacts the 128 cells held in c (slot order)
    const bool ne0 = (c.x | c.y) != 0, ne1 = (c.z | c.w) != 0;
    const uint64_t m0 = __ballot(ne0), m1 = __ballot(ne1);
    uint32_t rank = w.written + (uint32_t)__popcll(m0 & lt) + (uint32_t)__popcll(m1 & lt);
    if (DBG != 1 && ne0 && rank < w.cap) ret[w.off + rank] = pack_cell(c.x, c.y);
    rank += ne0;
    if (DBG != 1 && ne1 && rank < w.cap) ret[w.off + rank] = pack_cell(c.z, c.w);
    w.written += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
  };
  auto fetch = [&](const Row& w, uint32_t p0) -> uint4 {
    const uint32_t p = p0 + 2 * lane;
    if (DBG == 2) return p < w.size && (p & 3u) ? make_uint4(p, 1, 0, 0) : make_uint4(0, 0, 0, 0);
    return p < w.size ? w.cells[p >> 1] : make_uint4(0, 0, 0, 0);
  };
  for (uint32_t r0 = wave; r0 < n; r0 += 2 * nwaves) {
    Row w[2];
    uint32_t X[2], h[2];
    uint4 s[2];
    for (int k = 0; k < 2; k++) {
      w[k].r = r0 + k * nwaves;
      w[k].live = w[k].r < n;
      w[k].scan = false;
      w[k].written = 0;
      w[k].size = 0;
      X[k] = w[k].live ? xs[w[k].r] : 0u;
      h[k] = fmix32(X[k]) & dmask;
    }
    for (int k = 0; k < 2; k++) s[k] = *reinterpret_cast<const uint4*>(&dir[h[k]]);     // both in flight
    for (int k = 0; k < 2; k++) {
      if (!w[k].live) continue;
      while ((s[k].x & META_USED) && s[k].y != X[k]) {                                   // rare: probe on
        h[k] = (h[k] + 1) & dmask;
        s[k] = *reinterpret_cast<const uint4*>(&dir[h[k]]);
      }
      if (!(s[k].x & META_USED) || s[k].z == 0) continue;                                // no such row: 0 pairs
      w[k].size = 1u << meta_lg(s[k].x);
      if (w[k].size > GETROW_WAVE_MAX) {
        if (lane == 0) big[1 + atomicAdd(&big[0], 1u)] = w[k].r;
        w[k].live = false;                                                               // k_getrow_big writes its count
        w[k].size = 0;
        continue;
      }
      w[k].cells = reinterpret_cast<const uint4*>(row_cells(arena, s[k].z));
      w[k].scan = true;
    }
    // FAST PATH (wave-uniform): both rows are there and have at most 256 cells -- the CF shape.  Straight-line code: four
    // 1 KiB loads, the four offsets, then compaction and stores, nothing data-dependent between the loads' issue and
    // their first use.  (Round 3: tools/probe/row_gather.cpp does exactly this in 7.8 ms for 13 M rows on a box where
    // the general loop below takes 10.0.)
    if (AHEAD >= 2 && w[0].scan && w[1].scan && w[0].size <= 256 && w[1].size <= 256) {
      uint4 c[2][2];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        c[k][0] = fetch(w[k], 0);
        c[k][1] = fetch(w[k], 128);
      }
      uint64_t o0[2], o1[2];
#pragma unroll
      for (int k = 0; k < 2; k++) { o0[k] = offsets[w[k].r]; o1[k] = offsets[w[k].r + 1]; }
#pragma unroll
      for (int k = 0; k < 2; k++) {
        w[k].off = o0[k];
        const uint64_t cc = o1[k] - o0[k];
        w[k].cap = cc > 0xffffffffull ? 0xffffffffu : (uint32_t)cc;
        step(w[k], c[k][0]);
        step(w[k], c[k][1]);
        if (lane == 0) counts[w[k].r] = min(w[k].written, w[k].cap);
      }
      continue;
    }
    for (int k = 0; k < 2; k++) {
      if (w[k].scan) {
        w[k].off = offsets[w[k].r];
        w[k].cap = getrow_cap(offsets, w[k].r);
      }
    }
    uint4 c0[2];
    for (int k = 0; k < 2; k++) c0[k] = w[k].scan ? fetch(w[k], 0) : make_uint4(0, 0, 0, 0);   // both in flight
    for (int k = 0; k < 2; k++) {
      if (w[k].scan) {
        step(w[k], c0[k]);
        for (uint32_t p0 = 128; p0 < w[k].size && w[k].written < w[k].cap; p0 += 256) {
          const uint4 a = fetch(w[k], p0), b2 = fetch(w[k], p0 + 128);                  // two steps in flight
          step(w[k], a);
          if (p0 + 128 < w[k].size && w[k].written < w[k].cap) step(w[k], b2);
        }
        if (w[k].written > w[k].cap) w[k].written = w[k].cap;
      }
      if (w[k].live && lane == 0) counts[w[k].r] = w[k].written;
    }
  }
}

// Rows noted down by k_getrow are cut into SEGMENTS of GETROW_SEG cells, one workgroup each, so that one giant row
// (config 2: 2 M slots, 16 MB) is read by as many workgroups as it has segments instead of by one:
//   k_getrow_plan       seg_start[b] = first segment id of noted row b (rows of up to 2 segments' worth stay whole)
//   k_getrow_big<true>  per segment of a CUT row: the number of non-empty cells -> seg_cnt[]
//   k_getrow_big<false> per segment: the pairs, in slot order, at  offset + (pairs in the segments before it)
// A row that stays whole needs no count pass: its single workgroup compacts from rank 0 as before.
constexpr uint32_t GETROW_SEG = 32768;

__device__ inline uint32_t getrow_nseg(uint32_t size) { return size >= 2 * GETROW_SEG ? size / GETROW_SEG : 1u; }

// `budget`: segments the caller's seg_cnt array has room for BEYOND one per noted row.  A batch may name one giant row
// many times (a hot item requested by many callers): every occurrence is noted and would want all of its segments, so
// the total is not bounded by the arena's size.  Occurrences are cut while the budget lasts (in list order, an
// occurrence that does not fit does not consume); the others stay whole -- one workgroup walks the row, as before
// the segmentation -- and need no count entry.
__global__ __launch_bounds__(1024) void k_getrow_plan(DirSlot* dir, uint32_t dmask, const uint32_t* __restrict__ xs,
                                                      const uint32_t* big, uint32_t* seg_start, uint32_t budget) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t s_base, s_extra;
  const uint32_t nbig = big[0];
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) { s_base = 0; s_extra = 0; }
  __syncthreads();
  auto block_scan = [&](uint32_t v, uint32_t carry, uint32_t* total) -> uint32_t {   // exclusive prefix over the workgroup + carry
    uint32_t incl = v;
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
      if (lane >= d) incl += o;
    }
    __syncthreads();
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t before = carry, tot = 0;
    for (uint32_t i = 0; i < 16; i++) { const uint32_t t = wsum[i]; if (i < w) before += t; tot += t; }
    *total = tot;
    return before + incl - v;
  };
  for (uint32_t b0 = 0; b0 < nbig; b0 += 1024) {
    const uint32_t b = b0 + threadIdx.x;
    uint32_t want = 0;
    if (b < nbig) {
      uint4 s;
      dir_find(dir, dmask, xs[big[1 + b]], &s);
      want = getrow_nseg(1u << meta_lg(s.x));
    }
    uint32_t tot_e = 0, tot_v = 0;
    const uint32_t extra = want ? want - 1u : 0u;
    const uint32_t ebefore = block_scan(extra, s_extra, &tot_e);
    const uint32_t v = (uint64_t)ebefore + extra <= budget ? want : (want ? 1u : 0u);
    const uint32_t start = block_scan(v, s_base, &tot_v);
    if (b < nbig) seg_start[b] = start;
    __syncthreads();
    if (threadIdx.x == 0) { s_base += tot_v; s_extra = (uint32_t)min((uint64_t)s_extra + tot_e, (uint64_t)0xffffffffu); }
    __syncthreads();
  }
  if (threadIdx.x == 0) seg_start[nbig] = s_base;
}

template <bool COUNT>
__global__ __launch_bounds__(1024) void k_getrow_big(DirSlot* dir, uint32_t dmask, uint8_t* arena,
                                                     const uint32_t* __restrict__ xs,
                                                     const uint64_t* __restrict__ offsets,
                                                     uint64_t* __restrict__ ret,
                                                     uint32_t* __restrict__ counts, const uint32_t* big,
                                                     const uint32_t* __restrict__ seg_start, uint32_t* seg_cnt) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t s_written;
  const uint32_t nbig = big[0];
  const uint32_t nseg_all = nbig ? seg_start[nbig] : 0;
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t lt = (1ull << lane) - 1;
  auto block_sum = [&](uint32_t v) -> uint32_t {          // sum over the workgroup, to every lane
    for (uint32_t d = 32; d; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    __syncthreads();
    if (lane == 0) wsum[w] = v;
    __syncthreads();
    uint32_t t = 0;
    for (uint32_t i = 0; i < 16; i++) t += wsum[i];
    __syncthreads();
    return t;
  };
  for (uint32_t t = blockIdx.x; t < nseg_all; t += gridDim.x) {
    uint32_t lo = 0, hi = nbig;                           // the noted row whose segments include t
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg_start[mid] <= t) lo = mid; else hi = mid; }
    const uint32_t b = lo, first = seg_start[b], nseg = seg_start[b + 1] - first, g = t - first;
    if (COUNT && nseg == 1) continue;
    const uint32_t r = big[1 + b];
    uint4 s;
    dir_find(dir, dmask, xs[r], &s);
    const uint32_t size = 1u << meta_lg(s.x);
    const uint4* cells = reinterpret_cast<const uint4*>(row_cells(arena, s.z));
    const uint32_t p_begin = nseg == 1 ? 0u : g * GETROW_SEG, p_end = nseg == 1 ? size : p_begin + GETROW_SEG;
    if (COUNT) {
      uint32_t c = 0;
      for (uint32_t p0 = p_begin; p0 < p_end; p0 += 2048) {
        const uint4 q = cells[(p0 >> 1) + threadIdx.x];
        c += ((q.x | q.y) != 0) + ((q.z | q.w) != 0);
      }
      c = block_sum(c);
      if (threadIdx.x == 0) seg_cnt[t] = c;
      continue;
    }
    const uint64_t off = offsets[r];
    const uint32_t cap = getrow_cap(offsets, r);
    uint32_t before_me = 0;
    if (nseg > 1) {
      uint32_t mine = 0, all = 0;
      for (uint32_t i = threadIdx.x; i < nseg; i += 1024) { const uint32_t c = seg_cnt[first + i]; all += c; if (i < g) mine += c; }
      before_me = block_sum(mine);
      if (g == 0) {                                         // the row's first segment also reports the row's count
        all = block_sum(all);
        if (threadIdx.x == 0) counts[r] = all > cap ? cap : all;
      }
    }
    if (threadIdx.x == 0) s_written = before_me;
    __syncthreads();
    for (uint32_t p0 = p_begin; p0 < p_end; p0 += 2048) {
      const uint32_t written = s_written;
      if (written >= cap) break;
      const uint4 c = cells[(p0 >> 1) + threadIdx.x];            // size is a multiple of 2048 here
      const bool ne0 = (c.x | c.y) != 0, ne1 = (c.z | c.w) != 0;
      const uint64_t m0 = __ballot(ne0), m1 = __ballot(ne1);
      if (lane == 0) wsum[w] = (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
      __syncthreads();
      uint32_t before = 0, total = 0;
      for (uint32_t i = 0; i < 16; i++) { const uint32_t v = wsum[i]; if (i < w) before += v; total += v; }
      uint32_t rank = written + before + (uint32_t)__popcll(m0 & lt) + (uint32_t)__popcll(m1 & lt);
      if (ne0 && rank < cap) ret[off + rank] = pack_cell(c.x, c.y);
      rank += ne0;
      if (ne1 && rank < cap) ret[off + rank] = pack_cell(c.z, c.w);
      __syncthreads();
      if (threadIdx.x == 0) s_written = written + total;
      __syncthreads();
    }
    if (nseg == 1 && threadIdx.x == 0) counts[r] = s_written > cap ? cap : s_written;
    __syncthreads();
  }
}

// ---- CF-recommender read path, fused (examples/cf_recommender.c:50-86) ---------------------------
// For item a: total = get(a,0); every (b, cc) of getrow(a) scores  cc / (sqrt(total)*sqrt(get(b,0)))
// with the example's guards (b_total 0 -> 1; den == 0 -> 0; num > den -> 0), all in double.  One wave
// per item: the row scan of k_getrow, and each lane that holds a neighbour does that neighbour's
// get(b,0) itself -- 64..128 independent lookups in flight per wave instead of one call per neighbour.
// Output in slot order like the example's loop; at most offsets[i+1]-offsets[i] neighbours per item.
__global__ __launch_bounds__(256) void k_cf_neighbors(DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n,
                                                      const uint32_t* __restrict__ items,
                                                      const uint64_t* __restrict__ offsets,
                                                      uint32_t* __restrict__ ids, double* __restrict__ scores,
                                                      uint32_t* __restrict__ counts) {
  uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t lane = threadIdx.x & 63;
  uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint64_t lt = (1ull << lane) - 1;
  for (uint32_t r = wave; r < n; r += nwaves) {
    uint4 s;
    DirSlot* d = dir_find(dir, dmask, items[r], &s);
    uint32_t written = 0;
    if (d && s.z != 0) {
      bool dummy = false;
      const uint32_t a_total = apply_one<OP_GET>(dir, dmask, arena, items[r], 0u, 0u, &dummy);
      const double sa = sqrt((double)a_total);
      const uint32_t size = 1u << meta_lg(s.x);
      const uint64_t off = offsets[r];
      const uint32_t cap = getrow_cap(offsets, r);
      const uint64_t* cells = row_cells(arena, s.z);
      for (uint32_t p0 = 0; p0 < size && written < cap; p0 += 64) {
        const uint32_t p = p0 + lane;
        const uint64_t c = p < size ? cells[p] : 0;
        const bool ne = c != 0;
        const uint64_t m = __ballot(ne);
        const uint32_t rank = written + (uint32_t)__popcll(m & lt);
        if (ne && rank < cap) {
          uint32_t b_total = apply_one<OP_GET>(dir, dmask, arena, cell_key(c), 0u, 0u, &dummy);
          if (b_total == 0) b_total = 1;
          const double num = (double)cell_val(c);
          const double den = sa * sqrt((double)b_total);
          double score = 0.0;
          if (den != 0.0 && !(num > den)) score = num / den;
          ids[off + rank] = cell_key(c);
          scores[off + rank] = score;
        }
        written += (uint32_t)__popcll(m);
      }
      if (written > cap) written = cap;
    }
    if (lane == 0) counts[r] = written;
  }
}

// ---- CF-recommender read path, the k best neighbours only ---------------------------------------------------
// Same candidates and the same score as k_cf_neighbors (every entry the example's loop would print, the (0,total) entry
// included), but only the k <= 64 best per item leave the kernel: best score first, equal scores in table slot order.
// One wave per item; lane i holds the i-th best so far.  Per 64 cells: the candidates are sorted across the wave (bitonic,
// shuffles only), merged with the running list (the better of A[i] and B[63-i] is a bitonic sequence of the best 64 of
// both; six more stages sort it), and a step none of whose candidates beats the current k-th is skipped -- which is
// nearly every step of a long row.
struct CfCand {
  long long key;      // the score's bit pattern (scores are >= 0: ordered like signed integers); LLONG_MIN = no candidate
  uint32_t slot, id;
};
__device__ __forceinline__ bool cf_better(const CfCand& a, const CfCand& b) {
  return a.key > b.key || (a.key == b.key && a.slot < b.slot);
}
__device__ __forceinline__ CfCand cf_shfl_xor(const CfCand& v, int j) {
  CfCand o;
  o.key = ((long long)__shfl_xor((int)(v.key >> 32), j) << 32) | (uint32_t)__shfl_xor((int)v.key, j);
  o.slot = (uint32_t)__shfl_xor((int)v.slot, j);
  o.id = (uint32_t)__shfl_xor((int)v.id, j);
  return o;
}
__device__ __forceinline__ CfCand cf_shfl(const CfCand& v, int src) {
  CfCand o;
  o.key = ((long long)__shfl((int)(v.key >> 32), src) << 32) | (uint32_t)__shfl((int)v.key, src);
  o.slot = (uint32_t)__shfl((int)v.slot, src);
  o.id = (uint32_t)__shfl((int)v.id, src);
  return o;
}
// stages j = from, from/2, .. 1 of a bitonic network over the wave's 64 lanes, best first
__device__ __forceinline__ void cf_merge_stages(CfCand& v, uint32_t lane, uint32_t from) {
  for (uint32_t j = from; j; j >>= 1) {
    const CfCand o = cf_shfl_xor(v, (int)j);
    const bool want_better = (lane & j) == 0;              // the lower lane of a pair keeps the better one
    if (cf_better(o, v) == want_better) v = o;
  }
}
__device__ __forceinline__ void cf_sort64(CfCand& v, uint32_t lane) {
  for (uint32_t k2 = 2; k2 <= 64; k2 <<= 1) {
    for (uint32_t j = k2 >> 1; j; j >>= 1) {
      const CfCand o = cf_shfl_xor(v, (int)j);
      const bool down = (lane & k2) == 0 || k2 == 64;     // blocks alternate direction; the last pass is best-first
      const bool want_better = ((lane & j) == 0) == down;
      if (cf_better(o, v) == want_better) v = o;
    }
  }
}

__global__ __launch_bounds__(256) void k_cf_topk(DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n,
                                                 const uint32_t* __restrict__ items, uint32_t k,
                                                 uint32_t* __restrict__ ids, double* __restrict__ scores,
                                                 uint32_t* __restrict__ counts) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  constexpr long long NONE = (long long)0x8000000000000000ull;
  for (uint32_t r = wave; r < n; r += nwaves) {
    uint4 s;
    DirSlot* d = dir_find(dir, dmask, items[r], &s);
    CfCand top{NONE, 0xffffffffu, 0u};
    if (d && s.z != 0) {
      bool dummy = false;
      const uint32_t a_total = apply_one<OP_GET>(dir, dmask, arena, items[r], 0u, 0u, &dummy);
      const double sa = sqrt((double)a_total);
      const uint32_t size = 1u << meta_lg(s.x);
      const uint64_t* cells = row_cells(arena, s.z);
      for (uint32_t p0 = 0; p0 < size; p0 += 64) {
        const uint32_t p = p0 + lane;
        const uint64_t c = p < size ? cells[p] : 0;
        CfCand cand{NONE, p, cell_key(c)};
        if (c != 0) {
          uint32_t b_total = apply_one<OP_GET>(dir, dmask, arena, cell_key(c), 0u, 0u, &dummy);
          if (b_total == 0) b_total = 1;
          const double num = (double)cell_val(c);
          const double den = sa * sqrt((double)b_total);
          double score = 0.0;
          if (den != 0.0 && !(num > den)) score = num / den;
          cand.key = __double_as_longlong(score);
        }
        const CfCand kth = cf_shfl(top, (int)k - 1);
        if (!__any(cand.key != NONE && cf_better(cand, kth))) continue;
        cf_sort64(cand, lane);
        const CfCand rev = cf_shfl(cand, 63 - (int)lane);
        if (cf_better(rev, top)) top = rev;
        cf_merge_stages(top, lane, 32);
      }
    }
    const bool have = lane < k && top.key != NONE;
    if (have) {
      ids[(uint64_t)r * k + lane] = top.id;
      scores[(uint64_t)r * k + lane] = __longlong_as_double(top.key);
    }
    const uint64_t m = __ballot(have);
    if (lane == 0) counts[r] = (uint32_t)__popcll(m);
  }
}

// ---- CF-recommender write path (examples/cf_recommender.c:36-47) ------------------------------------------
// A session of L ids is L*L incr ops: op r of the session has n = r / L, i = r % L and is (ids[n], 0, +1) when i == n,
// (ids[n], ids[i], +1) otherwise.  op_off[s] = sum of L*L over the sessions before s.  One lane per op of the range
// [t0, t0 + count): a binary search for its session, then the pair.
__global__ __launch_bounds__(256) void k_cf_expand(uint64_t t0, uint32_t count, uint32_t n_sessions,
                                                   const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ ids,
                                                   const uint64_t* __restrict__ op_off,
                                                   uint32_t* __restrict__ xs, uint32_t* __restrict__ ys, uint32_t* __restrict__ vs) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const uint64_t t = t0 + k;
  uint32_t lo = 0, hi = n_sessions;                      // the last session with op_off[s] <= t
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (op_off[mid] <= t) lo = mid; else hi = mid; }
  const uint64_t first = offsets[lo], L = offsets[lo + 1] - first, r = t - op_off[lo];
  const uint64_t n = r / L, i = r - n * L;
  xs[k] = ids[first + n];
  ys[k] = i == n ? 0u : ids[first + i];
  vs[k] = 1u;
}
