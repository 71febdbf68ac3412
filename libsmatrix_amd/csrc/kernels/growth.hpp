// kernels/growth.hpp -- row doubling (in-LDS rehash, chunked passes, clustered rows on an occupancy bitmap), re-partitioning of big rows, directory growth.
// A fragment of smx_kernels.hpp (round 5: the 4 500-line header split by concern, no kernel changed): included there, in order,
// INSIDE namespace smx; not a header of its own.

// ---- growth -------------------------------------------------------------------
//
// smatrix_rmap_resize (src/smatrix.c:383-416) doubles the table and re-inserts
// every non-empty cell IN OLD SLOT ORDER.  The same final layout is produced in
// parallel by priority linear probing: a cell's priority is its old slot index,
// an arriving cell evicts a resident of lower priority (later old slot) and the
// evicted cell moves on.  The fixed point is unique and equals the sequential
// first-come-first-served layout (each cell sits in the first slot at/after its
// home not taken by an earlier cell).  While moving, a new cell holds
// {key, old_slot+1}; k_grow_finish swaps the index for the value.

// allocate the new block of every task -- from the stack of retired blocks of its size class where
// one is left (popped with one atomic per class and workgroup), else from the arena -- and assign
// the chunk ranges of the move/finish passes
// task_budget / arena_cap_units: what the host has made room for.  When the host has read prep's counters back it has
// sized everything for them and neither limit can bind; in the device-driven round (speculative chain) they are
// estimates, and a task that does not fit is REFUSED -- new_base 0: every later pass skips it, the commit takes the
// row's growth flag back, its ops stay deferred and the host-driven loop finishes them.
constexpr uint32_t CHUNK_NONE = 0xFFFFFFFFu;
// pend_ctl / pend_cap_keys / task_of (round 6, clustered matrices): every accepted task takes a bucket of old size / 2 keys in
// the round's key buffer (one reservation per workgroup on pend_ctl[1]; what does not fit gets none) and leaves its number at
// its directory slot's entry of task_of, where k_pend_group finds it
__device__ __forceinline__ void grow_plan_body(VGrid g, Ctl* ctl, GrowTask* tasks, uint64_t arena_cap_units, FreeLists fl,
                                               uint32_t task_budget, uint32_t chunk_cap, uint32_t* pend_ctl = nullptr, uint32_t pend_cap_keys = 0,
                                               uint32_t* task_of = nullptr) {
  __shared__ uint32_t l_pend, l_pend0, l_pend_ok;
  __shared__ uint32_t l_want[N_CLASSES], l_got[N_CLASSES];
  __shared__ int32_t l_top[N_CLASSES];
  // the two bump counters (chunk ranges, arena units) are reserved ONCE PER WORKGROUP, look-then-compare-and-swap so that
  // they never overshoot what the host has made room for (a per-task CAS loop is quadratic in the contenders: 10^5 tasks
  // of a young table took seconds); a workgroup whose share does not fit has all of that share refused
  __shared__ uint32_t l_chunks, l_chunk0, l_chunk_ok;
  __shared__ unsigned long long l_units, l_unit0;
  __shared__ uint32_t l_unit_ok;
  const uint32_t n = aload(&ctl->n_tasks);
  for (uint32_t t0 = g.bid * blockDim.x; t0 < n; t0 += g.nb * blockDim.x) {    // block-uniform
    if (threadIdx.x < N_CLASSES) l_want[threadIdx.x] = 0;
    if (threadIdx.x == 0) { l_chunks = 0; l_units = 0; l_chunk_ok = 1; l_unit_ok = 1; l_pend = 0; l_pend_ok = 0; }
    __syncthreads();
    const uint32_t t = t0 + threadIdx.x;
    const bool live = t < n;
    uint32_t cls = 0, rank = 0;
    bool refused = live && t >= task_budget;
    // chunked tasks (old size > 8192 cells: the new table has exactly twice the old one's 64-cell chunks) take their
    // range of the chunk -> task maps first
    const bool chunked = live && !refused && grow_kind(tasks[t].old_lg) == GROW_CHUNKED;
    uint32_t my_chunk = 0;
    if (chunked) my_chunk = atomicAdd(&l_chunks, 1u << (tasks[t].old_lg - 6));
    __syncthreads();
    if (threadIdx.x == 0 && l_chunks) {
      uint32_t cur = aload(&ctl->n_chunks);
      for (;;) {
        if ((uint64_t)cur + l_chunks > chunk_cap) { l_chunk_ok = 0; ctl->spec_failed = 1; break; }
        const uint32_t prev = atomicCAS(&ctl->n_chunks, cur, cur + l_chunks);
        if (prev == cur) { l_chunk0 = cur; break; }
        cur = prev;
      }
    }
    __syncthreads();
    if (live) {
      GrowTask& k = tasks[t];
      k.chunk0 = CHUNK_NONE;
      if (chunked) {
        if (l_chunk_ok) { k.chunk0 = l_chunk0 + my_chunk; k.chunk0_new = 2u * k.chunk0; }
        else refused = true;
      }
    }
    if (live && !refused) {
      cls = tasks[t].old_lg + 1 - ROW_FIRST_LG;
      rank = atomicAdd(&l_want[cls], 1u);
    }
    __syncthreads();
    if (threadIdx.x < N_CLASSES && l_want[threadIdx.x]) {
      const uint32_t c = threadIdx.x, w = l_want[c];
      const int32_t top = atomicSub(&ctl->free_cnt[c], (int32_t)w);
      const uint32_t got = top > 0 ? min((uint32_t)top, w) : 0u;
      if (got < w) atomicAdd(&ctl->free_cnt[c], (int32_t)(w - got));
      l_top[c] = top;
      l_got[c] = got;
    }
    __syncthreads();
    const bool fresh = live && !refused && rank >= l_got[cls];     // no retired block left for it: arena
    unsigned long long my_unit = 0;
    if (fresh) my_unit = atomicAdd(&l_units, (unsigned long long)block_units(tasks[t].old_lg + 1));
    __syncthreads();
    if (threadIdx.x == 0 && l_units) {
      // ONE add per workgroup.  With the host's exact sizing (task_budget == all) the cap cannot bind; in the device-driven
      // round a share that lands beyond the cap is refused and its units are simply lost to the bump pointer (the host maps
      // past them) -- rare by construction (the estimates are 4x the previous batch), and cheaper than a compare-and-swap
      // loop that hundreds of workgroups spin on (measured: 15 -> 85 us for this kernel)
      l_unit0 = atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->arena_next), l_units);
      if (l_unit0 + l_units > arena_cap_units) { l_unit_ok = 0; ctl->spec_failed = 1; if (task_budget == 0xFFFFFFFFu) ctl->arena_oom = 1; }
    }
    __syncthreads();
    if (live) {
      GrowTask& k = tasks[t];
      uint64_t u = 0;
      if (refused) {
      } else if (!fresh) {
        u = fl.list[cls][l_top[cls] - 1 - (int32_t)rank];
      } else if (l_unit_ok) {
        u = l_unit0 + my_unit;
      }
      if (u == 0) ctl->spec_failed = 1;
      k.new_base = (uint32_t)u;
      k.count = 0;
      k.dup = 0;
      k.wrap_from = k.wrap_seen = 0xFFFFFFFFu;
      k.n_disp = 0;
      k.pend_off = k.pend_cap = k.n_pend = 0;
    }
    if (pend_ctl) {                                                         // (uniform)
      uint32_t my_pend = 0;
      const bool takes = live && tasks[t].new_base != 0;
      const uint32_t want = takes ? max(8u, (1u << tasks[t].old_lg) / 2u) : 0u;
      if (takes) my_pend = atomicAdd(&l_pend, want);
      __syncthreads();
      if (threadIdx.x == 0 && l_pend) {
        l_pend0 = atomicAdd(&pend_ctl[1], l_pend);
        l_pend_ok = (uint64_t)l_pend0 + l_pend <= pend_cap_keys;
      }
      __syncthreads();
      if (takes && l_pend_ok) {
        tasks[t].pend_off = l_pend0 + my_pend;
        tasks[t].pend_cap = want;
        task_of[tasks[t].dslot] = t;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_grow_plan(Ctl* ctl, GrowTask* tasks, uint64_t arena_cap_units, FreeLists fl,
                                                   uint32_t task_budget, uint32_t chunk_cap, uint32_t* pend_ctl, uint32_t pend_cap_keys, uint32_t* task_of) {
  grow_plan_body(SMX_VG, ctl, tasks, arena_cap_units, fl, task_budget, chunk_cap, pend_ctl, pend_cap_keys, task_of);
}

// ---- growth takes the waiting keys in (round 6) ------------------------------------------------------------------------------------
// An op whose key is absent from a row at its threshold waits for the row to double (src/smatrix.c:346-348) and then went in
// through the retry: one compare-and-swap per key on the first empty cell of its probe.  On a clustered row the waiting keys of
// one run all probe to the SAME cell, one wins, the others look at the next cell and race again: the k-th key of a front made k
// round trips -- 22 000 ops of a late dense-id batch took 0.3 ms EACH, and the retry's wave-per-op pass (0.6 ms; 1.7 ms per
// young batch) lasted as long as its longest front whatever its grid.  But the kernels that rebuild a row hold its whole new
// table in LDS (k_grow_lds) or its occupancy (k_grow_rest_lds), where a key finds its cell without a trip to memory.  So:
//   k_prep          leaves {directory slot, key} for every op it finds absent;
//   k_grow_plan     gives every row that doubles a bucket of old size / 2 keys -- the new table has no room for more;
//   k_pend_group    after the plan: each record finds its row's task (task_of), enters a hash set (one op per key) and the bucket;
//   k_grow_lds      re-inserts the old cells, then the waiting keys, by the same priority probing (a waiting key's priority is
//                   old size + its place in the bucket: behind every old cell -- the reference's resize, then its inserts);
//   k_grow_rest_lds the workgroup of a row's LAST slice -- its bitmap is the whole new table -- enters them like the cells in
//                   front of a slice (rest_enter) and stores {key, PEND_MARK}; k_grow_finish turns the mark into the value 0.
// As many keys as the new table may take (used <= size/2 before each insert): the rest, and the keys of rows that are moved by
// the global-memory passes, wait for the retry as before.  The retry then finds the keys in place and adds its amounts
// ({key, 0} first, then the op's arithmetic: src/smatrix.c:354-356 then :230/:241/:252); keys that landed beyond the lane's
// budget are in the hint table.  Off while a probe chain may have been cut (ArenaHead::twins: the duplicate check walks old cells).
constexpr uint32_t PEND_MARK = 0xFFFFFFFFu;              // the "old slot" of a key that had none
__global__ __launch_bounds__(256) void k_pend_group(const Ctl* ctl, GrowTask* tasks, const uint32_t* task_of, const uint2* rec, uint32_t rec_cap,
                                                    const uint32_t* pend_ctl, unsigned long long* hash, uint32_t hmask, uint32_t* keys) {
  // (places in a bucket are handed out per WORKGROUP and row: a young table's hottest row has 10^5 keys waiting, and as many
  //  returning atomics on its one counter queued at the memory side -- 1.2 ms for this kernel in the dense stream's third batch)
  constexpr uint32_t LT = 512;                                             // 2 x the workgroup: the rows its 256 records name
  __shared__ uint32_t l_task[LT], l_cnt[LT], l_base[LT];
  const uint32_t n = min(pend_ctl[0], rec_cap), n_tasks = aload(&ctl->n_tasks);
  for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {      // block-uniform
    for (uint32_t q = threadIdx.x; q < LT; q += blockDim.x) { l_task[q] = 0xFFFFFFFFu; l_cnt[q] = 0; }
    __syncthreads();
    const uint32_t i = i0 + threadIdx.x;
    bool mine = false;
    uint32_t t = 0, y = 0, slot = 0, rank = 0, off = 0, cap = 0;
    if (i < n) {
      const uint2 r = rec[i];
      t = task_of[r.x];
      y = r.y;
      if (t < n_tasks) {
        const GrowTask k = tasks[t];
        if (k.dslot == r.x && k.new_base != 0 && k.pend_cap != 0) {        // (else: a stale entry, a refused task, no bucket)
          off = k.pend_off; cap = k.pend_cap;
          const unsigned long long key = ((unsigned long long)r.x << 32) | r.y;      // (y != 0: never the empty entry)
          uint32_t e = fmix32(r.x * 0x9E3779B1u ^ r.y * 0x85EBCA77u) & hmask;
          for (uint32_t guard = 0; guard < 64; guard++) {
            unsigned long long prev = hash[e];
            if (prev == 0ull) prev = atomicCAS(&hash[e], 0ull, key);
            if (prev == 0ull) { mine = true; break; }
            if (prev == key) break;                                        // another op has named this key
            e = (e + 1) & hmask;
          }                                                                // (crowded here: the key waits for the retry)
        }
      }
    }
    if (mine) {
      slot = (t * 0x9E3779B1u) >> 23;                                      // 9 bits
      for (;;) {
        const uint32_t prev = atomicCAS(&l_task[slot], 0xFFFFFFFFu, t);
        if (prev == 0xFFFFFFFFu || prev == t) break;
        slot = (slot + 1) & (LT - 1);
      }
      rank = atomicAdd(&l_cnt[slot], 1u);
    }
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < LT; q += blockDim.x)
      if (l_task[q] != 0xFFFFFFFFu) l_base[q] = atomicAdd(&tasks[l_task[q]].n_pend, l_cnt[q]);
    __syncthreads();
    if (mine) {
      const uint32_t at = l_base[slot] + rank;
      if (at < cap) keys[off + at] = y;
    }
    __syncthreads();                                                        // (the table is reused by the next trip)
  }
}

// chunk -> task maps, filled one wave per CHUNKED task (prep lists them: a steady batch has ~50 of them among 60 000
// tasks, and a wave per task of ALL kinds made this trivial pass 40 us of the growth round's critical path)
// arena != nullptr (clustered rows, k_grow_move_home): GrowTask::wrap_from is worked out as well
// WRAP: the two-pass move of a clustered matrix follows (GrowTask::wrap_from is needed); without, the kernel only fills the maps
template <bool WRAP>
__device__ __forceinline__ void grow_map_body(VGrid g, const Ctl* ctl, GrowTask* tasks, const uint32_t* list,
                                              uint32_t* map_old, uint32_t* map_new, uint8_t* arena) {
  // one WORKGROUP per chunked task (a 2 M-slot row has 10^5 chunk entries: one wave writing them all was 30 us of the
  // growth round's critical path)
  const uint32_t n = aload(&ctl->n_kind[GROW_CHUNKED]);
  for (uint32_t li = g.bid; li < n; li += g.nb) {            // block-uniform
    const uint32_t t = list[li];
    const GrowTask k = tasks[t];
    if (grow_kind(k.old_lg) != GROW_CHUNKED || k.chunk0 == CHUNK_NONE) continue;   // (a range whose task got no block is
    const uint32_t oc = 1u << (k.old_lg - 6), nc = 2u * oc;                          //  still mapped: the passes skip it by new_base)
    for (uint32_t c = threadIdx.x; c < oc; c += blockDim.x) map_old[k.chunk0 + c] = t;
    for (uint32_t c = threadIdx.x; c < nc; c += blockDim.x) map_new[k.chunk0_new + c] = t;
    if (WRAP && arena && k.new_base != 0) {
      // GrowTask::wrap_from: the table's first run, window by window up to its first empty slot
      __shared__ uint32_t l_wrap, l_end;
      if (threadIdx.x == 0) { l_wrap = 0xFFFFFFFFu; l_end = 0xFFFFFFFFu; }
      __syncthreads();
      const uint64_t* O = row_cells(arena, k.old_base);
      const uint32_t old_size = 1u << k.old_lg;
      // (one window first -- a scrambled row's first empty slot is in it --, then eight per turn: a dense row's first run is 10^5
      //  cells, and 400 turns of two barriers were 0.2 ms of the round's critical path)
      auto turn = [&](auto nw, uint32_t b0) -> bool {                           // (block-uniform)
        constexpr uint32_t W = decltype(nw)::value;
        uint64_t c[W];
#pragma unroll
        for (uint32_t q = 0; q < W; q++) { const uint32_t p = b0 + q * blockDim.x + threadIdx.x; c[q] = p < old_size ? O[p] : ~0ull; }
#pragma unroll
        for (uint32_t q = 0; q < W; q++) if (c[q] == 0) atomicMin(&l_end, b0 + q * blockDim.x + threadIdx.x);
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < W; q++) {
          const uint32_t p = b0 + q * blockDim.x + threadIdx.x;
          if (c[q] != 0 && p < old_size && p < l_end && (cell_key(c[q]) & (old_size - 1u)) > p) atomicMin(&l_wrap, cell_key(c[q]) & (old_size - 1u));
        }
        const bool done = l_end != 0xFFFFFFFFu;                                   // (uniform: read between two barriers)
        __syncthreads();
        return done;
      };
      bool done = turn(std::integral_constant<uint32_t, 1>{}, 0u);
      for (uint32_t b0 = blockDim.x; !done && b0 < old_size; b0 += 8u * blockDim.x) done = turn(std::integral_constant<uint32_t, 8>{}, b0);
      __syncthreads();
      if (threadIdx.x == 0) tasks[t].wrap_from = l_wrap;
      __syncthreads();
    }
  }
}
template <bool WRAP>
__global__ __launch_bounds__(256) void k_grow_map(const Ctl* ctl, GrowTask* tasks, const uint32_t* list,
                                                  uint32_t* map_old, uint32_t* map_new, uint8_t* arena) {
  grow_map_body<WRAP>(SMX_VG, ctl, tasks, list, map_old, map_new, arena);
}

// Rows whose old and new table fit in LDS are rebuilt there by one wave or one workgroup (the SCOPE).
// The same priority probing as k_grow_move, but on a table of OLD SLOT INDICES in LDS, where an
// arrival is a single 32-bit atomicMin: the smaller index (earlier old slot) keeps the slot, the
// larger one moves on.  Then the duplicate check of k_grow_finish, the new table written out
// coalesced, and the old block zeroed for reuse -- one read and one write of each block in all.
template <uint32_t THREADS>
struct BlockScope {
  static constexpr uint32_t T = THREADS;
  __device__ static uint32_t tid() { return threadIdx.x; }
  __device__ static void sync() { __syncthreads(); }
};
struct WaveScope {                                   // the lanes of one wave; LDS traffic of a wave is in order
  static constexpr uint32_t T = 64;
  __device__ static uint32_t tid() { return __lane_id(); }
  __device__ static void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
};
//   l_old : 2^old_lg cells, l_tab : 2^(old_lg+1) slot indices, l_cd : {count, dup}, all private to the scope
__device__ inline uint32_t rest_enter(unsigned long long* B, unsigned long long* S, uint32_t nw, uint32_t nmask, bool valid, uint32_t home);   // (below)
// l_bits (round 6): room for one bit per new slot + one per 64-bit word of those, for the waiting keys of big tables
// BITS: this instantiation may meet tables of >= 4096 new cells (k_grow_lds<1024, 13>): with rest_enter inlined the kernels of the
// small kinds went from 18 to 93-96 registers, and the growth round of the SCRAMBLED stream from 2.52 to 2.56 ms per step
// PEND: the matrix is clustered and its waiting keys come with the rebuild; the scrambled stream's launches are the instantiation without
template <typename S, bool BITS = false, bool PEND = false>
__device__ __forceinline__ void grow_lds_task(GrowTask* task, uint8_t* arena, uint64_t* l_old, uint32_t* l_tab,
                                              uint32_t* l_cd, const uint32_t* pend_keys = nullptr, unsigned long long* l_bits = nullptr) {
  constexpr uint32_t NONE = 0xFFFFFFFFu;
  const uint32_t tid = S::tid();
  const ArenaHead* ah = reinterpret_cast<const ArenaHead*>(arena);
  const bool twins = ah->twins != 0, home_on = ah->home_on != 0;      // (uniform)
  if (task->new_base == 0) return;                   // refused by the plan (scope-uniform)
  const uint32_t old_lg = task->old_lg;
  const uint32_t old_size = 1u << old_lg, new_size = 2u << old_lg, nmask = new_size - 1u;
  uint64_t* O = row_cells(arena, task->old_base);
  uint64_t* T = row_cells(arena, task->new_base);
  if (tid == 0) { l_cd[0] = 0; l_cd[1] = 0; }
  for (uint32_t q = tid; q < new_size; q += S::T) l_tab[q] = NONE;
  for (uint32_t p = tid; p < old_size; p += S::T) l_old[p] = O[p];
  S::sync();
  uint32_t mine = 0;
  for (uint32_t p = tid; p < old_size; p += S::T) {
    const uint64_t c = l_old[p];
    if (c == 0) continue;
    mine++;
    uint32_t cur = p, i = cell_key(c) & nmask;
    for (;;) {
      const uint32_t prev = atomicMin(&l_tab[i], cur);
      if (prev == NONE) break;                       // the slot was free
      if (prev > cur) cur = prev;                    // evicted a later cell: carry it onward
      i = (i + 1) & nmask;
    }
  }
  if (mine) atomicAdd(&l_cd[0], mine);
  S::sync();
  // the keys that wait for this doubling (k_pend_group), behind the old cells: priority old_size + place in the bucket
  const uint32_t* pend = nullptr;
  if (PEND && pend_keys && !twins && task->pend_cap) {
    const uint32_t n_old = l_cd[0], cap = new_size / 2u + 1u;
    const uint32_t take = n_old < cap ? min(min(task->n_pend, task->pend_cap), cap - n_old) : 0u;
    S::sync();                                       // (everybody has read the count)
    if (take) {
      pend = pend_keys + task->pend_off;
      const uint32_t nw = new_size >> 6;
      if (BITS && l_bits && nw >= 64 && S::T >= 64) {
        // tables of >= 4096 new cells: on a BITMAP of the slots the old cells took, like the cells in front of a slice
        // (rest_enter).  By priority probing the waiting keys of one run evict each other one cell at a time down the whole
        // run: 0.36 ms for the 8192-cell rows of a late dense-id batch, as long as the chunked passes beside them.
        unsigned long long* Bm = l_bits;
        unsigned long long* Sm = l_bits + nw;
        for (uint32_t q = tid; q < new_size; q += S::T) {
          const uint64_t m = __ballot(l_tab[q] != NONE);
          if ((q & 63u) == 0) Bm[q >> 6] = m;
        }
        S::sync();
        for (uint32_t sw = tid; sw < (nw >> 6); sw += S::T) {
          unsigned long long m = 0;
          for (uint32_t b = 0; b < 64; b++) if (Bm[sw * 64 + b] == ~0ull) m |= 1ull << b;
          Sm[sw] = m;
        }
        S::sync();
        for (uint32_t i0 = tid & ~63u; i0 < take; i0 += S::T) {             // (wave-uniform)
          const uint32_t i = i0 + (tid & 63u);
          const bool valid = i < take;
          const uint32_t z = rest_enter(Bm, Sm, nw, nmask, valid, valid ? pend[i] & nmask : 0u);
          if (valid && z <= nmask) l_tab[z] = old_size + i;
        }
      } else
      for (uint32_t i = tid; i < take; i += S::T) {
        uint32_t cur = old_size + i, q = pend[i] & nmask;
        for (;;) {
          const uint32_t prev = atomicMin(&l_tab[q], cur);
          if (prev == NONE) break;
          if (prev > cur) cur = prev;
          q = (q + 1) & nmask;
        }
      }
      if (tid == 0) l_cd[0] = n_old + take;
    }
    S::sync();
  }
  // a key that a probe from its home finds in ANOTHER slot first is a duplicate (grow_fixdup_one) -- possible only once a
  // probe chain has been cut (ArenaHead::twins)
  if (twins)
  for (uint32_t q = tid; q < new_size; q += S::T) {
    const uint32_t r = l_tab[q];
    if (r == NONE) continue;
    const uint32_t key = cell_key(l_old[r]);
    uint32_t i = key & nmask;
    while (i != q) {
      const uint32_t r2 = l_tab[i];
      if (r2 == NONE || cell_key(l_old[r2]) == key) break;
      i = (i + 1) & nmask;
    }
    if (i != q) l_cd[1] = 1;
  }
  S::sync();
  const uint32_t dup = l_cd[1];
  if (!dup) {
    // (the new table's at-home bitmap, HOME_LG: written whole when the matrix keeps them -- the lanes of a wave hold 64
    //  consecutive slots; otherwise it stays all-zero as the block was handed out)
    const bool bits = home_on && old_lg + 1 >= HOME_LG;
    unsigned long long* hb = row_home(arena, task->new_base, old_lg + 1);
    for (uint32_t q = tid; q < new_size; q += S::T) {
      const uint32_t r = l_tab[q];
      const uint64_t c = r == NONE ? 0ull : PEND && r >= old_size ? pack_cell(pend[r - old_size], 0u) : l_old[r];
      T[q] = c;
      // (a waiting key that landed beyond a lane's budget is remembered: the retry asks the hint table.  Re-hinting the OLD cells
      //  that move far as well was measured -- the old table's hints die with its block -- and bought nothing: 0.31 vs 0.18 ms here)
      if (PEND && r != NONE && r >= old_size && ((q - cell_key(c)) & nmask) > HINT_BUDGET) hint_put(arena, T, cell_key(c), q);
      if (bits) {
        const uint64_t hm = __ballot(c != 0 && cell_key(c) != 0 && (cell_key(c) & nmask) == q);
        if ((q & 63u) == 0) hb[q >> 6] = hm;
      }
    }
    for (uint32_t p = tid; p < old_size; p += S::T) O[p] = 0;
    if (old_lg >= HOME_LG)                                             // the retired block goes back all-zero, bitmap included
      for (uint32_t w = tid; w < (old_size >> 6); w += S::T) row_home(arena, task->old_base, old_lg)[w] = 0;
  }
  if (tid == 0) {
    task->count = l_cd[0];
    task->dup = dup;                                 // the redo reads the (intact) old block
  }
  S::sync();
}

__host__ __device__ inline size_t grow_lds_bytes(uint32_t max_lg) { return ((size_t)16 << max_lg) + ((size_t)2 << max_lg) / 8 + 64; }
// one workgroup (THREADS = 64: one wave) per task of the given kind
template <int THREADS, uint32_t MAX_LG, bool PEND>
__global__ __launch_bounds__(THREADS) void k_grow_lds(const Ctl* ctl, GrowTask* tasks, const uint32_t* list,
                                                      uint32_t kind, uint8_t* arena, const uint32_t* pend_keys) {
  extern __shared__ uint64_t l_dyn[];                               // 2^MAX_LG cells ...
  uint32_t* l_tab = reinterpret_cast<uint32_t*>(l_dyn + (1u << MAX_LG));   // ... and 2^(MAX_LG+1) slot indices
  unsigned long long* l_bits = reinterpret_cast<unsigned long long*>(l_tab + (2u << MAX_LG));   // ... and a bit per new slot (+ summary): grow_lds_bytes
  __shared__ uint32_t l_cd[2];
  const uint32_t n = ctl->n_kind[kind];
  for (uint32_t li = blockIdx.x; li < n; li += gridDim.x)                     // block-uniform
    grow_lds_task<BlockScope<THREADS>, (PEND && MAX_LG >= 12), PEND>(&tasks[list[li]], arena, l_dyn, l_tab, l_cd, pend_keys, l_bits);
}

// one wave per 64 old slots
__device__ __forceinline__ void grow_move_body(VGrid g, const Ctl* ctl, GrowTask* tasks,
                                               const uint32_t* map_old, uint8_t* arena) {
  uint32_t nchunks = aload(&ctl->n_chunks);
  uint32_t wave = (g.bid * blockDim.x + threadIdx.x) >> 6;
  uint32_t lane = threadIdx.x & 63;
  uint32_t nwaves = (g.nb * blockDim.x) >> 6;
  for (uint32_t ch = wave; ch < nchunks; ch += nwaves) {
    uint32_t t = map_old[ch];
    GrowTask& k = tasks[t];
    if (k.new_base == 0) continue;                 // refused by the plan (wave-uniform)
    uint32_t old_size = 1u << k.old_lg;
    uint32_t p = (ch - k.chunk0) * 64 + lane;
    uint64_t cur = 0;
    if (p < old_size) cur = row_cells(arena, k.old_base)[p];
    bool ne = cur != 0;
    uint64_t m = __ballot(ne);
    if (lane == 0 && m) {
      // a giant row is moved by thousands of waves: shard its count over the NEW block's
      // (still unused) sub-counter lines instead of serialising on one word
      if (k.old_lg + 1 >= BIG_LG)
        atomicAdd(&row_subs(arena, k.new_base, k.old_lg + 1)[ch & (SUBS - 1u)].cnt, (uint32_t)__popcll(m));
      else
        atomicAdd(&k.count, (uint32_t)__popcll(m));
    }
    if (ne) {
      uint64_t* T = row_cells(arena, k.new_base);
      uint32_t nmask = (2u << k.old_lg) - 1u;
      uint32_t i = cell_key(cur) & nmask;
      cur = pack_cell(cell_key(cur), p + 1);        // {key, priority}
      uint64_t c = ld_relaxed(&T[i]);
      for (;;) {
        if (c == 0) {
          uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&T[i]), 0ull,
                                    (unsigned long long)cur);
          if (prev == 0) break;
          c = prev;
          continue;
        }
        if (cell_val(c) > cell_val(cur)) {           // resident came later in old order: evict it
          uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&T[i]),
                                    (unsigned long long)c, (unsigned long long)cur);
          if (prev != c) { c = prev; continue; }
          cur = c;                                    // carry the evicted cell onward
        }
        i = (i + 1) & nmask;
        c = ld_relaxed(&T[i]);
      }
    }
  }
}
__global__ __launch_bounds__(256) void k_grow_move(const Ctl* ctl, GrowTask* tasks,
                                                   const uint32_t* map_old, uint8_t* arena) {
  grow_move_body(SMX_VG, ctl, tasks, map_old, arena);
}

// ---- clustered rows (dense ids): the chunked rehash in two passes with a bitmap of the cells that stay AT HOME ---------
// With unscrambled ids a big row is one dense run: keys below the table size sit at home (identity hash), and every key
// that wraps onto the run walks to its end -- 10^4..10^5 cells, one dependent load each, for thousands of cells per
// doubling (k_grow_move took 19.6 ms of a 43 ms step).  Two facts about smatrix_rmap_resize's re-insertion in old slot
// order (src/smatrix.c:392-404) make the walk cheap:
//   (1) a cell never ends further from its new home than it was from its old one (the cells in front of it in old slot
//       order that can reach its new probe sequence at all are the ones that sat between its old home and itself);
//   (2) hence a cell that sat AT HOME in the old table (slot == key mod size) sits at home in the new one -- at slot p or
//       p + size -- whatever the others do, and any cell whose walk comes across it has a LATER old slot (lower priority).
// So pass 1 (k_grow_move_home) stores every at-home cell at its final place with a plain store and leaves, per 64 new
// slots, the mask of the slots it filled: the two mask words of an old chunk are exactly new chunks c and c + size/64,
// written whole by the one wave that owns the old chunk -- no atomics, no initialisation.  Pass 2 (k_grow_move_rest) moves
// the displaced cells with the usual priority probing, but steps over at-home residents 64 at a time by the masks
// without looking at them; k_grow_finish's duplicate check skips them the same way (an at-home resident's key is
// congruent to its own slot, so beyond the first slot of a probe sequence it cannot be the key looked for).
// Taken when a batch has shown long probe sequences (Matrix::clustered); scrambled ids keep the single pass.
// disp (round 6, the time-sliced k_grow_rest_lds): per old chunk the mask of its DISPLACED cells -- non-empty, not stored here --
// and their number per task (GrowTask::n_disp): the slices of a row are cut by these counts, and a slice enters the cells in
// front of it without loading the at-home ones
__device__ __forceinline__ void grow_move_home_body(VGrid g, const Ctl* ctl, GrowTask* tasks, const uint32_t* map_old,
                                                    uint8_t* arena, unsigned long long* disp) {
  const uint32_t nchunks = aload(&ctl->n_chunks);
  const uint32_t wave = (g.bid * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, nwaves = (g.nb * blockDim.x) >> 6;
  for (uint32_t ch = wave; ch < nchunks; ch += nwaves) {
    const uint32_t t = map_old[ch];
    GrowTask& k = tasks[t];
    if (k.new_base == 0) continue;                 // refused by the plan (wave-uniform)
    const uint32_t old_size = 1u << k.old_lg, c = ch - k.chunk0, p = c * 64 + lane;
    const uint64_t cur = row_cells(arena, k.old_base)[p];      // (chunked rows have >= 16384 cells: p < old_size)
    const uint64_t m = __ballot(cur != 0);
    if (lane == 0 && m) {
      if (k.old_lg + 1 >= BIG_LG) atomicAdd(&row_subs(arena, k.new_base, k.old_lg + 1)[ch & (SUBS - 1u)].cnt, (uint32_t)__popcll(m));
      else atomicAdd(&k.count, (uint32_t)__popcll(m));
    }
    const uint32_t key = cell_key(cur), h_old = key & (old_size - 1u);
    if (cur != 0 && h_old > p) atomicMin(&k.wrap_seen, h_old);  // (a wrapped cell: a handful per table at most)
    // (key 0 is never "at home": its (0, v) cell may turn back into an empty one, quirk Q1, and a set bit must stay true)
    const bool home = cur != 0 && key != 0 && h_old == p && p < k.wrap_from;
    const bool hi = home && (key & old_size);                   // new home = p + old_size
    if (home) row_cells(arena, k.new_base)[hi ? p + old_size : p] = pack_cell(key, p + 1);     // {key, priority}, like a moving cell
    const uint64_t lo_m = __ballot(home && !hi), hi_m = __ballot(hi);
    if (disp) {
      const uint64_t dm = m & ~(lo_m | hi_m);
      if (lane == 0) disp[ch] = dm;                  // (counted per row by k_grow_rest_count: one atomic per chunk on a row's one word made this pass 0.5 ms)
    }
    if (lane == 0) {
      // the masks ARE the new table's at-home bitmap (HOME_LG): they stay behind the block for the op kernels' probes
      unsigned long long* hb = row_home(arena, k.new_base, k.old_lg + 1);
      hb[c] = lo_m;
      hb[c + (old_size >> 6)] = hi_m;
    }
  }
}
__global__ __launch_bounds__(256) void k_grow_move_home(const Ctl* ctl, GrowTask* tasks, const uint32_t* map_old, uint8_t* arena,
                                                        unsigned long long* disp) {
  grow_move_home_body(SMX_VG, ctl, tasks, map_old, arena, disp);
}

// the first slot at/after i (cyclically) that no at-home cell holds (`bits`: the row's mask words), as a walk that keeps
// the mask word it is in: successive slots of a walk mostly lie in one word
struct HomeWalk {
  const unsigned long long* bits;
  uint32_t nmask, widx;
  unsigned long long word;
  __device__ inline uint32_t next(uint32_t i) {
    for (uint32_t guard = 0; guard <= (nmask >> 6) + 1u; guard++) {
      if ((i >> 6) != widx) { widx = i >> 6; word = bits[widx]; }
      const unsigned long long free = ~word >> (i & 63u);
      if (free) return i + (uint32_t)__ffsll(free) - 1u;           // (bits beyond the word's end are zero after the shift)
      i = ((i | 63u) + 1u) & nmask;
    }
    return i;
  }
};

// (rows whose displaced cells k_grow_rest_lds places, below: the new table's bitmap fits in LDS and no cell is wrapped)
constexpr uint32_t REST_LDS_MAX_LG = 20;                 // new table: 2^20 bits = 128 KB of LDS
constexpr uint32_t REST_TAKEN = 0x80000000u;             // GrowTask::n_disp: k_grow_rest_plan dealt the row out to k_grow_rest_lds
__device__ inline bool rest_by_lds(const GrowTask& k) {
  return k.old_lg + 1 <= REST_LDS_MAX_LG && k.wrap_seen >= k.wrap_from;      // (wrap_seen < wrap_from: redone serially at the commit)
}
__device__ __forceinline__ void grow_move_rest_body(VGrid g, const Ctl* ctl, GrowTask* tasks, const uint32_t* map_old,
                                                    uint8_t* arena, bool by_lds) {
  const uint32_t nchunks = aload(&ctl->n_chunks);
  const uint32_t wave = (g.bid * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, nwaves = (g.nb * blockDim.x) >> 6;
  for (uint32_t ch = wave; ch < nchunks; ch += nwaves) {
    const uint32_t t = map_old[ch];
    const GrowTask k = tasks[t];
    if (k.new_base == 0) continue;
    const uint32_t old_size = 1u << k.old_lg, p = (ch - k.chunk0) * 64 + lane;
    uint64_t cur = row_cells(arena, k.old_base)[p];
    if (k.wrap_seen < k.wrap_from && lane == 0) tasks[t].dup = 1;             // (see GrowTask::wrap_seen: redone serially at the commit)
    if (by_lds && (k.n_disp & REST_TAKEN)) continue;                              // (k_grow_rest_lds has placed this row's displaced cells)
    if (cur == 0 || (cell_key(cur) != 0 && (cell_key(cur) & (old_size - 1u)) == p && p < k.wrap_from)) continue;           // empty, or placed by the first pass
    uint64_t* T = row_cells(arena, k.new_base);
    const unsigned long long* bits = row_home(arena, k.new_base, k.old_lg + 1);
    const uint32_t nmask = (2u << k.old_lg) - 1u;
    HomeWalk hw{bits, nmask, 0xFFFFFFFFu, 0ull};
    uint32_t i = hw.next(cell_key(cur) & nmask);
    cur = pack_cell(cell_key(cur), p + 1);          // {key, priority}
    // The displaced cells of a dense row pile up behind its run of at-home cells, and a late one walks over all that
    // came before it: the next MOVE_AHEAD slots of the walk are worked out from the masks and loaded TOGETHER.  A value
    // read early is as good as one read in turn: a slot's resident only ever gives way to one of higher priority, so
    // "came before me" stays true, and every claim or eviction is a compare-and-swap against what was read.
    constexpr int MOVE_AHEAD = 8;
    bool placed = false;
    while (!placed) {
      uint32_t at[MOVE_AHEAD];
      uint64_t seen[MOVE_AHEAD];
      at[0] = i;
#pragma unroll
      for (int b = 1; b < MOVE_AHEAD; b++) at[b] = hw.next((at[b - 1] + 1) & nmask);
#pragma unroll
      for (int b = 0; b < MOVE_AHEAD; b++) seen[b] = ld_relaxed(&T[at[b]]);
#pragma unroll
      for (int b = 0; b < MOVE_AHEAD; b++) {
        if (placed) break;
        uint64_t c = seen[b];
        for (;;) {
          if (c == 0) {
            const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&T[at[b]]), 0ull, (unsigned long long)cur);
            if (prev == 0) { placed = true; break; }
            c = prev;
            continue;
          }
          if (cell_val(c) > cell_val(cur)) {             // resident came later in old order: evict it
            const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&T[at[b]]), (unsigned long long)c, (unsigned long long)cur);
            if (prev != c) { c = prev; continue; }
            cur = c;                                      // carry the evicted cell onward
          }
          break;
        }
      }
      i = hw.next((at[MOVE_AHEAD - 1] + 1) & nmask);
    }
  }
}
__global__ __launch_bounds__(256) void k_grow_move_rest(const Ctl* ctl, GrowTask* tasks, const uint32_t* map_old, uint8_t* arena, bool by_lds) {
  grow_move_rest_body(SMX_VG, ctl, tasks, map_old, arena, by_lds);
}

// ---- clustered rows: the displaced cells placed through an occupancy bitmap in LDS (round 5) ---------------------------------
// k_grow_move_rest's priority probing is correct but SLOW on clustered rows: all displaced cells of a run start at once, early
// arrivals of low priority are evicted one by one by the cells that should have come first, and every eviction is a dependent
// compare-and-swap -- chains of thousands (6.4-6.9 ms per dense-id step for ~70 rows, 300 000 displaced cells).  Here ONE
// workgroup takes a row and does what smatrix_rmap_resize does (src/smatrix.c:392-404: re-insert in old slot order, each cell
// into the first free slot from its home) on a BITMAP of the new table kept in LDS -- the at-home masks of the first pass plus
// every cell placed so far -- so "first free slot from home" is a scan of mask words (a summary level steps over runs of full
// words), never a walk over cells:
//   * the old table is cut at EMPTY old slots: a cell never ends further from its new home than it sat from its old one, so
//     the cells between two empty old slots land strictly between them (in the low or the high half) and the pieces are
//     independent; each wave takes a range of pieces, in old slot order;
//   * a wave collects its displaced cells in that order and places them 64 at a time.  Within a step lane l has priority over
//     the lanes above it.  Every pending lane looks up t = its first free slot in the bitmap as it stands; lanes of a run of
//     neighbours with the same t (a pile behind a dense run) take the following free slots in order (z = the r-th free slot
//     from t).  A lane COMMITS -- sets its bit, stores its cell -- when no lower pending lane has the same z (it would lose the
//     slot to it) and no lower lane that does not commit in this round has a smaller z (that lane's place is still open and
//     may turn out to be this very slot); the others look again in the next round.  The lowest pending lane always commits.
//     What a lane commits is exactly its place in the sequential order: everything from its home up to z is taken by then,
//     and nobody before it takes z.
// Rows whose new bitmap does not fit (more than 2^REST_LDS_MAX_LG cells) keep k_grow_move_rest.
constexpr uint32_t REST_THREADS = 512, REST_WAVES = REST_THREADS / 64;     // (8 waves: bitmap + summary + 8 x 2.5 KB of staged cells stay under 160 KB)
constexpr uint32_t REST_STAGE = 320;                     // staged cells per wave (a step takes 64; up to 4 x 64 arrive at once)
constexpr uint32_t REST_BUCKETS = 256;                   // per wave: {slot, lowest lane that wants it}, open addressing
__host__ __device__ inline size_t rest_lds_bytes() {
  return ((size_t)1 << (REST_LDS_MAX_LG - 3)) + ((size_t)1 << (REST_LDS_MAX_LG - 9)) + (size_t)REST_WAVES * REST_STAGE * 8 + (size_t)REST_WAVES * REST_BUCKETS * 4 + (REST_WAVES + 4) * 4;
}
// the first clear bit at/after slot i (cyclically) of the nw-word bitmap B; S: one bit per word of B, set when the word is full
__device__ inline uint32_t lds_first_zero(const unsigned long long* B, const unsigned long long* S, uint32_t nw, uint32_t i) {
  uint32_t w = i >> 6;
  unsigned long long z = ~B[w] & (~0ull << (i & 63u));
  for (uint32_t guard = 0; z == 0 && guard < 2 * nw + 4; guard++) {
    w = (w + 1) & (nw - 1);
    z = ~B[w];
    if (z == 0) {
      // a full word: the summary names the next word that is not (nw >= 64: every summary word is whole)
      const uint32_t ns = nw >> 6;
      uint32_t sw = w >> 6;
      unsigned long long sz = ~S[sw] & (~0ull << (w & 63u));
      for (uint32_t g2 = 0; sz == 0 && g2 <= ns; g2++) { sw = (sw + 1) & (ns - 1); sz = ~S[sw]; }
      if (sz == 0) return 0xFFFFFFFFu;                     // (cannot happen: the table is at most half full)
      w = (sw << 6) + (uint32_t)__ffsll(sz) - 1u;
      z = ~B[w];                                           // (the summary may lag behind a word that has just filled up: the loop goes on)
    }
  }
  return (w << 6) + (uint32_t)__ffsll(z) - 1u;
}
// the r-th (0-based) clear bit at/after slot t (t itself is clear)
__device__ inline uint32_t lds_nth_zero(const unsigned long long* B, uint32_t nw, uint32_t t, uint32_t r) {
  uint32_t w = t >> 6;
  unsigned long long z = ~B[w] & (~0ull << (t & 63u));
  for (uint32_t guard = 0; guard < 2 * nw + 4; guard++) {
    const uint32_t c = (uint32_t)__popcll(z);
    if (r < c) return (w << 6) + select_bit(z, r);
    r -= c;
    w = (w + 1) & (nw - 1);
    z = ~B[w];
  }
  return 0xFFFFFFFFu;
}

static_assert(((size_t)1 << (REST_LDS_MAX_LG - 3)) + ((size_t)1 << (REST_LDS_MAX_LG - 9)) + (size_t)REST_WAVES * REST_STAGE * 8 + (size_t)REST_WAVES * REST_BUCKETS * 4 + (REST_WAVES + 4) * 4 <= 160 * 1024,
              "k_grow_rest_lds: the LDS of one CU");

// ---- time slices (round 6) -------------------------------------------------------------------------------------------------------
// One wave placing a piece's cells 64 per step is a SERIAL path as long as the piece: a dense row's run has no empty old slot in
// 10^5 cells, 25 000 displaced cells in it were 390 steps of one wave (1.55 ms per dense-id batch while 250 CUs stood idle), and
// the cells of its largest cluster -- a third of them, tools/probe/rest_census.c -- depend on each other through every hole of
// the run, so no cut in SPACE exists.  The cut in TIME does: the set of slots taken after any prefix of the re-insertion order
// does not depend on the order within the prefix (linear probing: a cell ends in the first free slot from its home, whoever
// came before), only WHO sits where does.  So a row's displaced cells are cut into K slices of old slot order; the workgroup
// of slice j first enters the cells of slices 0..j-1 into its bitmap in ANY order -- all 512 lanes at once, claims by atomic OR,
// nothing stored -- and then places its own cells in order on exactly the bitmap the sequential re-insertion would have found
// (src/smatrix.c:392-404), storing only those.  K workgroups of a row run on K compute units without talking to each other.
//   k_grow_move_home leaves, per old chunk, the mask of its displaced cells and their count per row (GrowTask::n_disp);
//   k_grow_rest_plan (one workgroup) deals out slices: ceil(n_disp / REST_SLICE_CELLS) per row, at most REST_MAX_SLICES, the
//     rows with the most slices first, and marks the rows it took (bit 31 of n_disp: k_grow_move_rest leaves those alone);
//   k_grow_rest_lds: a workgroup per slice.  The slice's range of old chunks is cut by the displaced counts (cell-balanced).
constexpr uint32_t REST_SLICE_CELLS = 1024;              // displaced cells a slice places in order (16 steps of one wave when they are one piece; swept 256 / 512 /
                                                         // 1024 / 2048 / 4096 on the dense-id stream: 5.20 / 5.08 / 5.01 / 5.03 / 5.26 ms per step)
constexpr uint32_t REST_MAX_SLICES = 64;
// GrowTask::n_disp: a wave per 64 old chunks adds up their masks' bits, one atomic per wave and row
__global__ __launch_bounds__(256) void k_grow_rest_count(const Ctl* ctl, GrowTask* tasks, const uint32_t* map_old, const unsigned long long* disp) {
  const uint32_t nchunks = aload(&ctl->n_chunks);
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t c0 = wave * 64u; c0 < nchunks; c0 += nwaves * 64u) {        // (wave-uniform)
    const uint32_t ch = c0 + lane;
    uint32_t t = 0xFFFFFFFFu, cnt = 0;
    if (ch < nchunks) { t = map_old[ch]; if (tasks[t].new_base != 0) cnt = (uint32_t)__popcll(disp[ch]); else t = 0xFFFFFFFFu; }
    uint64_t todo = __ballot(t != 0xFFFFFFFFu);
    while (todo) {                                                          // (one turn per row among the 64 chunks: mostly one)
      const uint32_t l = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
      const uint32_t tv = (uint32_t)__shfl((int)t, (int)l);
      const bool mine = t == tv;
      uint32_t sum = mine ? cnt : 0u;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d);
      if (lane == l && sum) atomicAdd(&tasks[tv].n_disp, sum);
      todo &= ~__ballot(mine);
    }
  }
}
__global__ __launch_bounds__(1024) void k_grow_rest_plan(const Ctl* ctl, GrowTask* tasks, const uint32_t* list, uint32_t* tab, uint32_t tab_cap,
                                                         uint32_t slice_cells) {
  __shared__ uint32_t l_hist[REST_MAX_SLICES + 1], l_base[REST_MAX_SLICES + 1], l_cnt[REST_MAX_SLICES + 1], l_fits;
  const uint32_t n = aload(&ctl->n_kind[GROW_CHUNKED]);
  if (threadIdx.x <= REST_MAX_SLICES) { l_hist[threadIdx.x] = 0; l_cnt[threadIdx.x] = 0; }
  __syncthreads();
  auto slices_of = [&](uint32_t li) -> uint32_t {
    const GrowTask k = tasks[list[li]];
    if (k.new_base == 0 || grow_kind(k.old_lg) != GROW_CHUNKED || k.chunk0 == CHUNK_NONE || !rest_by_lds(k)) return 0u;
    return min(max(((k.n_disp & ~REST_TAKEN) + slice_cells - 1u) / slice_cells, 1u), REST_MAX_SLICES);
  };
  for (uint32_t li = threadIdx.x; li < n; li += blockDim.x) {
    const uint32_t K = slices_of(li);
    if (K) atomicAdd(&l_hist[K], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t K = REST_MAX_SLICES; K >= 1; K--) { l_base[K] = acc; acc += l_hist[K] * K; }
    l_fits = acc <= tab_cap;                               // (the host sizes the table for every chunk the plan can have accepted: a guard)
    tab[0] = l_fits ? acc : 0u;
  }
  __syncthreads();
  if (!l_fits) return;                                     // nothing is marked: k_grow_move_rest moves every row by priority probing
  for (uint32_t li = threadIdx.x; li < n; li += blockDim.x) {
    const uint32_t K = slices_of(li);
    if (!K) continue;
    const uint32_t t = list[li], at = l_base[K] + atomicAdd(&l_cnt[K], 1u) * K;
    for (uint32_t j = 0; j < K; j++) { tab[1 + 2 * (at + j)] = t; tab[2 + 2 * (at + j)] = j | (K << 16); }
    tasks[t].n_disp |= REST_TAKEN;
  }
}

// `valid` lanes enter a cell with home slot `home` into the bitmap, in any order (the set that results is the one of every order):
// a lane claims the first free slot from its home by atomic OR; the lanes of the wave that lose the SAME slot go on behind it
// together, the r-th of them to the r-th free slot (one slot per lane and turn instead of one winner per turn).  Every slot a
// lane steps over was seen taken or is claimed in this turn by a lane that takes it or finds it taken.  Wave-uniform.
__device__ inline uint32_t rest_enter(unsigned long long* B, unsigned long long* S, uint32_t nw, uint32_t nmask, bool valid, uint32_t home) {
  const uint32_t lane = threadIdx.x & 63u;
  bool pending = valid;
  uint32_t cur = home, r = 0, got = 0xFFFFFFFFu;                            // got: the slot the lane ended in
  while (__any(pending)) {                                                  // (wave-uniform)
    uint32_t z = 0xFFFFFFFFu;
    if (pending) {
      z = lds_first_zero(B, S, nw, cur);
      if (r && z <= nmask) z = lds_nth_zero(B, nw, z, r);
      if (z > nmask) pending = false;                                       // (cannot happen: the table is at most half full)
    }
    if (pending) {
      const unsigned long long bit = 1ull << (z & 63u);
      const unsigned long long before = atomicOr(&B[z >> 6], bit);
      if (!(before & bit)) {
        if ((before | bit) == ~0ull) atomicOr(&S[z >> 12], 1ull << ((z >> 6) & 63u));
        pending = false;
        got = z;
      }
    }
    uint64_t todo = __ballot(pending);
    while (todo) {                                                          // (wave-uniform: one turn per slot that was lost)
      const uint32_t l = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
      const uint32_t zv = (uint32_t)__shfl((int)z, (int)l);
      const uint64_t same = __ballot(pending && z == zv);
      if (pending && z == zv) r = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
      todo &= ~same;
    }
    cur = (z + 1u) & nmask;
  }
  return got;
}

// dbg (measurement builds only, SMX_REST_DBG): counters {steps, rounds, cells, most steps of one wave, trips, most trips of one
// wave}; bit 0 of dbg_mode: the staged cells are dropped instead of placed (what the loads alone cost: tables wrong afterwards)
__global__ __launch_bounds__(REST_THREADS) void k_grow_rest_lds(const Ctl* ctl, GrowTask* tasks, const uint32_t* tab, const unsigned long long* disp,
                                                                uint8_t* arena, unsigned long long* dbg, uint32_t dbg_mode, const uint32_t* pend_keys) {
  extern __shared__ unsigned long long l_rest[];
  unsigned long long* B = l_rest;                                           // 2^(REST_LDS_MAX_LG - 6) words
  unsigned long long* S = B + (1u << (REST_LDS_MAX_LG - 6));                // 2^(REST_LDS_MAX_LG - 12) words
  uint64_t* stage_all = reinterpret_cast<uint64_t*>(S + (1u << (REST_LDS_MAX_LG - 12)));
  uint32_t* scratch_all = reinterpret_cast<uint32_t*>(stage_all + REST_WAVES * REST_STAGE);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint64_t* stage = stage_all + wave * REST_STAGE;                          // {key, old slot + 1} of this wave's pending displaced cells, in old slot order
  uint32_t* bucket = scratch_all + wave * REST_BUCKETS;
  uint32_t* bound = scratch_all + REST_WAVES * REST_BUCKETS;                // where each wave's range of old slots begins
  uint32_t* cut = bound + REST_WAVES + 2;                                   // the slice's first chunk and the one behind its last
  const uint32_t n_slices = tab[0];
  for (uint32_t si = blockIdx.x; si < n_slices; si += gridDim.x) {          // block-uniform
    const uint32_t ti = tab[1 + 2 * si], s_j = tab[2 + 2 * si] & 0xFFFFu, s_k = tab[2 + 2 * si] >> 16;
    const GrowTask k = tasks[ti];
    const long long d_t0 = dbg ? clock64() : 0;
    long long d_place = 0;
    const uint32_t old_size = 1u << k.old_lg, omask = old_size - 1u, new_size = 2u * old_size, nmask = new_size - 1u, nw = new_size >> 6;
    const uint64_t* O = row_cells(arena, k.old_base);
    uint64_t* T = row_cells(arena, k.new_base);
    const unsigned long long* hb = row_home(arena, k.new_base, k.old_lg + 1);
    __syncthreads();                                                        // (the previous task's bitmap is done with)
    for (uint32_t w = threadIdx.x; w < nw; w += REST_THREADS) B[w] = hb[w];
    __syncthreads();
    for (uint32_t sw = threadIdx.x; sw < (nw >> 6); sw += REST_THREADS) {
      unsigned long long m = 0;
      for (uint32_t b = 0; b < 64; b++) if (B[sw * 64 + b] == ~0ull) m |= 1ull << b;
      S[sw] = m;
    }
    __syncthreads();
    const long long d_b = dbg ? clock64() : 0;
    // the slice: old chunks [c_lo, c_hi), cut where the row's running count of displaced cells passes j/K and (j+1)/K of them
    // (every workgroup of the row works the same cuts out of the same masks)
    const uint32_t nck = old_size >> 6;
    const unsigned long long* dm = disp + k.chunk0;
    uint32_t c_lo = 0, c_hi = nck;
    if (s_k > 1) {
      const uint32_t n_disp = k.n_disp & ~REST_TAKEN;
      const uint32_t t_lo = (uint32_t)((uint64_t)s_j * n_disp / s_k), t_hi = (uint32_t)((uint64_t)(s_j + 1u) * n_disp / s_k);
      const uint32_t per = (nck + REST_THREADS - 1u) / REST_THREADS;
      const uint32_t b0 = min(threadIdx.x * per, nck), b1 = min(b0 + per, nck);
      uint32_t mine = 0;
      for (uint32_t c = b0; c < b1; c++) mine += (uint32_t)__popcll(dm[c]);
      uint32_t incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if (lane >= (uint32_t)d) incl += o; }
      uint32_t* wsum = reinterpret_cast<uint32_t*>(stage_all);              // (the staging area is idle until the cells come)
      if (lane == 63) wsum[wave] = incl;
      if (threadIdx.x == 0) { cut[0] = 0; cut[1] = nck; }
      __syncthreads();
      uint32_t front = incl - mine;                                         // displaced cells in front of chunk b0
      for (uint32_t w = 0; w < wave; w++) front += wsum[w];
      // the cut for a count t > 0: behind the first chunk that brings the running count to t
      for (int e = 0; e < 2; e++) {
        const uint32_t t = e ? t_hi : t_lo;
        if ((e && s_j + 1u == s_k) || t == 0 || !(t > front && t <= front + mine)) continue;
        uint32_t cum = front;
        for (uint32_t c = b0; c < b1; c++) { cum += (uint32_t)__popcll(dm[c]); if (cum >= t) { cut[e] = c + 1u; break; } }
      }
      __syncthreads();
      c_lo = cut[0]; c_hi = cut[1];
      __syncthreads();
    }
    const uint32_t s_lo = c_lo << 6, s_hi = c_hi << 6;
    const long long d_u0 = dbg ? clock64() : 0;
    // the cells in front of the slice enter the bitmap in any order: 64 chunk masks per wave and trip, their cells' slots laid
    // out in the staging area (up to 8 per chunk and turn), the keys of up to 512 cells loaded together
    if (c_lo) {
      uint32_t* st32 = reinterpret_cast<uint32_t*>(stage);                  // 2 * REST_STAGE old slot indices
      const uint32_t* keys = reinterpret_cast<const uint32_t*>(O);
      for (uint32_t c0 = wave * 64u; c0 < c_lo; c0 += REST_WAVES * 64u) {   // (wave-uniform)
        const uint32_t c = c0 + lane;
        unsigned long long mk = c < c_lo ? dm[c] : 0ull;
        while (__any(mk != 0)) {                                            // (wave-uniform)
          const uint32_t cnt = min(8u, (uint32_t)__popcll(mk));
          uint32_t incl = cnt;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if (lane >= (uint32_t)d) incl += o; }
          const uint32_t tot = (uint32_t)__shfl((int)incl, 63);
          uint32_t at = incl - cnt;
          for (uint32_t i = 0; i < cnt; i++) { st32[at++] = (c << 6) + (uint32_t)__ffsll(mk) - 1u; mk &= mk - 1ull; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          uint32_t key[8];
#pragma unroll
          for (uint32_t q = 0; q < 8; q++) key[q] = q * 64u + lane < tot ? keys[2u * st32[q * 64u + lane]] : 0u;
#pragma unroll
          for (uint32_t q = 0; q < 8; q++)
            if (q * 64u < tot) rest_enter(B, S, nw, nmask, q * 64u + lane < tot, key[q] & nmask);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
      __syncthreads();
    }
    const long long d_u1 = dbg ? clock64() : 0;
    // this wave's range of the slice's old slots: from the first empty old slot at/after its nominal start to the one of the
    // next wave (the slice itself begins wherever its count does: what lies in front of it is in the bitmap)
    {
      uint32_t b = s_lo + ((wave * (c_hi - c_lo)) / REST_WAVES) * 64u;
      if (wave != 0) {
        for (bool found = false; !found;) {                                 // (wave-uniform; eight 64-cell windows in flight)
          uint64_t c[8];
#pragma unroll
          for (int q = 0; q < 8; q++) { const uint32_t p = b + (uint32_t)q * 64u + lane; c[q] = p < s_hi ? O[p] : 1ull; }
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const uint64_t m = __ballot(c[q] == 0);
            if (m && !found) { b += (uint32_t)q * 64u + (uint32_t)__ffsll((unsigned long long)m) - 1u; found = true; }
          }
          if (!found) { b += 512; if (b >= s_hi) { b = s_hi; found = true; } }
        }
      }
      if (lane == 0) bound[wave] = b;
      if (threadIdx.x == 0) { bound[REST_WAVES] = s_hi; bound[REST_WAVES + 1] = 0; }
    }
    __syncthreads();
    const uint32_t lo = bound[wave], hi = bound[wave + 1];
    const long long d_t1 = dbg ? clock64() : 0;
    // A table whose first run continues its last one round the end (wrapped cells: GrowTask::wrap_from): the wrapped cells sit
    // in the FIRST piece and come first in old slot order, but land among the cells of the LAST piece -- so the wave that holds
    // the end of the table starts only when wave 0 is through (bound[REST_WAVES + 1]); all other pieces stay independent.
    // (Time slices: a slice that begins inside the first piece places the wrapped cells it holds with its wave 0; what lies in
    //  front of a slice is in its bitmap already.  The cells the wrapped ones can meet are those of old slots >= wrap_from --
    //  the last run, which the first pass did not take for cells at home -- wherever the slice's cuts fall in it: a slice of
    //  a 32768-cell row held the first piece AND the first 300 cells of the last run, and ended before the table did.)
    if (k.wrap_seen != 0xFFFFFFFFu && wave != 0 && hi > k.wrap_from && lo < hi)
      while (__hip_atomic_load(&bound[REST_WAVES + 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(8);
    uint32_t n_st = 0;                                                      // staged cells (wave-uniform)
    uint32_t d_steps = 0, d_rounds = 0, d_trips = 0;
    // a step: the first `cnt` staged cells (cnt <= 64), lane l = the l-th of them in old slot order
    auto place = [&](uint32_t cnt) {
      const long long d_p0 = dbg ? clock64() : 0;
      const bool valid = lane < cnt;
      const uint64_t cell = valid ? stage[lane] : 0ull;                     // {key, priority}
      uint32_t cur = cell_key(cell) & nmask;
      bool pending = valid && !(dbg_mode & 1u);
      d_steps++;
      while (__any(pending)) {                                              // (wave-uniform)
        d_rounds++;
        uint32_t t = 0xFFFFFFFFu, z = 0xFFFFFFFFu;
        if (pending) t = lds_first_zero(B, S, nw, cur);
        // RUNS of pending neighbours that fill one stretch of free slots: the r-th lane of a run takes the r-th free slot from the
        // run's base.  A run begins where the first free slot changes; two runs are one when the second one's first free slot is
        // among the slots the first run is going to take (its base <= t <= the slot of the lane before): lanes in old slot order
        // mostly have rising homes, and a pile behind a run of taken slots grows exactly like that.  (Valid as LOWER bounds
        // whatever the homes are: by the time such a lane's turn comes, the lanes of its run below it have taken -- or found
        // taken -- every free slot from the base up to its own.)
        // (neighbours = pending lanes bound for the same HALF of the new table: a step's cells alternate between the two -- new
        //  home = old home or old home + old size -- and the halves do not meet except at their ends)
        const uint64_t hi_half = __ballot(pending && t >= old_size);
        const uint64_t same = t >= old_size ? hi_half : ~hi_half;
        const uint64_t pm = __ballot(pending) & same;
        const uint64_t lower = pm & ((1ull << lane) - 1ull);
        const uint32_t prev = lower ? 63u - (uint32_t)__clzll((unsigned long long)lower) : lane;   // the pending lane before this one
        const uint32_t t_prev = (uint32_t)__shfl((int)t, (int)prev);
        uint64_t starts = __ballot(pending && (lower == 0 || t != t_prev));                      // lanes that begin a run
        for (;;) {                                                                               // (wave-uniform)
          const uint64_t sb = starts & same & ((2ull << lane) - 1ull);
          const uint32_t start_lane = sb ? 63u - (uint32_t)__clzll((unsigned long long)sb) : 0u;
          const uint32_t r = (uint32_t)__popcll(lower & ~((1ull << start_lane) - 1ull));          // pending lanes of the run below this one
          const uint32_t t_run = (uint32_t)__shfl((int)t, (int)start_lane);
          z = 0xFFFFFFFFu;
          if (pending) z = r ? lds_nth_zero(B, nw, t_run, r) : t_run;
          const uint32_t z_prev = (uint32_t)__shfl((int)z, (int)prev), t_run_prev = (uint32_t)__shfl((int)t_run, (int)prev);
          const uint64_t mm = __ballot(pending && lower != 0 && ((starts >> lane) & 1ull) && t >= t_run_prev && t <= z_prev);
          if (!mm) break;
          starts &= ~mm;
        }
        // RELAXATION to a fixed point.  Invariant of every pending lane: each free slot from its starting point up to (not
        // including) its z is taken, by the time its turn comes, by a lane below it.  A lane that shares its z with a lower lane
        // gives way: that slot is taken too by then, so its z moves on to the next free one -- the invariant holds again.  When
        // no two pending lanes share a slot, every lane's z IS its place in the sequential order (induction over the lanes: all
        // that is free before z is gone, and nobody below ends at z), and all of them commit at once.
        // Who shares: an open-addressed table of {slot, lowest lane that wants it}; only lanes that give way insert again (the
        // entry of the slot they leave keeps naming the lower lane), everybody looks at its own entry again.
#pragma unroll
        for (uint32_t q = 0; q < REST_BUCKETS; q += 64) bucket[q + lane] = 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t bk = 0, n_keys = (uint32_t)__popcll(__ballot(pending));
        bool insert = pending;
        uint64_t losers = 0;
        for (;;) {                                                          // (wave-uniform)
          if (insert) {
            const uint32_t mine = (z << 6) | lane;
            bk = ((z * 0x9E3779B1u) >> 16) & (REST_BUCKETS - 1u);
            for (;;) {
              const uint32_t old = atomicCAS(&bucket[bk], 0xFFFFFFFFu, mine);
              if (old == 0xFFFFFFFFu) break;
              if ((old >> 6) == z) { atomicMin(&bucket[bk], mine); break; }
              bk = (bk + 1u) & (REST_BUCKETS - 1u);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const bool loser = pending && (bucket[bk] & 63u) != lane;
          losers = __ballot(loser);
          if (dbg && (dbg_mode & 2u) && lane == 0) { atomicAdd(&dbg[8], (unsigned long long)__popcll(losers)); atomicAdd(&dbg[9], 1ull); }
          if (!losers) break;
          n_keys += (uint32_t)__popcll(losers);
          if (n_keys > REST_BUCKETS * 3u / 4u) break;                       // (the table is filling up: what is settled commits, the rest starts over)
          insert = loser;
          if (loser) z = lds_first_zero(B, S, nw, (z + 1u) & nmask);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // everything commits -- or, when the table ran full, the lanes below the lowest one that still shares a slot
        const uint32_t upto = losers ? (uint32_t)__ffsll((unsigned long long)losers) - 1u : 64u;
        if (pending && lane < upto) {
          const unsigned long long bit = 1ull << (z & 63u);
          const unsigned long long before = atomicOr(&B[z >> 6], bit);
          if ((before | bit) == ~0ull) atomicOr(&S[z >> 12], 1ull << ((z >> 6) & 63u));
          T[z] = cell;
          pending = false;
        }
        if (pending) cur = t;                                               // (everything below t is taken: the next look starts there)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      // what is left moves to the front
      const uint32_t rest = n_st - cnt;
      uint64_t mv[(REST_STAGE + 63) / 64];
#pragma unroll
      for (uint32_t q = 0; q < (REST_STAGE + 63) / 64; q++) mv[q] = q * 64 + lane < rest ? stage[cnt + q * 64 + lane] : 0ull;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (uint32_t q = 0; q < (REST_STAGE + 63) / 64; q++) if (q * 64 + lane < rest) stage[q * 64 + lane] = mv[q];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      n_st = rest;
      if (dbg) d_place += clock64() - d_p0;
    };
    // the range, eight chunks of 64 old slots per trip (their loads in flight together), staged four at a time
    for (uint32_t p0 = lo; p0 < hi; p0 += 512) {                             // (wave-uniform)
      d_trips++;
      uint64_t c[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t p = p0 + (uint32_t)q * 64u + lane;
        c[q] = p < hi ? O[p] : 0ull;
      }
#pragma unroll
      for (int half = 0; half < 2; half++) {
#pragma unroll
        for (int q = half * 4; q < half * 4 + 4; q++) {
          const uint32_t p = p0 + (uint32_t)q * 64u + lane;
          const uint32_t key = cell_key(c[q]);
          const bool displaced = c[q] != 0 && !(key != 0 && (key & omask) == p && p < k.wrap_from);      // (at-home cells were stored by the first pass)
          const uint64_t dm = __ballot(displaced);
          if (displaced) stage[n_st + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull))] = pack_cell(key, p + 1u);
          n_st += (uint32_t)__popcll(dm);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        while (n_st >= 64) place(64);
      }
    }
    if (n_st) place(n_st);
    if (wave == 0 && lane == 0) __hip_atomic_store(&bound[REST_WAVES + 1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    // the row's LAST slice: the bitmap now is the whole new table -- the keys that wait for this doubling go in (see k_pend_group)
    if (pend_keys && s_j + 1u == s_k && k.pend_cap && !reinterpret_cast<const ArenaHead*>(arena)->twins) {      // (block-uniform)
      __syncthreads();                                                      // (every wave has placed its cells)
      if (wave == 0) {
        // non-empty old cells: the first pass's count (a big row's sits in the new block's sub-counter lines)
        uint32_t n_old = lane == 0 ? aload(&tasks[ti].count) : 0u;
        if (k.old_lg + 1 >= BIG_LG) n_old += __hip_atomic_load(&row_subs(arena, k.new_base, k.old_lg + 1)[lane].cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (SUBS == 64)
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) n_old += (uint32_t)__shfl_xor((int)n_old, dd);
        const uint32_t cap = new_size / 2u + 1u;
        if (lane == 0) cut[0] = n_old < cap ? min(min(aload(&tasks[ti].n_pend), k.pend_cap), cap - n_old) : 0u;
      }
      __syncthreads();
      const uint32_t take = cut[0];
      const uint32_t* pend = pend_keys + k.pend_off;
      // (a key that lands at home gets its bit in the at-home bitmap from k_grow_finish, not here: the workgroups of the row's
      //  other slices may still be loading that bitmap -- they run in no particular order once a launch has more slices than
      //  workgroups -- and must find the cells the FIRST pass stored, nothing else)
      for (uint32_t i0 = wave * 64u; i0 < take; i0 += REST_THREADS) {        // (wave-uniform)
        const bool valid = i0 + lane < take;
        const uint32_t key = valid ? pend[i0 + lane] : 0u;
        const uint32_t z = rest_enter(B, S, nw, nmask, valid, key & nmask);
        if (valid && z <= nmask) {
          T[z] = pack_cell(key, PEND_MARK);
          if (((z - key) & nmask) > HINT_BUDGET) hint_put(arena, T, key, z);
        }
      }
      if (threadIdx.x == 0 && take) atomicAdd(&tasks[ti].count, take);
    }
    if (dbg && lane == 0) {
      atomicAdd(&dbg[0], (unsigned long long)d_steps); atomicAdd(&dbg[1], (unsigned long long)d_rounds);
      atomicMax(&dbg[3], (unsigned long long)d_steps); atomicAdd(&dbg[4], (unsigned long long)d_trips); atomicMax(&dbg[5], (unsigned long long)d_trips);
      atomicMax(&dbg[6], (unsigned long long)d_rounds);
      const long long d_t2 = clock64();                                     // (clock ticks: the longest row, its set-up, its longest wave, the placing in it; sums for the averages)
      atomicMax(&dbg[50], (unsigned long long)(d_t2 - d_t0)); atomicMax(&dbg[51], (unsigned long long)(d_t1 - d_t0));
      atomicMax(&dbg[52], (unsigned long long)(d_t2 - d_t1)); atomicMax(&dbg[53], (unsigned long long)d_place);
      atomicAdd(&dbg[54], (unsigned long long)(d_t2 - d_t1)); atomicAdd(&dbg[55], (unsigned long long)d_place); atomicAdd(&dbg[56], 1ull);
      atomicMax(&dbg[57], (unsigned long long)(d_u1 - d_u0)); atomicAdd(&dbg[58], (unsigned long long)(d_u1 - d_u0));   // (the cells in front of the slice)
      atomicMax(&dbg[59], (unsigned long long)(d_b - d_t0)); atomicAdd(&dbg[60], (unsigned long long)(d_b - d_t0));     // (the bitmap)
      atomicMax(&dbg[61], (unsigned long long)(d_u0 - d_b)); atomicAdd(&dbg[62], (unsigned long long)(d_u0 - d_b));     // (the cuts)
      atomicMax(&dbg[63], (unsigned long long)(d_t1 - d_u1)); atomicAdd(&dbg[49], (unsigned long long)(d_t1 - d_u1));   // (the waves' ranges)
      if (wave == 0) atomicMax(&dbg[64 + ((dbg_mode >> 8) & 31u)], ((unsigned long long)((d_t2 - d_t0) >> 10) << 40) | ((unsigned long long)((d_t1 - d_t0) >> 10) << 20) | (unsigned long long)k.old_lg);
    }
  }
}

// ---- the far join: the absent keys of a row placed TOGETHER (round 6) ------------------------------------------------------------
// A claimed insert (far_claim_insert) costs three atomics on words its row's other claimers want too -- the word's rank counter,
// the live occupancy word, the ticket -- and every same-address atomic queues at the memory side: the 2 000 new keys that wrap
// onto one hot row's run in a dense-id batch (10^4 in a young table's) went through their front's words one after the other,
// and the pass in front of prep lasted as long as its hottest front, whatever the grid (0.65 ms per late batch, 1.7 ms per
// young one).  The keys the join calls absent are known before that pass starts -- the entries of F without a slot -- and WHICH
// free cell an absent key gets is the library's to choose as long as none between its home and its cell stays empty
// (src/smatrix.c:343-380: some order of the reference's inserts).  So, between the scan and the pass:
//   k_far_absent   one pass over F: every absent key takes a place in its row's bucket (one fetch-add on the row's count, kept at
//                  the row's first unit; 32 entries per unit = cells / 16 per row), rows met for the first time enter a list;
//   k_far_place    a workgroup per listed row: the row's occupancy words (as scanned) into LDS, tickets for all its keys at once
//                  (sub_tickets_bulk / the row's `used`), the keys entered into the bitmap like the cells in front of a slice
//                  (rest_enter: LDS atomics, ties broken by rank), {key, 0} stored with a compare-and-swap -- a cell the words
//                  call free may hold the row's (0, v) entry -- and the slot written into the key's entry of F.
// The pass then finds these keys like any far key that sits in its cell, and adds its amounts there.  What gets no ticket (the
// row stands at its threshold) or no bucket entry stays absent and takes the old way (claim, deferral, prep).  Rows whose
// bitmap does not fit in LDS (more than 2^REST_LDS_MAX_LG cells) too.
constexpr uint32_t FAR_BUCKET_PER_UNIT = 32;
// (two launches: rows of up to 2^FAR_PLACE_SMALL_LG cells a WAVE per row with 8 KB of LDS -- a thousand such rows per batch, and
//  a workgroup that reserves the largest row's 130 KB keeps its compute unit to itself: 170 us --, the larger ones a workgroup of
//  512 per row)
constexpr uint32_t FAR_PLACE_SMALL_LG = 16;
__host__ __device__ inline size_t far_place_lds_bytes(uint32_t max_lg) { return ((size_t)1 << (max_lg - 3)) + ((size_t)1 << (max_lg - 9)) + 64; }
// rcnt[u]: absent keys of the row whose first unit is u (zeroed by k_far_scan); rcnt[cap_units]: rows in `rows`
__global__ __launch_bounds__(256) void k_far_absent(const DirSlot* dir, const uint32_t* unit_row, uint8_t* arena, const uint4* tab, uint32_t tmask, uint32_t* rcnt,
                                                    uint32_t cap_units, uint32_t* bucket, uint32_t* rows, uint32_t rows_cap) {
  const ArenaHead* ah = reinterpret_cast<const ArenaHead*>(arena);
  if (ah->twins || ah->far_overflow) return;                               // (uniform: the pass does not use the join then / takes no claimed inserts)
  // (a list of the entries k_far_keys creates instead of this pass over the table was measured: this kernel 64 -> 28 us, k_far_keys
  //  52 -> 98 -- one returning atomic per wave on the list's one counter)
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e <= tmask; e += gridDim.x * blockDim.x) {
    const uint4 v = tab[e];
    if (v.x == 0 || v.z != FAR_NOT_FOUND) continue;                         // (unused, a row's own entry, or a key the scan has found)
    const uint4* row = far_entry(const_cast<uint4*>(tab), tmask, v.y, 0u);
    if (!row) continue;
    const uint32_t fu = row->z;
    const uint32_t at = atomicAdd(&rcnt[fu], 1u);
    if (at == 0) { const uint32_t r = atomicAdd(&rcnt[cap_units], 1u); if (r < rows_cap) rows[r] = fu; }
    // (the bucket: FAR_BUCKET_PER_UNIT entries per unit of the row, from the row's first unit on; what does not fit stays absent)
    const uint32_t units = 1u << (meta_lg(dir[unit_row[fu]].meta) - FAR_UNIT_LG);
    if (at < units * FAR_BUCKET_PER_UNIT) bucket[(size_t)fu * FAR_BUCKET_PER_UNIT + at] = e;
  }
}
template <uint32_t FAR_PLACE_THREADS, uint32_t MIN_LG, uint32_t MAX_LG>
__global__ __launch_bounds__(FAR_PLACE_THREADS) void k_far_place(DirSlot* dir, const uint32_t* unit_row, uint8_t* arena, uint4* tab, const uint32_t* rcnt,
                                                                 uint32_t cap_units, const uint32_t* bucket, const uint32_t* rows, uint32_t rows_cap,
                                                                 unsigned long long* occ, uint32_t* zeros, const unsigned long long* occ0) {
  extern __shared__ unsigned long long l_place[];
  unsigned long long* B = l_place;
  unsigned long long* S = B + (1u << (MAX_LG - 6));
  uint32_t* l_got = reinterpret_cast<uint32_t*>(S + (MAX_LG >= 12 ? 1u << (MAX_LG - 12) : 1u));
  const ArenaHead* ah = reinterpret_cast<const ArenaHead*>(arena);
  if (ah->twins || ah->far_overflow) return;
  const uint32_t n_rows = min(rcnt[cap_units], rows_cap);
  for (uint32_t ri = blockIdx.x; ri < n_rows; ri += gridDim.x) {            // block-uniform
    const uint32_t fu = rows[ri], dslot = unit_row[fu];
    const DirSlot d = dir[dslot];
    const uint32_t lg = meta_lg(d.meta);
    if (lg > MAX_LG || lg < MIN_LG) continue;                              // (block-uniform: another launch's rows, or none's)
    const uint32_t mask = (1u << lg) - 1u, nw = 1u << (lg - 6), units = 1u << (lg - FAR_UNIT_LG);
    const uint32_t n = min(rcnt[fu], units * FAR_BUCKET_PER_UNIT);         // (what the bucket holds)
    uint64_t* cells = row_cells(arena, d.base);
    __syncthreads();                                                        // (the previous row's bitmap is done with)
    {                                                                      // (four words per lane in flight: a 2^20-cell row's 16 384 words one load after the other were 30 us)
      const unsigned long long* src = occ0 + (size_t)fu * FAR_UNIT_WORDS;
      for (uint32_t w0 = threadIdx.x; w0 < nw; w0 += 4u * FAR_PLACE_THREADS) {
        unsigned long long q[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) q[j] = w0 + j * FAR_PLACE_THREADS < nw ? src[w0 + j * FAR_PLACE_THREADS] : 0ull;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) if (w0 + j * FAR_PLACE_THREADS < nw) B[w0 + j * FAR_PLACE_THREADS] = q[j];
      }
    }
    __syncthreads();
    for (uint32_t sw = threadIdx.x; sw < (nw >> 6); sw += FAR_PLACE_THREADS) {
      unsigned long long m = 0;
      for (uint32_t b = 0; b < 64; b++) if (B[sw * 64 + b] == ~0ull) m |= 1ull << b;
      S[sw] = m;
    }
    if (threadIdx.x < 64) {
      // src/smatrix.c:346: a key goes in only while used <= size/2 -- tickets for as many of the row's keys as there is room for.
      // A big row's 64 shares are looked at by the 64 lanes of the first wave together (nobody else touches the row in this
      // launch; sub_tickets_bulk's walk from share to share was 45 us per big row): lane a takes from share a what the lanes
      // below it leave wanted.
      uint32_t got;
      if (lg >= BIG_LG) {
        SubCtr* sc = row_subs(arena, d.base, lg) + threadIdx.x;             // (SUBS == 64)
        const uint32_t cnt = sc->cnt, quota = sc->quota, room = quota > cnt ? quota - cnt : 0u;
        uint32_t incl = room;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, dd); if ((int)threadIdx.x >= dd) incl += o; }
        const uint32_t before = incl - room, take = before >= n ? 0u : min(room, n - before);
        if (take) sc->cnt = cnt + take;
        got = min((uint32_t)__shfl((int)incl, 63), n);
      } else {
        const uint32_t cap = (mask + 1u) / 2u + 1u;
        got = d.used < cap ? min(n, cap - d.used) : 0u;
        if (got && threadIdx.x == 0) dir[dslot].used = d.used + got;
      }
      if (threadIdx.x == 0) {
        if (got && !(d.meta & META_DIRTY)) dir[dslot].meta = d.meta | META_DIRTY;
        *l_got = got;
      }
    }
    __syncthreads();
    const uint32_t got = *l_got;
    for (uint32_t i0 = (threadIdx.x & ~63u); i0 < got; i0 += FAR_PLACE_THREADS) {     // (wave-uniform)
      const uint32_t i = i0 + (threadIdx.x & 63u);
      bool todo = i < got;
      uint32_t e = 0, Y = 0, from = 0;
      if (todo) { e = bucket[(size_t)fu * FAR_BUCKET_PER_UNIT + i]; Y = tab[e].x; from = Y & mask; }
      while (__any(todo)) {                                                 // (wave-uniform)
        const uint32_t z = rest_enter(B, S, nw, mask, todo, from);
        if (todo) {
          if (z > mask) { todo = false; continue; }                         // (cannot happen: the row holds at most size/2 + 1 keys)
          const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&cells[z]), 0ull, (unsigned long long)pack_cell(Y, 0u));
          if (prev == 0) {
            tab[e].z = z;
            atomicOr(&occ[(size_t)fu * FAR_UNIT_WORDS + (z >> 6)], 1ull << (z & 63u));                  // (the live words: what the claimers and the probes read)
            atomicSub(&zeros[fu + (z >> FAR_UNIT_LG)], 1u);
            todo = false;
          } else from = (z + 1u) & mask;                                    // (the row's (0, v) entry sits there: taken, on)
        }
      }
    }
  }
}

// one wave per 64 new slots: replace the carried old-slot index by the value
// home_bits != nullptr: the two-pass move ran (clustered rows); the duplicate check steps over at-home residents
__device__ __forceinline__ void grow_finish_body(VGrid g, const Ctl* ctl, GrowTask* tasks,
                                                 const uint32_t* map_new, uint8_t* arena, bool two_pass = false) {
  uint32_t nchunks = 2u * aload(&ctl->n_chunks);    // (chunked rows: the new table has twice the old one's chunks)
  const bool twins = reinterpret_cast<const ArenaHead*>(arena)->twins != 0;     // (uniform) no chain was ever cut: no key sits twice
  uint32_t wave = (g.bid * blockDim.x + threadIdx.x) >> 6;
  uint32_t lane = threadIdx.x & 63;
  uint32_t nwaves = (g.nb * blockDim.x) >> 6;
  for (uint32_t ch = wave; ch < nchunks; ch += nwaves) {
    uint32_t t = map_new[ch];
    const GrowTask k = tasks[t];
    if (k.new_base == 0) continue;
    uint32_t new_size = 2u << k.old_lg;
    uint32_t q = (ch - k.chunk0_new) * 64 + lane;
    if (q < new_size) {
      uint64_t* T = row_cells(arena, k.new_base);
      uint64_t c = T[q];
      // a key that waited for this doubling: {key, 0}, and its bit in the at-home bitmap when it sits at home (one word per wave)
      const bool waited = c != 0 && cell_val(c) == PEND_MARK;
      const uint64_t wh = __ballot(waited && cell_key(c) != 0 && (cell_key(c) & (new_size - 1u)) == q);
      if (wh && lane == 0 && k.old_lg + 1 >= HOME_LG) atomicOr(&row_home(arena, k.new_base, k.old_lg + 1)[q >> 6], (unsigned long long)wh);
      if (c != 0) {
        if (waited) { T[q] = pack_cell(cell_key(c), 0u); continue; }
        uint64_t o = row_cells(arena, k.old_base)[cell_val(c) - 1];
        T[q] = pack_cell(cell_key(c), cell_val(o));
        if (!twins) continue;
        // a key that a probe from its home finds in ANOTHER slot first is a duplicate
        // (keys are stable during this kernel, only value words change)
        uint32_t nmask = new_size - 1u, i = cell_key(c) & nmask;
        if (two_pass) {
          const unsigned long long* bits = row_home(arena, k.new_base, k.old_lg + 1);
          // (q itself is not at home unless q == i: the walk stops there at the latest)
          if (i != q && cell_key(T[i]) != cell_key(c)) {
            // (eight slots of the walk at a time, like k_grow_move_rest: keys do not change in this kernel)
            HomeWalk hw{bits, nmask, 0xFFFFFFFFu, 0ull};
            i = hw.next((i + 1) & nmask);
            for (bool done = false; !done;) {
              uint32_t at[8];
              uint32_t kk[8];
              at[0] = i;
#pragma unroll
              for (int b = 1; b < 8; b++) at[b] = at[b - 1] == q ? q : hw.next((at[b - 1] + 1) & nmask);
#pragma unroll
              for (int b = 0; b < 8; b++) kk[b] = cell_key(T[at[b]]);
#pragma unroll
              for (int b = 0; b < 8; b++)
                if (!done && (at[b] == q || kk[b] == cell_key(c))) { done = true; i = at[b]; }
              if (!done) i = hw.next((at[7] + 1) & nmask);
            }
          }
        } else {
          while (i != q && cell_key(T[i]) != cell_key(c)) i = (i + 1) & nmask;
        }
        if (i != q) tasks[t].dup = 1;
      }
    }
  }
}
__global__ __launch_bounds__(256) void k_grow_finish(const Ctl* ctl, GrowTask* tasks,
                                                     const uint32_t* map_new, uint8_t* arena, bool two_pass) {
  grow_finish_body(SMX_VG, ctl, tasks, map_new, arena, two_pass);
}

// A row table can hold one key twice: y=0 writes may turn the uncounted (0,v) cell back
// into an empty one (quirk Q1/Q3) and so cut a probe chain, after which the key behind the
// cut is inserted again (the same happens after a reload that dropped a value-0 key, Q4).
// smatrix_rmap_resize merges such twins -- the second one finds the first through
// rmap_insert, keeps its slot and overwrites its value (src/smatrix.c:353-357,:401-402).
// Priority probing cannot express the merge, so these (rare) rows are redone here the
// reference's way: one lane, old slot order.  The old block is left zeroed, like the other paths
// leave it (k_grow_lds / k_grow_zero skip rows marked dup).
__device__ inline void grow_fixdup_one(GrowTask& k, uint8_t* arena) {
  const uint32_t old_size = 1u << k.old_lg, nmask = 2u * old_size - 1u;
  uint64_t* O = row_cells(arena, k.old_base);
  uint64_t* T = row_cells(arena, k.new_base);
  for (uint32_t q = 0; q <= nmask; q++) T[q] = 0;
  uint32_t used = 0;
  for (uint32_t p = 0; p < old_size; p++) {
    const uint64_t c = O[p];
    if (c == 0) continue;
    const uint32_t key = cell_key(c);
    uint32_t i = key & nmask;
    while (cell_key(T[i]) != key && T[i] != 0) i = (i + 1) & nmask;   // :363-380
    if (cell_key(T[i]) == 0 || cell_key(T[i]) != key) used++;          // :353-354
    T[i] = c;
  }
  k.count = used;
  for (uint32_t p = 0; p < old_size; p++) O[p] = 0;
  // the at-home bitmaps (HOME_LG): the new one is rebuilt for the table as it now stands, the retired block's is wiped
  if (k.old_lg + 1 >= HOME_LG) {
    unsigned long long* hb = row_home(arena, k.new_base, k.old_lg + 1);
    for (uint32_t w = 0; w <= (nmask >> 6); w++) {
      unsigned long long m = 0;
      for (uint32_t b = 0; b < 64; b++) {
        const uint64_t c = T[w * 64 + b];
        if (c != 0 && cell_key(c) != 0 && (cell_key(c) & nmask) == w * 64 + b) m |= 1ull << b;
      }
      hb[w] = m;
    }
  }
  if (k.old_lg >= HOME_LG)
    for (uint32_t w = 0; w < (old_size >> 6); w++) row_home(arena, k.old_base, k.old_lg)[w] = 0;
}

// one wave per 64 old slots: a retired block goes back to its size class's stack ZEROED
// (row creation and growth rely on fresh blocks being all-empty)
__device__ __forceinline__ void grow_zero_body(VGrid g, const Ctl* ctl, const GrowTask* tasks,
                                               const uint32_t* map_old, uint8_t* arena) {
  uint32_t nchunks = aload(&ctl->n_chunks);
  uint32_t wave = (g.bid * blockDim.x + threadIdx.x) >> 6;
  uint32_t lane = threadIdx.x & 63;
  uint32_t nwaves = (g.nb * blockDim.x) >> 6;
  for (uint32_t ch = wave; ch < nchunks; ch += nwaves) {
    const GrowTask k = tasks[map_old[ch]];
    if (k.dup || k.new_base == 0) continue;          // grow_fixdup_one still needs (and then zeroes) it; refused: untouched
    const uint32_t p = (ch - k.chunk0) * 64 + lane;
    if (p < (1u << k.old_lg)) row_cells(arena, k.old_base)[p] = 0;
    if (lane == 0) row_home(arena, k.old_base, k.old_lg)[ch - k.chunk0] = 0;       // (chunked rows have >= 2^14 cells: HOME_LG)
  }
}
__global__ __launch_bounds__(256) void k_grow_zero(const Ctl* ctl, const GrowTask* tasks,
                                                   const uint32_t* map_old, uint8_t* arena) {
  grow_zero_body(SMX_VG, ctl, tasks, map_old, arena);
}

// publish the new tables (src/smatrix.c:408-410) and push the old blocks on their classes' stacks
// (one atomic per class and workgroup; the host sized every stack for this round's pushes beforehand)
__device__ __forceinline__ void grow_commit_body(VGrid g, Ctl* ctl, GrowTask* tasks, DirSlot* dir, uint8_t* arena,
                                                 FreeLists fl, uint32_t* big_list = nullptr, uint32_t big_cap = 0) {
  __shared__ uint32_t l_want[N_CLASSES], l_at[N_CLASSES];
  const uint32_t n = aload(&ctl->n_tasks);
  for (uint32_t t0 = g.bid * blockDim.x; t0 < n; t0 += g.nb * blockDim.x) {    // block-uniform
    if (threadIdx.x < N_CLASSES) l_want[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t t = t0 + threadIdx.x;
    bool live = t < n;
    GrowTask k = {};
    uint32_t cls = 0, rank = 0;
    if (live && tasks[t].new_base == 0) {               // refused by the plan: the row stays as it is
      atomicAnd(&dir[tasks[t].dslot].meta, ~META_GROW);
      live = false;
    }
    if (live) {
      if (tasks[t].dup) grow_fixdup_one(tasks[t], arena);
      k = tasks[t];
      DirSlot& d = dir[k.dslot];
      const uint32_t lg = k.old_lg + 1;
      uint32_t count = k.count;
      if (lg >= BIG_LG && !k.dup) {                     // k_grow_move's sharded count (the redo recounts itself)
        const SubCtr* sc = row_subs(arena, k.new_base, lg);
        for (uint32_t i = 0; i < SUBS; i++) count += sc[i].cnt;
      }
      d.meta = META_USED | META_DIRTY | (lg << META_LG_SHIFT);
      d.base = k.new_base;
      d.used = count;
      if (big_list && lg == FAR_ROW_LG) {             // the row joins the rows the far join scans (k_far_rows: big_list)
        const uint32_t at = atomicAdd(&big_list[0], 1u);
        if (at < big_cap) big_list[1u + at] = k.dslot;
      }
      if (lg >= BIG_LG) {
        const uint32_t cap = (1u << lg) / 2u + 1u;
        subs_init(row_subs(arena, k.new_base, lg), cap > count ? cap - count : 0u);
      }
      if (k.old_lg >= BIG_LG) {                         // the old block's sub-counter lines, zeroed too
        uint64_t* z = reinterpret_cast<uint64_t*>(row_subs(arena, k.old_base, k.old_lg));
        for (uint32_t i = 0; i < SUBS * 8; i++) z[i] = 0;
      }
      cls = k.old_lg - ROW_FIRST_LG;
      rank = atomicAdd(&l_want[cls], 1u);
    }
    __syncthreads();
    if (threadIdx.x < N_CLASSES && l_want[threadIdx.x])
      l_at[threadIdx.x] = (uint32_t)atomicAdd(&ctl->free_cnt[threadIdx.x], (int32_t)l_want[threadIdx.x]);
    __syncthreads();
    if (live) {
      fl.list[cls][l_at[cls] + rank] = k.old_base;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_grow_commit(Ctl* ctl, GrowTask* tasks, DirSlot* dir, uint8_t* arena,
                                                     FreeLists fl, uint32_t* big_list, uint32_t big_cap) {
  grow_commit_body(SMX_VG, ctl, tasks, dir, arena, fl, big_list, big_cap);
}

// The at-home bitmaps of all rows of >= 2^HOME_LG cells, rebuilt from the tables as they stand: run once when a matrix turns
// out clustered (until then nobody sets bits) and after a file has been loaded into a clustered matrix.
// k_home_list: the directory slots of such rows; k_home_rebuild: blockIdx.y = entry of that list, a wave per 64 cells.
__global__ __launch_bounds__(256) void k_home_list(const DirSlot* dir, uint32_t dir_size, uint32_t* list, uint32_t* n_list, uint32_t cap) {
  for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < dir_size; h += gridDim.x * blockDim.x) {
    const DirSlot d = dir[h];
    if ((d.meta & META_USED) && d.base != 0 && meta_lg(d.meta) >= HOME_LG) {
      const uint32_t at = atomicAdd(n_list, 1u);
      if (at < cap) list[at] = h;
    }
  }
}
__global__ __launch_bounds__(256) void k_home_rebuild(const DirSlot* dir, const uint32_t* list, uint32_t first, uint8_t* arena) {
  const DirSlot d = dir[list[first + blockIdx.y]];
  const uint32_t lg = meta_lg(d.meta), nwords = 1u << (lg - 6), mask = (1u << lg) - 1u;
  const uint64_t* cells = row_cells(arena, d.base);
  unsigned long long* hb = row_home(arena, d.base, lg);
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6); w < nwords; w += gridDim.x * 4u) {
    const uint32_t p = w * 64u + lane;
    const uint64_t c = cells[p];
    const uint64_t m = __ballot(c != 0 && cell_key(c) != 0 && (cell_key(c) & mask) == p);
    if (lane == 0) hb[w] = m;
  }
}

// big rows flagged by prep: fold the sub-counters into `used`, share out what room is left
__device__ __forceinline__ void rebal_body(VGrid g, const Ctl* ctl, const uint32_t* rebal, DirSlot* dir, uint8_t* arena) {
  uint32_t n = aload(&ctl->n_rebal);
  for (uint32_t t = g.bid * blockDim.x + threadIdx.x; t < n; t += g.nb * blockDim.x) {
    DirSlot& d = dir[rebal[t]];
    const uint32_t lg = meta_lg(d.meta);
    SubCtr* sc = row_subs(arena, d.base, lg);
    uint32_t used = d.used;
    for (uint32_t k = 0; k < SUBS; k++) used += sc[k].cnt;
    const uint32_t cap = (1u << lg) / 2u + 1u;
    d.used = used;
    d.meta &= ~META_REBAL;
    subs_init(sc, cap > used ? cap - used : 0u);
  }
}
__global__ void k_rebal(const Ctl* ctl, const uint32_t* rebal, DirSlot* dir, uint8_t* arena) {
  rebal_body(SMX_VG, ctl, rebal, dir, arena);
}

// Between two op rounds that the HOST does not separate (speculative chain): what round 0 deferred becomes the length
// of the list the next round reads, round 0's counters are kept for the host's statistics, and the per-round part of the
// control block starts from zero again (what ctl_reset_round does from the host).  One lane.
__global__ void k_round_advance(Ctl* ctl, const uint32_t* rebal, DirSlot* dir, uint8_t* arena) {
  rebal_body(VGrid{0, 1}, ctl, rebal, dir, arena);     // (the handful of big rows whose quotas want re-partitioning: no launch of their own)
  __syncthreads();
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  ctl->n_prev = ctl->n_defer;
  ctl->spec_nd0 = ctl->n_defer;
  ctl->spec_nt0 = ctl->n_tasks;
  ctl->spec_gu0 = ctl->grow_units;
  ctl->spec_nrebal0 = ctl->n_rebal;
  ctl->spec_dirfull0 = ctl->dir_full;
  for (int k = 0; k < 4; k++) ctl->spec_nkind0[k] = ctl->n_kind[k];
  const uint32_t keep_long = ctl->n_long, keep_oom = ctl->arena_oom, keep_long_ops = ctl->n_long_ops, n_absent = ctl->n_absent;
  uint64_t* z = reinterpret_cast<uint64_t*>(ctl);
  for (uint32_t i = 0; i < CTL_ROUND_BYTES / 8; i++) z[i] = 0;
  ctl->n_defer = n_absent;                           // (the list the next round WRITES begins with the ops the folding kernel put there)
  ctl->n_long = keep_long;                           // (sticky for the batch: the host switches the retries to lane-per-op)
  ctl->n_long_ops = keep_long_ops;                   // (summed over the rounds of a chain)
  ctl->arena_oom = keep_oom;
}

// ---- directory growth -----------------------------------------------------------
__global__ __launch_bounds__(256) void k_dir_rehash(const DirSlot* old, uint32_t old_size,
                                                    DirSlot* dir, uint32_t dmask) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= old_size) return;
  DirSlot s = old[p];
  if (!(s.meta & META_USED)) return;
  uint32_t h = fmix32(s.x) & dmask;
  uint64_t want = (uint64_t)s.meta | ((uint64_t)s.x << 32);
  for (;;) {
    uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&dir[h]), 0ull,
                              (unsigned long long)want);
    if (prev == 0) break;
    h = (h + 1) & dmask;
  }
  dir[h].base = s.base;
  dir[h].used = s.used;
}
