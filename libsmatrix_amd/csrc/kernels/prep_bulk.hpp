// kernels/prep_bulk.hpp -- prep (row creation, growth decisions) and the bulk path (deferred ops grouped by row, one wave per row).
// A fragment of smx_kernels.hpp (round 5: the 4 500-line header split by concern, no kernel changed): included there, in order,
// INSIDE namespace smx; not a header of its own.

// ---- prep kernel --------------------------------------------------------------
//
// Runs over the ops the op kernel deferred, on a quiescent table:
//  * creates missing rows (src/smatrix.c:641-662: 16 zeroed cells, used 0),
//    refusing (op stays deferred) when the directory stands at its load limit;
//  * flags a row for growth iff the op's key is ABSENT and the row stands at the
//    reference's threshold -- the exact condition under which the reference's
//    next insert would call smatrix_rmap_resize (src/smatrix.c:346-348).
// One leader per distinct 32-bit key among the lanes of this wave that `want`: calls f(key) on the
// leader lane only.  Deferred ops cluster on few rows (a row at its threshold defers every new key),
// and every expensive step of prep -- the creation protocol, the sub-counter sum, the flag atomics --
// is per ROW, not per op: without the election a million lanes hammered the same directory word
// (measured: 5.6 ms of a 9.5 ms step).
template <typename F>
__device__ inline void per_distinct(bool want, uint32_t key, F f) {
  // election first (ALU + ballots only), then ALL leaders run f together so that their memory
  // round trips overlap -- running f inside the loop would serialise a wave with 64 distinct rows
  uint64_t todo = __ballot(want);
  const uint32_t lane = __lane_id();
  bool leader_here = false;
  while (todo) {
    const uint32_t leader = __ffsll((unsigned long long)todo) - 1;
    const uint32_t k0 = __shfl(key, leader);
    todo &= ~__ballot(want && key == k0);
    leader_here |= lane == leader;
  }
  if (leader_here) f(key);
}

constexpr uint32_t PREP_THREADS = 1024;

__device__ __forceinline__ void prep_body(
    VGrid g, Ctl* ctl, DirSlot* dir, uint32_t dmask, uint32_t dir_limit, uint8_t* arena,
    uint64_t arena_cap_units, const uint32_t* defer, const uint32_t* __restrict__ xs,
    const uint32_t* __restrict__ ys, GrowTask* tasks, uint32_t* klist, uint32_t kcap, uint32_t* rebal,
    FreeLists fl, uint32_t st, uint32_t create_only, uint32_t wpo_max, uint2* pend_rec = nullptr, uint32_t pend_rec_cap = 0, uint32_t* pend_ctl = nullptr) {
  // pend_rec (round 6, clustered matrices): every op whose key is ABSENT leaves {directory slot, key} there (pend_ctl[0] counts
  //             them): the rows that double in this round take these keys in as they are rebuilt (k_pend_group, growth.hpp)
  // wpo_max (clustered matrices): a list of at most so many ops is taken a WAVE per op -- lane 0 holds the op, the wave finishes
  //             its long probe -- like k_apply_wpo: a few hundred deferred ops of big clustered rows, 64 to a wave, walked their
  //             10^4..10^5 cells one lane after the other (7-16 ms for 300-700 ops of the dense-id stream's late rounds)
  // create_only bit 0: rows are created, nothing is flagged for growth (the bulk path decides growth itself, k_fix_rows)
  //             bit 1: the listed ops' keys are known to be ABSENT (k_insert_keys has just looked: a key that exists is never
  //                    deferred, and nobody inserts another list entry's key) -- step C's probe is skipped
  // block-scope scratch of the row-creation step
  __shared__ uint32_t l_set[2 * PREP_THREADS];     // row ids this block is creating (hash set, dedupe)
  __shared__ uint32_t l_cnt[4];                    // [0] lanes at an empty slot, [1] winners, [2] r0, [3] added
  __shared__ unsigned long long l_u0;
  __shared__ uint32_t l_k[8], l_kb[8];             // growth tasks filed by this block: total, by kind; list bases
  __shared__ unsigned long long l_units;
  const uint32_t n = aload(&ctl->n_defer);
  const uint64_t stride = (uint64_t)g.nb * blockDim.x;
  const bool wpo = n <= wpo_max;
  const uint64_t n_lanes = wpo ? (uint64_t)n * 64u : (uint64_t)n;
  for (uint64_t t064 = (uint64_t)g.bid * blockDim.x; t064 < n_lanes; t064 += stride) {       // block-uniform trip count (64-bit: no wrap near 2^32)
    const uint64_t tl = t064 + threadIdx.x;
    const uint32_t t = wpo ? (uint32_t)(tl >> 6) : (uint32_t)tl;
    const bool live = tl < n_lanes && (!wpo || (tl & 63u) == 0);
    uint32_t X = 0, Y = 0;
    if (live) {
      const uint32_t j = defer ? defer[t] : t;         // (no list: the ops are the n_defer entries of xs / ys themselves -- packed keys)
      X = xs[(size_t)j * st];
      Y = ys[(size_t)j * st];
    }
    // A. where does X live?  (read-only probe)
    uint32_t h = fmix32(X) & dmask;
    bool missing = false;
    uint64_t mx = 0;
    if (live) {
      // PLAIN loads: a million deferred ops may all ask for the one hottest row, and L1-bypassing
      // loads of a single word queue up at one L2 channel (5 ms measured).  A stale line can only
      // show an empty slot where a row has just been created; the creation step below
      // re-reads atomically, so that is harmless.  Keys of claimed slots never change.
      for (;;) {
        mx = *reinterpret_cast<const uint64_t*>(&dir[h]);           // {meta, x}
        if (mx == 0) { missing = true; break; }
        if ((uint32_t)(mx >> 32) == X) break;
        h = (h + 1) & dmask;
      }
    }
    // B. create missing rows, once per row id and BLOCK (src/smatrix.c:641-662).  The directory
    //    counter and the arena bump pointer are single words: both are reserved once per block for
    //    all of its new rows (per-op they queued 2x10^5 returning atomics on two addresses: 5 ms;
    //    per wave still 3.7 ms on a batch that creates 10^5 rows).
    if (__syncthreads_or(missing)) {
      for (uint32_t i = threadIdx.x; i < 2 * PREP_THREADS; i += PREP_THREADS) l_set[i] = 0xFFFFFFFFu;
      if (threadIdx.x < 4) l_cnt[threadIdx.x] = 0;
      __syncthreads();
      // B0. one lane per distinct row id (the id 0xFFFFFFFF cannot use the set: it always tries)
      bool mine = missing;
      if (missing && X != 0xFFFFFFFFu) {
        uint32_t q = (X * 0x9E3779B1u) >> 21;            // 11 bits
        for (;;) {
          uint32_t prev = atomicCAS(&l_set[q], 0xFFFFFFFFu, X);
          if (prev == 0xFFFFFFFFu) break;               // first of its id in this block
          if (prev == X) { mine = false; break; }
          q = (q + 1) & (2 * PREP_THREADS - 1);
        }
      }
      // B1. walk (atomically) to the first slot that is empty or already holds X
      bool at_empty = false;
      uint32_t hh = h, rank = 0;
      if (mine) {
        for (;;) {
          uint64_t cur = ld_relaxed(reinterpret_cast<uint64_t*>(&dir[hh]));
          if (cur == 0) { at_empty = true; break; }
          if ((uint32_t)(cur >> 32) == X) break;         // another block created it meanwhile
          hh = (hh + 1) & dmask;
        }
        if (at_empty) rank = atomicAdd(&l_cnt[0], 1u);
      }
      __syncthreads();
      // B2. one directory reservation for the block
      if (threadIdx.x == 0 && l_cnt[0] &&
          __hip_atomic_load(&ctl->dir_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < dir_limit) {
        l_cnt[2] = atomicAdd(&ctl->dir_used, l_cnt[0]);  // may still land beyond the limit: given back below
        l_cnt[3] = 1;
      }
      __syncthreads();
      // B3. claim {meta,x} in one CAS.  A slot lost to ANOTHER row id is not a reason to wait for the next
      //     round (K new ids with one first-empty slot would need K rounds -- ids with equal fmix32(x) & mask are
      //     easy to craft): the lane walks on to the next slot that is empty or holds X, like the reference's
      //     insert does under its lock (src/smatrix.c:677-693).  Load <= 1/2, so an empty slot always exists.
      bool won = false;
      uint32_t rank2 = 0;
      if (at_empty) {
        if (l_cnt[3] && (uint64_t)l_cnt[2] + rank < dir_limit) {
          const uint64_t want = (uint64_t)(META_USED | META_DIRTY | (ROW_FIRST_LG << META_LG_SHIFT)) | ((uint64_t)X << 32);
          for (;;) {
            const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&dir[hh]), 0ull, (unsigned long long)want);
            if (prev == 0) { won = true; break; }
            if ((uint32_t)(prev >> 32) == X) break;         // another workgroup created this very row meanwhile
            uint64_t cur;                                  // (ONE load per slot: it may be claimed between two looks)
            do {
              hh = (hh + 1) & dmask;
              cur = ld_relaxed(reinterpret_cast<uint64_t*>(&dir[hh]));
            } while (cur != 0 && (uint32_t)(cur >> 32) != X);
            if (cur != 0) break;                           // it holds X
          }
          if (won) rank2 = atomicAdd(&l_cnt[1], 1u);
        } else {
          ctl->dir_full = 1;                             // directory at its limit
        }
      }
      __syncthreads();
      // B4. give back what was reserved but not used; the winners' 16-cell blocks come from the
      //     stack of retired (zeroed) class-0 blocks first, the rest from ONE arena reservation
      if (threadIdx.x == 0) {
        const uint32_t n_res = l_cnt[3] ? l_cnt[0] : 0u, n_won = l_cnt[1];
        if (n_res > n_won) atomicSub(&ctl->dir_used, n_res - n_won);
        uint32_t got = 0;
        int32_t top = 0;
        if (n_won) {
          top = atomicSub(&ctl->free_cnt[0], (int32_t)n_won);            // old height
          got = top > 0 ? min((uint32_t)top, n_won) : 0u;
          if (got < n_won) atomicAdd(&ctl->free_cnt[0], (int32_t)(n_won - got));
          if (got < n_won)
            l_u0 = atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->arena_next), (unsigned long long)(n_won - got));
        }
        l_cnt[2] = got;
        l_cnt[3] = (uint32_t)top;
      }
      __syncthreads();
      if (won) {
        const uint32_t got = l_cnt[2];
        uint64_t u;
        if (rank2 < got) u = fl.list[0][l_cnt[3] - 1u - rank2];
        else u = l_u0 + (rank2 - got);
        if (u >= arena_cap_units) ctl->arena_oom = 1;                        // host guarantees this never fires
        else __hip_atomic_store(&dir[hh].base, (uint32_t)u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();                                   // l_cnt / l_set are reused by the next trip
    }
    // C. the row exists (base==0: created a moment ago in this very launch -> empty, nothing to flag):
    //    is this op's key absent?
    bool absent = false;
    uint32_t base = 0, lg = 0;
    LongProbe lp{false, nullptr, 0, 0};
    if (create_only & 1u) continue;                 // (block-uniform)
    if ((create_only & 2u) && live && !missing && Y != 0) {
      base = dir[h].base;
      lg = meta_lg((uint32_t)mx);
      absent = base != 0;
    } else if (live && !missing && Y != 0) {
      base = dir[h].base;          // plain: 0 only for a row created in this very launch
      if (base != 0) {
        lg = meta_lg((uint32_t)mx);
        const uint32_t mask = (1u << lg) - 1u;
        const uint64_t* cells = row_cells(arena, base);
        uint32_t pos = Y & mask;
        absent = true;                              // also when the table has no empty cell left
        for (uint32_t step = 0; step <= mask; step++) {
          uint64_t c = cells[pos];
          if (cell_key(c) == Y) { absent = false; break; }
          if (c == 0) break;
          pos = (pos + 1) & mask;
          if (step >= PROBE_BUDGET) { lp = LongProbe{true, cells, mask, pos}; break; }
        }
      }
    }
    // (the far join of this batch, while it is valid: the key's cell is known, or the rest of the probe goes by the occupancy words)
    const unsigned long long* occ = nullptr;
    {
      const ArenaHead* ah = reinterpret_cast<const ArenaHead*>(arena);
      if (lp.need && ah->far_on && !ah->twins) {
        const FarHit fh = far_find(arena, lp.cells, Y);
        if (fh.state == FAR_FOUND && cell_key(lp.cells[fh.slot]) == Y) { absent = false; lp.need = false; }
        // (the words AS SCANNED, not the live ones: a claimed insert marks its cell in the live words, and a probe by those steps
        //  over it -- the second op of a key that another op of the pass claimed and inserted found the key "absent" here, and a
        //  row that ended the batch at exactly size/2 + 1 keys was doubled for a key it held: tests/cold_soak.py, seed 12)
        else if (fh.state == FAR_ABSENT) occ = ah->far_occ0 + (fh.occ - ah->far_occ);
      }
    }
    while (__any(lp.need)) {                        // long sequences (dense ids): the wave finishes them (coop_probe)
      const uint32_t p = coop_probe(lp.need, lp.cells, lp.mask, Y, lp.pos, reinterpret_cast<const ArenaHead*>(arena)->home_on != 0, occ);
      if (lp.need) {
        lp.need = false;
        absent = p == PROBE_NONE || cell_key(lp.cells[p]) != Y;     // the table is quiescent here: the answer is final
      }
    }
    if (pend_rec) {                                 // (uniform) one reservation per wave
      const bool rec = absent && base != 0 && lg >= ROW_FIRST_LG;
      const uint64_t rm = __ballot(rec);
      if (rm) {
        uint32_t at = 0;
        if (__lane_id() == (uint32_t)__ffsll((unsigned long long)rm) - 1u) at = atomicAdd(&pend_ctl[0], (uint32_t)__popcll(rm));
        at = (uint32_t)__shfl((int)at, __ffsll((unsigned long long)rm) - 1);
        if (rec) {
          const uint32_t i = at + (uint32_t)__popcll(rm & ((1ull << __lane_id()) - 1ull));
          if (i < pend_rec_cap) pend_rec[i] = uint2{h, Y};
        }
      }
    }
    // D. once per row with an absent key: grow it iff it stands at the reference's threshold
    //    (src/smatrix.c:346-348); a big row with room left only has its quotas re-partitioned
    //    "Once per row" is decided in two steps: a wave-level election (ballots), then the wave
    //    leaders meet in a block-level LDS set.  All ~2400 waves of a launch are resident at once and
    //    most of them hold an op of the same few hot rows; with the wave election alone every one of
    //    them sent the flag atomic (and, for big rows, 64 sub-counter loads) to the same address.
    bool lead = false;
    per_distinct(absent, h, [&](uint32_t) { lead = true; });
    if (__syncthreads_or(lead)) {
      for (uint32_t i = threadIdx.x; i < 2 * PREP_THREADS; i += PREP_THREADS) l_set[i] = 0xFFFFFFFFu;
      if (threadIdx.x < 8) l_k[threadIdx.x] = 0;
      if (threadIdx.x == 0) l_units = 0;
      __syncthreads();
      if (lead) {
        uint32_t q = (h * 0x9E3779B1u) >> 21;              // 11 bits
        for (;;) {
          const uint32_t prev = atomicCAS(&l_set[q], 0xFFFFFFFFu, h);
          if (prev == 0xFFFFFFFFu) break;                 // first of its row in this block
          if (prev == h) { lead = false; break; }
          q = (q + 1) & (2 * PREP_THREADS - 1);
        }
      }
      bool mk = false;                               // this lane files a growth task
      uint32_t t_lg = 0, t_base = 0, kind = 0, rk = 0, rkk = 0;
      if (lead) {
        const uint32_t meta = __hip_atomic_load(&dir[h].meta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(meta & (META_GROW | META_REBAL))) {
          t_lg = meta_lg(meta);
          t_base = __hip_atomic_load(&dir[h].base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          uint32_t used = __hip_atomic_load(&dir[h].used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (t_lg >= BIG_LG) used += subs_sum(row_subs(arena, t_base, t_lg));
          if (used > (1u << t_lg) / 2u) {
            const uint32_t old = atomicOr(&dir[h].meta, META_GROW);
            mk = !(old & META_GROW);
          } else if (t_lg >= BIG_LG) {
            // room is left, but this op's sub-counter had used up its share: re-partition
            const uint32_t old = atomicOr(&dir[h].meta, META_REBAL);
            if (!(old & META_REBAL)) rebal[atomicAdd(&ctl->n_rebal, 1u)] = h;
          }
        }
      }
      // the task list and the per-kind work lists are reserved ONCE PER BLOCK, all five counters in
      // one wave instruction: every atomic instruction on these few words of one line queues at the
      // memory side (per task: 0.7 ms per batch; per wave: still 0.13 ms)
      if (mk) {
        kind = grow_kind(t_lg);
        rk = atomicAdd(&l_k[0], 1u);
        rkk = atomicAdd(&l_k[1 + kind], 1u);
        atomicAdd(&l_units, (unsigned long long)block_units(t_lg + 1));
      }
      __syncthreads();
      if (threadIdx.x < 5 && l_k[threadIdx.x])
        l_kb[threadIdx.x] = atomicAdd(threadIdx.x == 0 ? &ctl->n_tasks : &ctl->n_kind[threadIdx.x - 1], l_k[threadIdx.x]);
      if (threadIdx.x == 0 && l_units)
        atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->grow_units), l_units);
      __syncthreads();
      if (mk) {
        const uint32_t k = l_kb[0] + rk;
        klist[kind * kcap + l_kb[1 + kind] + rkk] = k;       // (kind 3, the chunked rows: k_grow_map walks that list)
        tasks[k].dslot = h;
        tasks[k].old_lg = t_lg;
        tasks[k].old_base = t_base;
      }
      __syncthreads();                                     // the LDS scratch is reused by the next trip
    }
  }
}

__global__ __launch_bounds__(PREP_THREADS) void k_prep(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint32_t dir_limit, uint8_t* arena,
    uint64_t arena_cap_units, const uint32_t* defer, const uint32_t* __restrict__ xs,
    const uint32_t* __restrict__ ys, GrowTask* tasks, uint32_t* klist, uint32_t kcap, uint32_t* rebal,
    FreeLists fl, uint32_t st, uint32_t create_only, uint32_t wpo_max, uint2* pend_rec, uint32_t pend_rec_cap, uint32_t* pend_ctl) {
  prep_body(SMX_VG, ctl, dir, dmask, dir_limit, arena, arena_cap_units, defer, xs, ys, tasks, klist, kcap, rebal, fl, st, create_only, wpo_max, pend_rec, pend_rec_cap, pend_ctl);
}

// ---- the bulk path: many deferred ops (bulk loads, the first batches of a matrix) ---------------------------
// A batch that CREATES its rows defers every op in round 0, and a new 115-key row then needs one round per doubling
// (create, 16 -> 32 -> ... -> 256: six rounds, each re-running the op kernel over everything still pending -- the
// config-3 build ran at 1 G ops/s against 12 G in steady state).  Here the deferred ops are grouped by row instead
// (count per directory slot, exclusive scan, scatter) and ONE WAVE per row then does what the reference does for that
// row's ops in list order -- smatrix_rmap_insert with its `used > size/2` test, smatrix_rmap_resize re-inserting in
// old slot order (src/smatrix.c:343-416) -- on a table held in LDS, start to finish, and writes the final table out
// once.  The sequential core is the reference's algorithm itself (one lane; the row's ops are staged and its results
// written back by all 64), so sizes, `used` and the layout are those of a legal serialisation by construction.
// Rows that would outgrow FIX_MAX_LG cells, big rows, rows that are missing and ops with y == 0 are handed back to the
// round loop through a new deferred list.
#ifndef SMX_FIX_MAX_LG
#define SMX_FIX_MAX_LG 9
#endif
constexpr uint32_t FIX_MAX_LG = SMX_FIX_MAX_LG;          // final table <= 512 cells: 2 x 4 KB + 2 KB of LDS per wave
constexpr uint32_t FIX_WAVES = 4;                        // waves (rows in flight) per workgroup
constexpr uint32_t FIX_NONE = 0xFFFFFFFFu;

// smallest lg with n <= 2^lg / 2 + 1 keys (src/smatrix.c:346 read backwards), at least `lg0`
__host__ __device__ inline uint32_t fix_lg_for(uint32_t n, uint32_t lg0) {
  uint32_t lg = lg0 < ROW_FIRST_LG ? ROW_FIRST_LG : lg0;
  while (lg < 31 && n > (1u << lg) / 2u + 1u) lg++;
  return lg;
}

// the largest table a row with `used` keys can end at when c ops are applied to it: every op a new key, plus one for the
// (0,v) cell of quirk Q1, which `used` leaves out until the next resize counts it (src/smatrix.c:353-354 vs :299)
__host__ __device__ inline uint32_t fix_bound_lg(uint32_t used, uint32_t c, uint32_t lg0) { return fix_lg_for(used + c + 1u, lg0); }

// pass 0: the rows the deferred ops name and the directory does not hold yet (src/smatrix.c:641-662).  k_prep's creation
// protocol reserves directory places and 16-cell blocks once per 1024 ops; on 15 M deferred ops that is 3 x 15 000
// atomics on three words (0.9 ms).  Here a workgroup folds 4096 ops by row id in LDS first, probes once per distinct
// id and reserves once per 4096 ops.  A refused reservation (directory at its limit) sets ctl->dir_full: the host
// rebuilds the directory and runs the pass again, exactly as for k_prep.
constexpr uint32_t FIXR_OPT = 16, FIXR_SLOTS = 8192;
__global__ __launch_bounds__(256) void k_fix_create(Ctl* ctl, DirSlot* dir, uint32_t dmask, uint32_t dir_limit, uint32_t n,
                                                    const uint32_t* defer, const uint32_t* __restrict__ xs, uint32_t st,
                                                    uint64_t arena_cap_units, FreeLists fl) {
  __shared__ uint32_t l_key[FIXR_SLOTS];                            // distinct row ids of the tile (FIX_NONE cannot use the set)
  __shared__ uint32_t l_cnt[8];                                     // [0] want, [1] reserved ok, [2] won, [3] popped, [4] old stack height
  __shared__ unsigned long long l_u0;
  for (uint32_t t0 = blockIdx.x * 256u * FIXR_OPT; t0 < n; t0 += gridDim.x * 256u * FIXR_OPT) {   // block-uniform
    for (uint32_t i = threadIdx.x; i < FIXR_SLOTS; i += 256) l_key[i] = FIX_NONE;
    if (threadIdx.x < 8) l_cnt[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll 4
    for (uint32_t k = 0; k < FIXR_OPT; k++) {
      const uint32_t t = t0 + k * 256u + threadIdx.x;
      if (t >= n) continue;
      const uint32_t X = xs[(size_t)defer[t] * st];
      if (X == FIX_NONE) continue;                                  // (left to the round loop's prep)
      uint32_t q = (X * 0x9E3779B1u) >> 19;                         // 13 bits
      for (;;) {
        const uint32_t prev = atomicCAS(&l_key[q], FIX_NONE, X);
        if (prev == FIX_NONE || prev == X) break;
        q = (q + 1) & (FIXR_SLOTS - 1);
      }
    }
    __syncthreads();
    // one lane per distinct id: is the row there?  (atomic loads: other workgroups create rows right now)
    uint32_t mine[FIXR_SLOTS / 256], hh[FIXR_SLOTS / 256], nm = 0;
    for (uint32_t i = threadIdx.x; i < FIXR_SLOTS; i += 256) {
      const uint32_t X = l_key[i];
      if (X == FIX_NONE) continue;
      uint32_t h = fmix32(X) & dmask;
      for (;;) {
        const uint64_t cur = ld_relaxed(reinterpret_cast<uint64_t*>(&dir[h]));
        if (cur == 0) { mine[nm] = X; hh[nm] = h; nm++; break; }
        if ((uint32_t)(cur >> 32) == X) break;
        h = (h + 1) & dmask;
      }
    }
    if (nm) atomicAdd(&l_cnt[0], nm);
    __syncthreads();
    if (threadIdx.x == 0 && l_cnt[0]) {
      const uint32_t before = atomicAdd(&ctl->dir_used, l_cnt[0]);
      if ((uint64_t)before + l_cnt[0] <= dir_limit) l_cnt[1] = 1;
      else { atomicSub(&ctl->dir_used, l_cnt[0]); ctl->dir_full = 1; }
    }
    __syncthreads();
    uint32_t wonm = 0, rank[FIXR_SLOTS / 256];
    if (l_cnt[1])
      for (uint32_t k = 0; k < nm; k++) {
        const uint32_t X = mine[k];
        const uint64_t want = (uint64_t)(META_USED | META_DIRTY | (ROW_FIRST_LG << META_LG_SHIFT)) | ((uint64_t)X << 32);
        uint32_t h = hh[k];
        for (;;) {
          const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&dir[h]), 0ull, (unsigned long long)want);
          if (prev == 0) { wonm |= 1u << k; rank[k] = atomicAdd(&l_cnt[2], 1u); hh[k] = h; break; }
          if ((uint32_t)(prev >> 32) == X) break;                  // another workgroup created it
          uint64_t cur;
          do {
            h = (h + 1) & dmask;
            cur = ld_relaxed(reinterpret_cast<uint64_t*>(&dir[h]));
          } while (cur != 0 && (uint32_t)(cur >> 32) != X);
          if (cur != 0) break;
        }
      }
    __syncthreads();
    if (threadIdx.x == 0 && l_cnt[1]) {
      const uint32_t n_res = l_cnt[0], n_won = l_cnt[2];
      if (n_res > n_won) atomicSub(&ctl->dir_used, n_res - n_won);
      uint32_t got = 0;
      int32_t top = 0;
      if (n_won) {
        top = atomicSub(&ctl->free_cnt[0], (int32_t)n_won);          // retired (zeroed) 16-cell blocks first
        got = top > 0 ? min((uint32_t)top, n_won) : 0u;
        if (got < n_won) atomicAdd(&ctl->free_cnt[0], (int32_t)(n_won - got));
        if (got < n_won)
          l_u0 = atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->arena_next), (unsigned long long)(n_won - got));
      }
      l_cnt[3] = got;
      l_cnt[4] = (uint32_t)top;
    }
    __syncthreads();
    for (uint32_t k = 0; k < nm; k++) {
      if (!(wonm & (1u << k))) continue;
      const uint32_t got = l_cnt[3];
      const uint64_t u = rank[k] < got ? fl.list[0][l_cnt[4] - 1u - rank[k]] : l_u0 + (rank[k] - got);
      if (u >= arena_cap_units) ctl->arena_oom = 1;                 // the host guarantees this never fires
      else __hip_atomic_store(&dir[hh[k]].base, (uint32_t)u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
}

// pass 0a (round 4): HOW MANY rows will pass 0 create?  The first batch of a matrix names 10^5..10^6 rows the directory
// (65 536 slots at open, src/smatrix.c:601) does not hold; pass 0 used to run into "directory full", the host rebuilt the
// directory four times as large and ran the pass again -- four passes over 15 M ops and three rebuilds for the first batch
// of config 2 (1.6 ms of its 13 ms).  Here the distinct MISSING row ids of the list are counted exactly -- the tile's ids
// folded in LDS like pass 0, each distinct id looked up once, the missing ones entered into a scratch set (64-bit slots,
// id + 1; a plain look before the compare-and-swap: hot ids are named by every tile) -- and the host sizes the directory
// ONCE.  The directory's layout is not observable through the API (the cmap rows), so sizing it in one step
// instead of four changes nothing a caller can see.
__global__ __launch_bounds__(256) void k_fix_count_rows(const DirSlot* dir, uint32_t dmask, uint32_t n, const uint32_t* defer,
                                                        const uint32_t* __restrict__ xs, uint32_t st, unsigned long long* set,
                                                        uint64_t set_mask, uint32_t* n_missing) {
  __shared__ uint32_t l_key[FIXR_SLOTS];
  __shared__ uint32_t l_won, l_none;
  for (uint32_t t0 = blockIdx.x * 256u * FIXR_OPT; t0 < n; t0 += gridDim.x * 256u * FIXR_OPT) {   // block-uniform
    for (uint32_t i = threadIdx.x; i < FIXR_SLOTS; i += 256) l_key[i] = FIX_NONE;
    if (threadIdx.x == 0) { l_won = 0; l_none = 0; }
    __syncthreads();
#pragma unroll 4
    for (uint32_t k = 0; k < FIXR_OPT; k++) {
      const uint32_t t = t0 + k * 256u + threadIdx.x;
      if (t >= n) continue;
      const uint32_t X = xs[(size_t)defer[t] * st];
      if (X == FIX_NONE) { l_none = 1; continue; }                  // (the id that cannot use the LDS set: counted as one more row)
      uint32_t q = (X * 0x9E3779B1u) >> 19;                         // 13 bits
      for (;;) {
        const uint32_t prev = atomicCAS(&l_key[q], FIX_NONE, X);
        if (prev == FIX_NONE || prev == X) break;
        q = (q + 1) & (FIXR_SLOTS - 1);
      }
    }
    __syncthreads();
    uint32_t won = 0;
    for (uint32_t i = threadIdx.x; i < FIXR_SLOTS; i += 256) {
      const uint32_t X = l_key[i];
      if (X == FIX_NONE) continue;
      uint32_t h = fmix32(X) & dmask;
      bool missing = false;
      for (;;) {
        const uint64_t cur = *reinterpret_cast<const uint64_t*>(&dir[h]);          // (the directory is stable during this pass)
        if (cur == 0) { missing = true; break; }
        if ((uint32_t)(cur >> 32) == X) break;
        h = (h + 1) & dmask;
      }
      if (!missing) continue;
      const unsigned long long key = (unsigned long long)X + 1ull;
      uint64_t g = splitmix_at(0x0d1full, X) & set_mask;
      for (;;) {
        unsigned long long prev = set[g];
        if (prev == 0ull) prev = atomicCAS(&set[g], 0ull, key);
        if (prev == 0ull) { won++; break; }
        if (prev == key) break;
        g = (g + 1) & set_mask;
      }
    }
    if (won) atomicAdd(&l_won, won);
    __syncthreads();
    if (threadIdx.x == 0 && l_won) atomicAdd(n_missing, l_won);
    if (threadIdx.x == 0 && l_none) n_missing[1] = 1;               // (benign race: all store 1)
    __syncthreads();
  }
}

// pass 0b (round 4): the rows pass 0a has just counted, created FROM ITS SET -- every non-empty slot of the scratch set is one
// distinct missing row id (id + 1), so creation is a sweep over the set's slots (268 MB for a 2^24-op batch, ~0.7 M rows)
// instead of a second fold of all 16.7 M ops (k_fix_create: 0.41 ms).  The directory was sized for them by the host, nobody
// else creates rows meanwhile, ids are distinct: a compare-and-swap on the first empty slot of the probe sequence always
// wins in the end.  Reservations (directory count, retired 16-cell blocks, arena units) once per workgroup, as in pass 0.
constexpr uint32_t FIXS_OPT = 16;
__global__ __launch_bounds__(256) void k_fix_create_set(Ctl* ctl, DirSlot* dir, uint32_t dmask, const unsigned long long* __restrict__ set,
                                                        uint64_t set_slots, uint64_t arena_cap_units, FreeLists fl) {
  // a workgroup owns one contiguous range of the set: it counts the range's rows first, reserves ONCE (three words that
  // every workgroup needs: one reservation per 4096 slots queued 8 192 x 3 same-address atomics, 1 ms), then creates
  __shared__ uint32_t l_total, l_next, l_got, l_top;
  __shared__ unsigned long long l_u0;
  const uint64_t chunk = ((set_slots + gridDim.x - 1) / gridDim.x + 255u) & ~255ull;
  const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = min(lo + chunk, set_slots);
  if (threadIdx.x == 0) { l_total = 0; l_next = 0; }
  __syncthreads();
  uint32_t cnt = 0;
  for (uint64_t t = lo + threadIdx.x; t < hi; t += 256) cnt += set[t] != 0ull;
  if (cnt) atomicAdd(&l_total, cnt);
  __syncthreads();
  if (threadIdx.x == 0 && l_total) {
    const uint32_t n_new = l_total;
    atomicAdd(&ctl->dir_used, n_new);                                // (the host sized the directory for exactly these rows)
    const int32_t top = atomicSub(&ctl->free_cnt[0], (int32_t)n_new);            // retired (zeroed) 16-cell blocks first
    const uint32_t got = top > 0 ? min((uint32_t)top, n_new) : 0u;
    if (got < n_new) atomicAdd(&ctl->free_cnt[0], (int32_t)(n_new - got));
    if (got < n_new) l_u0 = atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->arena_next), (unsigned long long)(n_new - got));
    l_got = got;
    l_top = (uint32_t)top;
  }
  __syncthreads();
  if (l_total == 0) return;
  for (uint64_t t = lo + threadIdx.x; t < hi; t += 256) {
    const unsigned long long key = set[t];
    if (!key) continue;
    const uint32_t X = (uint32_t)(key - 1ull), rank = atomicAdd(&l_next, 1u);
    const uint64_t want = (uint64_t)(META_USED | META_DIRTY | (ROW_FIRST_LG << META_LG_SHIFT)) | ((uint64_t)X << 32);
    uint32_t h = fmix32(X) & dmask;
    for (;;) {
      const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&dir[h]), 0ull, (unsigned long long)want);
      if (prev == 0) break;
      h = (h + 1) & dmask;                                           // (another new row took it: ids are distinct, walk on)
    }
    const uint64_t u = rank < l_got ? fl.list[0][l_top - 1u - rank] : l_u0 + (rank - l_got);
    if (u >= arena_cap_units) ctl->arena_oom = 1;                   // the host guarantees this never fires
    else __hip_atomic_store(&dir[h].base, (uint32_t)u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// pass 1: ops per directory slot; where[t] = the slot of deferred op t, or FIX_NONE for an op the bulk path does not take.
// A workgroup first folds its 2048 ops by slot in an LDS table (bulk loads name the same row many times in a row:
// the config-3 stream has 115 consecutive ops per row), then adds each distinct slot's count with ONE global atomic.
constexpr uint32_t FIXC_OPT = 8, FIXC_SLOTS = 4096;
__global__ __launch_bounds__(256) void k_fix_count(Ctl* ctl, DirSlot* dir, uint32_t dmask, uint32_t n, const uint32_t* defer,
                                                   const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys, uint32_t st,
                                                   uint32_t* cnt, uint32_t* where, uint32_t* defer_out, uint32_t* touched,
                                                   uint32_t* pos_of, uint32_t* rank_of) {
  // touched[0 .. ctl->n_tasks): the directory slots with pending ops (a slot is listed by whoever raises its count from
  // 0), pos_of[h] = its place in that list -- everything after this pass works on that list, not on the directory
  // rank_of[t] (round 4): the op's place among its row's pending ops -- the add that raises the row's count returns where this
  // workgroup's share of the row starts, the LDS add the op's place inside the share.  The scatter pass used to fold its
  // ops by row once more and reserve the same ranges again on a cursor word per row (8 192 workgroups on the hottest rows'
  // words: 1.5 ms of the first batch of config 2); now it only reads the rank
  __shared__ uint32_t l_key[FIXC_SLOTS], l_cnt[FIXC_SLOTS];
  __shared__ uint32_t l_n, l_base, l_first, l_fbase;
  for (uint32_t t0 = blockIdx.x * 256u * FIXC_OPT; t0 < n; t0 += gridDim.x * 256u * FIXC_OPT) {   // block-uniform
    for (uint32_t i = threadIdx.x; i < FIXC_SLOTS; i += 256) { l_key[i] = FIX_NONE; l_cnt[i] = 0; }
    if (threadIdx.x == 0) { l_n = 0; l_first = 0; }
    __syncthreads();
    uint32_t jb[FIXC_OPT], rk[FIXC_OPT], qb[FIXC_OPT];
    uint32_t backm = 0, takem = 0;
#pragma unroll
    for (uint32_t k = 0; k < FIXC_OPT; k++) {
      const uint32_t t = t0 + k * 256u + threadIdx.x;
      if (t >= n) continue;
      const uint32_t j = defer[t];
      jb[k] = j;
      uint4 sn;
      DirSlot* d = ys[(size_t)j * st] != 0 ? dir_find(dir, dmask, xs[(size_t)j * st], &sn) : nullptr;
      if (d && sn.z != 0 && meta_lg(sn.x) <= FIX_MAX_LG) {
        const uint32_t h = (uint32_t)(d - dir);
        where[t] = h;
        uint32_t q = (h * 0x9E3779B1u) >> 20;                       // 12 bits
        for (;;) {
          const uint32_t prev = atomicCAS(&l_key[q], FIX_NONE, h);
          if (prev == FIX_NONE || prev == h) break;
          q = (q + 1) & (FIXC_SLOTS - 1);
        }
        qb[k] = q;
        rk[k] = atomicAdd(&l_cnt[q], 1u);                           // the op's place in this workgroup's share of the row
        takem |= 1u << k;
      } else {
        where[t] = FIX_NONE;
        backm |= 1u << k;
        rk[k] = atomicAdd(&l_n, 1u);
      }
    }
    __syncthreads();
    uint32_t fh[FIXC_SLOTS / 256], fr[FIXC_SLOTS / 256], nf = 0;     // slots this lane raised from 0: they join the list
    for (uint32_t i = threadIdx.x; i < FIXC_SLOTS; i += 256)
      if (l_cnt[i]) {
        const uint32_t start = atomicAdd(&cnt[l_key[i]], l_cnt[i]);   // where this workgroup's share of the row starts
        l_cnt[i] = start;
        if (start == 0) { fh[nf] = l_key[i]; fr[nf] = atomicAdd(&l_first, 1u); nf++; }
      }
    __syncthreads();
    if (threadIdx.x == 0 && l_n) l_base = atomicAdd(&ctl->n_defer, l_n);
    if (threadIdx.x == 0 && l_first) l_fbase = atomicAdd(&ctl->n_tasks, l_first);
    __syncthreads();
    for (uint32_t k = 0; k < nf; k++) { touched[l_fbase + fr[k]] = fh[k]; pos_of[fh[k]] = l_fbase + fr[k]; }
#pragma unroll
    for (uint32_t k = 0; k < FIXC_OPT; k++) {
      if (backm & (1u << k)) defer_out[l_base + rk[k]] = jb[k];
      if (takem & (1u << k)) rank_of[t0 + k * 256u + threadIdx.x] = l_cnt[qb[k]] + rk[k];
    }
    __syncthreads();
  }
}

// pass 2: per directory slot {ops, units of a new block} -> exclusive scan (three launches: tiles, tile totals, add).
// A row is ELIGIBLE if its table can end at no more than 2^FIX_MAX_LG cells even if every pending op is a new key;
// it gets a block of that bound's size class when the bound exceeds its present size.
constexpr uint32_t SCAN_TILE = 2048;
// not for the bulk path: a row whose table could end above 2^FIX_MAX_LG cells, or that the round loop has flagged
__device__ inline bool fix_row_eligible(const DirSlot& d, uint32_t c) {
  return fix_bound_lg(d.used, c, meta_lg(d.meta)) <= FIX_MAX_LG && !(d.meta & (META_GROW | META_REBAL));
}
// Round 4: a row that is still SMALL but could outgrow the path (the hot rows of a first batch: millions of ops on a
// 16-cell table) gives the path its first FIX_PART_OPS ops: the wide pass takes the row as far as 2^FIX_MAX_LG cells filled to
// the reference's threshold and hands the rest back.  Any subset of a batch's ops may come first in its serialisation, so
// this is the state a cold start reaches after its first five doubling rounds (16 -> 512 cells) -- without those rounds
// (the first batch of config 2: 3 578 such rows, 5 of its 16 rounds).
constexpr uint32_t FIX_PART_OPS = 2048;
__device__ inline bool fix_row_partial(const DirSlot& d, uint32_t c) {
  return meta_lg(d.meta) <= FIX_MAX_LG && !(d.meta & (META_GROW | META_REBAL)) && fix_bound_lg(d.used, c, meta_lg(d.meta)) > FIX_MAX_LG;
}
__device__ inline uint64_t fix_elem(const DirSlot* dir, const uint32_t* cnt, const uint32_t* touched, uint32_t i, uint32_t nrows,
                                    uint64_t* wide) {
  if (i >= nrows) return 0;
  const uint32_t h = touched[i];
  const uint32_t c = cnt[h];
  const DirSlot d = dir[h];
  const uint32_t lg = meta_lg(d.meta), lgb = fix_bound_lg(d.used, c, lg);
  if (fix_row_partial(d, c)) {                                     // its first ops, and a block of the largest class
    *wide = 1;
    return (uint64_t)min(c, FIX_PART_OPS) | ((uint64_t)(lg < FIX_MAX_LG ? (uint32_t)units_of_lg(FIX_MAX_LG) : 0u) << 32);
  }
  if (!fix_row_eligible(d, c)) return 0;                           // its ops go straight back to the list (k_fix_scatter)
  if (lgb == FIX_MAX_LG) *wide = 1;                                // the second k_fix_rows pass has work (benign race: all store 1)
  return (uint64_t)c | ((uint64_t)(lgb > lg ? (uint32_t)units_of_lg(lgb) : 0u) << 32);
}
__global__ __launch_bounds__(256) void k_fix_scan_tiles(const Ctl* ctl, const DirSlot* dir, const uint32_t* cnt, const uint32_t* touched,
                                                        uint64_t* excl, uint64_t* tile_sum, uint64_t* wide) {
  __shared__ uint64_t l_w[4];
  const uint32_t dir_size = aload(&ctl->n_tasks);                  // (the list's length; the name is kept for the code below)
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8u;
  uint64_t v[8], run = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { v[k] = fix_elem(dir, cnt, touched, base + k, dir_size, wide); run += v[k]; }
  // both halves stay below 2^32 over the whole directory (ops < 2^32, units < 2^32): the packed sums never carry across
  uint64_t inc = run;
  const uint32_t lane = __lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t up = ((uint64_t)(uint32_t)__shfl_up((int)(inc >> 32), o) << 32) | (uint32_t)__shfl_up((int)inc, o);
    if (lane >= (uint32_t)o) inc += up;
  }
  if (lane == 63) l_w[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint64_t before = 0;
  for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += l_w[w];
  uint64_t e = before + inc - run;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (base + k < dir_size) excl[base + k] = e;
    e += v[k];
  }
  if (threadIdx.x == 255) tile_sum[blockIdx.x] = before + inc;
}
__global__ __launch_bounds__(1024) void k_fix_scan_tops(uint64_t* tile_sum, uint32_t ntiles, uint64_t* total) {
  __shared__ uint64_t l_w[16];
  __shared__ uint64_t l_run;
  if (threadIdx.x == 0) l_run = 0;
  __syncthreads();
  const uint32_t lane = __lane_id(), w = threadIdx.x >> 6;
  for (uint32_t t0 = 0; t0 < ntiles; t0 += 1024) {                  // block-uniform
    const uint32_t t = t0 + threadIdx.x;
    const uint64_t v = t < ntiles ? tile_sum[t] : 0;
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t up = ((uint64_t)(uint32_t)__shfl_up((int)(inc >> 32), o) << 32) | (uint32_t)__shfl_up((int)inc, o);
      if (lane >= (uint32_t)o) inc += up;
    }
    if (lane == 63) l_w[w] = inc;
    __syncthreads();
    uint64_t before = l_run;
    for (uint32_t k = 0; k < w; k++) before += l_w[k];
    if (t < ntiles) tile_sum[t] = before + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) l_run = before + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = l_run;
}
__global__ __launch_bounds__(256) void k_fix_scan_add(const Ctl* ctl, uint64_t* excl, const uint64_t* tile_sum) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < aload(&ctl->n_tasks)) excl[i] += tile_sum[i / SCAN_TILE];
}

// pass 3: the op indices, row by row.  Every op knows its place among its row's ops (rank_of, pass 1): ops of eligible rows
// go to their row's range of `grouped`; of a PARTIAL row (fix_row_partial) the first FIX_PART_OPS; everything else -- a hot row's
// millions among them -- goes straight back to the round loop's list, one reservation per workgroup (copying them back row by
// row, one wave per row, took 27 ms for the 4 M ops of one hot item).  No fold and no per-row cursor any more (round 4).
__global__ __launch_bounds__(256) void k_fix_scatter(Ctl* ctl, const DirSlot* dir, const uint32_t* cnt, uint32_t n, const uint32_t* defer,
                                                     const uint32_t* where, const uint64_t* excl, const uint32_t* pos_of, const uint32_t* rank_of,
                                                     uint32_t* grouped, uint32_t* defer_out) {
  __shared__ uint32_t l_nback, l_bbase;
  for (uint32_t t0 = blockIdx.x * 256u * FIXC_OPT; t0 < n; t0 += gridDim.x * 256u * FIXC_OPT) {   // block-uniform
    if (threadIdx.x == 0) l_nback = 0;
    __syncthreads();
    uint32_t jb[FIXC_OPT], at[FIXC_OPT];
    uint32_t backm = 0, takem = 0;
#pragma unroll
    for (uint32_t k = 0; k < FIXC_OPT; k++) {
      const uint32_t t = t0 + k * 256u + threadIdx.x;
      const uint32_t h = t < n ? where[t] : FIX_NONE;
      if (h == FIX_NONE) continue;                                    // (pass 1 has sent it back already)
      jb[k] = defer[t];
      const DirSlot d = dir[h];
      const uint32_t c = cnt[h], r = rank_of[t];
      const bool part = fix_row_partial(d, c);
      if (part ? r < FIX_PART_OPS : fix_row_eligible(d, c)) {
        at[k] = (uint32_t)excl[pos_of[h]] + r;
        takem |= 1u << k;
      } else {
        backm |= 1u << k;
      }
    }
    // the ops that go back: one list reservation per workgroup (a wave's share through one LDS add)
    const uint32_t mine = (uint32_t)__popc(backm);
    const uint64_t lanes_before = (1ull << __lane_id()) - 1ull;
    uint32_t wave_tot = mine, pre = 0;
    // prefix over the wave (six shuffle steps) -- lanes hold 0..FIXC_OPT ops each
    {
      uint32_t incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if (__lane_id() >= (uint32_t)o) incl += up;
      }
      pre = incl - mine;
      wave_tot = (uint32_t)__shfl((int)incl, 63);
      (void)lanes_before;
    }
    uint32_t wbase = 0;
    if (__lane_id() == 0 && wave_tot) wbase = atomicAdd(&l_nback, wave_tot);
    wbase = (uint32_t)__shfl((int)wbase, 0);
    __syncthreads();
    if (threadIdx.x == 0 && l_nback) l_bbase = atomicAdd(&ctl->n_defer, l_nback);
    __syncthreads();
    uint32_t o = l_bbase + wbase + pre;
#pragma unroll
    for (uint32_t k = 0; k < FIXC_OPT; k++) {
      if (takem & (1u << k)) grouped[at[k]] = jb[k];
      else if (backm & (1u << k)) defer_out[o++] = jb[k];
    }
    __syncthreads();
  }
}

// one lane, on the LDS table: the reference's probe (src/smatrix.c:363-380)
__device__ inline uint32_t fix_probe(const uint64_t* T, uint32_t mask, uint32_t key) {
  uint32_t i = key & mask;
  while (cell_key(T[i]) != key && T[i] != 0) i = (i + 1) & mask;
  return i;
}

// pass 4: one wave per row with pending ops, the row's table in LDS from the first op to the last.  64 ops at a time:
// every lane probes for its key; hits and as many new keys as the threshold admits are applied together (LDS CAS
// claims + LDS adds = some order of those ops in which every insert saw used <= size/2); if new keys are left over the
// table is doubled -- priority probing on old slot indices (LDS atomicMin, as in k_grow_lds) gives the layout of the
// reference's re-insertion in old slot order; a table that holds a key twice (quirk fallout) is redone by one lane
// exactly as smatrix_rmap_resize does it -- and the rest goes on.  (A first version ran the reference's code with one
// lane per row: 6.0 ms per 15 M-op batch; one op at a time with wave-wide probing: 3.4 ms.)
// Two instantiations share the rows: MAXLG = FIX_MAX_LG - 1 takes every row that can end at <= 256 cells (5 KB of LDS
// per wave: 28 waves per CU) and clears the counts of the ineligible ones (k_fix_scatter sent their ops back); MAXLG = FIX_MAX_LG takes the rows that may reach 512.
template <int OP, uint32_t MAXLG>
__global__ __launch_bounds__(64 * FIX_WAVES) void k_fix_rows(
    Ctl* ctl, DirSlot* dir, const uint32_t* touched, uint8_t* arena, uint32_t* cnt, uint32_t* cursor, const uint64_t* excl,
    const uint32_t* grouped, const uint32_t* __restrict__ ys, const uint32_t* __restrict__ vs, uint32_t st,
    uint32_t* __restrict__ out, uint32_t* defer_out, uint64_t new_base0, FreeLists fl) {
  static_assert(OP != OP_GET, "the bulk path takes writers (set: any value lands here, duplicates are resolved after the rounds, k_set_*)");
  static_assert(MAXLG == FIX_MAX_LG || MAXLG + 1 == FIX_MAX_LG, "two passes");
  constexpr uint32_t SMAX = 1u << MAXLG;
  __shared__ uint64_t l_tab[FIX_WAVES][2][SMAX];
  __shared__ uint32_t l_idx[FIX_WAVES][SMAX];                       // resize: old slot index per new slot
  __shared__ uint32_t l_ret[FIX_WAVES][64], l_rcls[FIX_WAVES][64];  // blocks this wave has retired: base, size class
  __shared__ uint32_t l_nret[FIX_WAVES];
  const uint32_t lane = __lane_id(), w = threadIdx.x >> 6;
  const uint32_t wave = blockIdx.x * FIX_WAVES + w, nwaves = gridDim.x * FIX_WAVES;
  const uint64_t lt = (1ull << lane) - 1ull;
  if (lane == 0) l_nret[w] = 0;
  auto wsync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // retired 16*2^c-cell blocks go back to their size class's stack (zeroed); one list reservation per 64 of them
  auto flush_retired = [&]() {
    const uint32_t nr = l_nret[w];
    for (uint32_t c = 0; c <= MAXLG - ROW_FIRST_LG; c++) {
      const bool mine = lane < nr && l_rcls[w][lane] == c;
      const uint64_t m = __ballot(mine);
      if (!m) continue;
      const int lead = __ffsll((unsigned long long)m) - 1;
      uint32_t at = 0;
      if ((int)lane == lead) at = (uint32_t)atomicAdd(&ctl->free_cnt[c], (int32_t)__popcll(m));
      at = (uint32_t)__shfl((int)at, lead);
      if (mine) fl.list[c][at + (uint32_t)__popcll(m & lt)] = l_ret[w][lane];
    }
    wsync();
    if (lane == 0) l_nret[w] = 0;
    wsync();
  };
  const uint32_t nrows = aload(&ctl->n_tasks);
  for (uint32_t ri = wave; ri < nrows; ri += nwaves) {                                    // wave-uniform
    {
      const uint32_t h = touched[ri];
      const uint32_t c_all = cnt[h];
      if (c_all == 0) continue;                                     // the other pass has taken it
      const DirSlot d = dir[h];
      const uint64_t e = excl[ri];
      const uint32_t p0 = (uint32_t)e;
      const uint32_t lg0 = meta_lg(d.meta);
      const bool partial = fix_row_partial(d, c_all);               // the row's first ops only, up to 2^FIX_MAX_LG cells (see fix_row_partial)
      const uint32_t lgb = partial ? FIX_MAX_LG : fix_bound_lg(d.used, c_all, lg0);
      if (!partial && !fix_row_eligible(d, c_all)) {
        // not for this path: k_fix_scatter has sent the row's ops back to the round loop already (the first pass,
        // which always runs, clears the row's count)
        if (MAXLG == FIX_MAX_LG - 1 && lane == 0) { cnt[h] = 0; cursor[h] = 0; }
        continue;
      }
      if (MAXLG == FIX_MAX_LG ? lgb != FIX_MAX_LG : lgb == FIX_MAX_LG) continue;         // the other pass's row
      wsync();
      if (lane == 0) { cnt[h] = 0; cursor[h] = 0; }                 // taken; and both arrays are all-zero again for the next batch
      const uint32_t c = partial ? min(c_all, FIX_PART_OPS) : c_all;    // the ops this wave has in `grouped`
      bool handed_back = false;
      uint32_t cur = 0;                                             // which of the two LDS tables is live
      uint64_t* cells = row_cells(arena, d.base);
      uint32_t lg = lg0, used = d.used;
      for (uint32_t i = lane; i < (1u << lg); i += 64) l_tab[w][0][i] = cells[i];
      wsync();
      for (uint32_t c0 = 0; c0 < c && !handed_back; c0 += 64) {
        // this lane's op of the chunk; its result ends up in `res`
        uint32_t j = 0, Yl = 0, Vl = 0, res = 0;
        if (c0 + lane < c) {
          j = grouped[p0 + c0 + lane];
          Yl = ys[(size_t)j * st];
          Vl = vs[(size_t)j * st];
        }
        // FILL -> GROW -> FILL: all hits of the chunk and as many of its new keys as the reference's threshold leaves
        // room for go in together (LDS CAS claims, LDS adds: some order of these ops -- each insert at a moment when
        // used <= size/2 held); when keys are left and the room is gone the table is doubled and the rest goes on
        bool pending = c0 + lane < c;
        while (__any(pending)) {                                    // wave-uniform
          uint64_t* T = l_tab[w][cur];
          const uint32_t S = 1u << lg, mask = S - 1u;
          bool absent = false;
          uint32_t slot = Yl & mask;
          bool stuck = false;                                       // no empty cell at all (only a foreign, over-full table): resize first
          if (pending)
            for (uint32_t steps = 0;; steps++) {                    // smatrix_rmap_probe, src/smatrix.c:363-380
              const uint64_t cc = T[slot];
              if (cell_key(cc) == Yl) break;
              if (cc == 0) { absent = true; break; }
              if (steps > mask) { absent = true; stuck = true; break; }
              slot = (slot + 1) & mask;
            }
          const uint32_t room = used <= S / 2u ? S / 2u + 1u - used : 0u;   // inserts the threshold still admits (:346)
          const uint64_t ma = __ballot(pending && absent);
          const bool go = pending && !stuck && (!absent || (uint32_t)__popcll(ma & lt) < room);
          bool inserted = false;
          if (go) {
            while (absent) {
              const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&T[slot]), 0ull,
                                              (unsigned long long)pack_cell(Yl, 0));      // :354-356
              if (prev == 0) { inserted = true; break; }
              for (;;) {                                            // the slot went to another lane: look again from here
                const uint64_t cc = T[slot];
                if (cell_key(cc) == Yl) { absent = false; break; }  // ... to a lane with the same key
                if (cc == 0) break;
                slot = (slot + 1) & mask;
              }
            }
            uint32_t* vp = reinterpret_cast<uint32_t*>(&T[slot]) + 1;
            if (OP == OP_SET) { atomicExch(vp, Vl); res = Vl; }                           // :230
            else res = OP == OP_INCR ? atomicAdd(vp, Vl) + Vl : atomicSub(vp, Vl) - Vl;   // :241 / :252
            pending = false;
          }
          used += (uint32_t)__popcll(__ballot(inserted));
          wsync();
          if (!__any(go) && __any(pending)) {
            if (partial && lg == MAXLG) {
              // a partial row has reached 2^FIX_MAX_LG cells at the reference's threshold: the ops that are left -- this
              // chunk's pending ones and the chunks behind it -- go back to the round loop, which doubles the row on
              const uint64_t pm = __ballot(pending);
              const uint32_t np = (uint32_t)__popcll(pm), rest = c - min(c0 + 64u, c);
              uint32_t at = 0;
              if (lane == 0) at = atomicAdd(&ctl->n_defer, np + rest);
              at = (uint32_t)__shfl((int)at, 0);
              if (pending) defer_out[at + (uint32_t)__popcll(pm & lt)] = j;
              for (uint32_t i = lane; i < rest; i += 64) defer_out[at + np + i] = grouped[p0 + c0 + 64u + i];
              handed_back = true;
              break;
            }
            // ---- smatrix_rmap_resize (src/smatrix.c:383-416): S -> 2S, old slot order
            uint64_t* N = l_tab[w][cur ^ 1u];
            const uint32_t nmask = 2u * S - 1u;
            for (uint32_t q = lane; q <= nmask; q += 64) l_idx[w][q] = FIX_NONE;
            wsync();
            uint32_t moved = 0;
            for (uint32_t p = lane; p < S; p += 64) {
              const uint64_t cc = T[p];
              if (cc == 0) continue;
              moved++;
              uint32_t carry = p, i2 = cell_key(cc) & nmask;
              for (;;) {
                const uint32_t prev = atomicMin(&l_idx[w][i2], carry);
                if (prev == FIX_NONE) break;
                if (prev > carry) carry = prev;                     // evicted a later cell: carry it onward
                i2 = (i2 + 1) & nmask;
              }
            }
            wsync();
            bool dup = false;                                       // a key that a probe from its home finds elsewhere first
            for (uint32_t q = lane; q <= nmask; q += 64) {
              const uint32_t r = l_idx[w][q];
              if (r == FIX_NONE) continue;
              const uint32_t key = cell_key(T[r]);
              uint32_t i2 = key & nmask;
              while (i2 != q) {
                const uint32_t r2 = l_idx[w][i2];
                if (r2 == FIX_NONE || cell_key(T[r2]) == key) break;
                i2 = (i2 + 1) & nmask;
              }
              if (i2 != q) dup = true;
            }
            if (!__any(dup)) {
              for (uint32_t q = lane; q <= nmask; q += 64) {
                const uint32_t r = l_idx[w][q];
                N[q] = r == FIX_NONE ? 0ull : T[r];
              }
              for (int o = 32; o > 0; o >>= 1) moved += (uint32_t)__shfl_xor((int)moved, o);
              used = moved;
            } else {
              uint32_t nu = 0;
              if (lane == 0) {                                      // the reference's way, one cell after the other
                for (uint32_t q = 0; q <= nmask; q++) N[q] = 0;
                for (uint32_t q = 0; q <= mask; q++) {
                  const uint64_t cc = T[q];
                  if (cc == 0) continue;
                  const uint32_t z = fix_probe(N, nmask, cell_key(cc));
                  if (cell_key(N[z]) == 0 || cell_key(N[z]) != cell_key(cc)) nu++;       // :353-354
                  N[z] = cc;
                }
              }
              used = (uint32_t)__shfl((int)nu, 0);
            }
            wsync();
            cur ^= 1u;
            lg++;
          }
        }
        if (c0 + lane < c && !pending) out[j] = res;                  // (handed-back ops get their results from the round loop)
      }
      uint64_t* T = l_tab[w][cur];
      uint64_t* dst = cells;
      if (lg != lg0) {
        dst = row_cells(arena, (uint32_t)(new_base0 + (e >> 32)));
        for (uint32_t i = lane; i < (1u << lg0); i += 64) cells[i] = 0;                   // retired blocks are all-empty
        if (lane == 0) {
          l_ret[w][l_nret[w]] = d.base;
          l_rcls[w][l_nret[w]] = lg0 - ROW_FIRST_LG;
          l_nret[w]++;
        }
      }
      for (uint32_t i = lane; i < (1u << lg); i += 64) dst[i] = T[i];
      if (lane == 0) {
        DirSlot nd;
        nd.meta = META_USED | META_DIRTY | (lg << META_LG_SHIFT);
        nd.x = d.x;
        nd.base = lg != lg0 ? (uint32_t)(new_base0 + (e >> 32)) : d.base;
        nd.used = used;
        dir[h] = nd;
      }
      wsync();
      if (l_nret[w] == 64) flush_retired();
    }
  }
  if (l_nret[w]) flush_retired();
}
