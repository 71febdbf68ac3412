// kernels/layout.hpp -- the HBM layout (directory slots, row blocks, sub-counters, at-home bitmaps), the control block, small helpers.
// A fragment of smx_kernels.hpp (round 5: the 4 500-line header split by concern, no kernel changed): included there, in order,
// INSIDE namespace smx; not a header of its own.

// ---- layout -----------------------------------------------------------------

struct DirSlot {
  uint32_t meta;   // bit0 USED | bits 8..13 log2(row size) | bit 16 GROW pending
  uint32_t x;      // row id
  uint32_t base;   // row block, in 128-byte arena units (0 = not yet allocated)
  uint32_t used;   // the reference's rmap->used
};
static_assert(sizeof(DirSlot) == 16, "DirSlot must be 16 bytes");

constexpr uint32_t META_USED = 1u;
constexpr uint32_t META_GROW = 1u << 16;
// the row changed since it was last written to the backing file (the reference's SMATRIX_RMAP_FLAG_DIRTY,
// src/smatrix.h:17, set by smatrix_rmap_sync_defer :418-425): set by every writer, by row creation and by growth;
// collected and cleared by the flush (k_dirty_collect).  Writers store it only when their snapshot of the slot does
// not show it yet, so a row pays one extra 4-byte store per flush interval (in memory mode: once).
constexpr uint32_t META_DIRTY = 1u << 18;
constexpr uint32_t META_LG_SHIFT = 8;
constexpr uint32_t ROW_FIRST_LG = 4;  // SMATRIX_RMAP_INITIAL_SIZE 16, src/smatrix.h:21
constexpr uint32_t UNIT_BYTES = 128;  // 16 cells

__host__ __device__ inline uint32_t meta_lg(uint32_t meta) { return (meta >> META_LG_SHIFT) & 63u; }
__host__ __device__ inline uint64_t units_of_lg(uint32_t lg) { return 1ull << (lg - ROW_FIRST_LG); }

// Big rows (>= 2^BIG_LG cells) count their inserts in SUBS sub-counters, one 64-byte line each,
// placed right behind the row's cells.  Under Zipf a single row takes 12 % of a batch; its one
// `used` word then serialises ~50 k returning atomics per batch at the memory side (~34 ns each,
// measured: +1.9 ms on a 1.8 ms kernel).  The reference's rule "insert only while used <= size/2"
// (src/smatrix.c:346) stays exact: the room left below the threshold is PARTITIONED into per
// sub-counter quotas, each enforced with its own returning atomic, so the row can never hold more
// than size/2+1 keys; `used` in the directory is the count at the last fold and
//   rowlen = used + sum(cnt)   at any quiescent point.
constexpr uint32_t META_REBAL = 1u << 17;  // quotas want re-partitioning (k_rebal)
#ifndef SMX_BIG_LG
#define SMX_BIG_LG 15
#endif
constexpr uint32_t BIG_LG = SMX_BIG_LG;
constexpr uint32_t SUBS = 64;     // 4 KB per big row (>= 256 KB of cells)
constexpr uint32_t SUB_UNITS = SUBS * 64 / 128;
struct SubCtr { uint32_t cnt, quota, pad[14]; };
static_assert(sizeof(SubCtr) == 64, "one sub-counter per 64-byte line");

// Rows of >= 2^HOME_LG cells carry an AT-HOME BITMAP behind their cells (and sub-counter lines): one bit per cell, set iff the
// cell holds a key whose home is that very slot (key mod size == slot, key != 0).  Row tables keep the reference's identity
// hash (src/smatrix.c:366), so dense ids build long runs of such cells, and a key that wraps onto a run walks to its end
// (src/smatrix.c:369-377).  A key whose home cell holds ANOTHER key can only sit in a cell that is NOT at home, so a probe may
// step over set bits 64 cells per 8-byte load without looking at the cells.  The bitmap is an accelerator, never a structure:
// a SET bit is always true (keys never leave their cell; key 0 -- whose (0,v) cell can turn back into an empty one, quirk
// Q1 -- never gets one), a CLEAR bit says nothing (the cell is loaded).  Bits are set by the inserting kernels of clustered
// matrices, written whole by growth (k_grow_move_home, k_grow_lds) and by k_home_rebuild; blocks are handed out zeroed.  The
// bitmap never reaches the backing file.
#ifndef SMX_HOME_LG
#define SMX_HOME_LG 12
#endif
constexpr uint32_t HOME_LG = SMX_HOME_LG;
static_assert(HOME_LG >= 10, "the bitmap of the smallest such row fills whole 128-byte units");
__host__ __device__ inline uint64_t home_units(uint32_t lg) { return lg >= HOME_LG ? 1ull << (lg - 10) : 0; }
__host__ __device__ inline uint64_t block_units(uint32_t lg) {
  return units_of_lg(lg) + (lg >= BIG_LG ? SUB_UNITS : 0) + home_units(lg);
}
// Endgame: with little room left an even split leaves every sub-counter one or two tickets, the patient
// retry (own share + three others) misses most of what remains, and the row bounces through one
// re-partition round after the other before it finally grows.  Below SUBS_ENDGAME tickets the whole room
// goes to sub-counter 0, which the patient path always tries last: the next round drains it exactly.
#ifndef SMX_ENDGAME
#define SMX_ENDGAME 8
#endif
constexpr uint32_t SUBS_ENDGAME = SMX_ENDGAME * SUBS;
__host__ __device__ inline void subs_init(SubCtr* sc, uint32_t room) {
  for (uint32_t k = 0; k < SUBS; k++) {
    sc[k].cnt = 0;
    sc[k].quota = room < SUBS_ENDGAME ? (k == 0 ? room : 0u) : room / SUBS + (k < room % SUBS ? 1u : 0u);
  }
  sc[0].pad[0] = 0;                                // "every share is used up" (sub_ticket_anywhere)
}

enum Op : int { OP_GET = 0, OP_SET = 1, OP_INCR = 2, OP_DECR = 3 };


// device-side control block, one per matrix.  The first part is zeroed at the start of every round;
// the persistent part is owned by the device between readbacks.
constexpr uint32_t N_CLASSES = 28;     // row block size classes: 16 * 2^c cells, c = log2(size) - 4
struct Ctl {
  // ---- per round ----
  uint32_t n_defer;      // ops deferred by the current op round
  uint32_t n_tasks;      // rows flagged for growth by prep
  uint32_t dir_full;     // prep refused a row creation (directory at its limit)
  uint32_t arena_oom;    // an allocation did not fit (host maps more and reruns)
  uint64_t grow_units;   // units the flagged growths will need (upper bound: recycled blocks need none)
  uint32_t n_chunks;     // 64-slot chunks over all growth tasks (old tables)
  uint32_t n_chunks_new; // same over the new tables
  uint32_t n_rebal;      // big rows whose sub-counter quotas want re-partitioning
  uint32_t n_kind[4];    // growth tasks by kind (grow_kind): LDS by wave / workgroup / large workgroup, chunked
  uint32_t n_long;       // the folding kernel deferred ops whose probe outran its budget (the lane-per-op kernel takes them)
  uint32_t n_long_ops;   // ... how many ops the lane-per-op WRITE kernel finished through the wave-cooperative probe in this round: a few on
                         // any large table at load 1/2, percents of a batch on a clustered one (dense ids) -- Matrix::clustered
  uint32_t n_absent;     // (clustered folding kernel, ArenaHead::absent_list) ops deferred into the second list: they wait for prep, the pass in
                         // front of it does not see them; k_round_advance starts that pass's output list behind them
  // ---- persistent ----
  uint32_t dir_used;     // rows in the directory
  uint32_t pad1;
  uint64_t arena_next;   // bump pointer, units
  int32_t  free_cnt[N_CLASSES];   // retired row blocks ready for reuse, per size class (stack heights)
  // ---- the device-driven round (k_round_advance; smx_runtime.hip "speculative chain") ----
  uint32_t n_prev;       // ops the previous op round deferred = the length of the list the next op round reads
  uint32_t spec_nd0;     // round 0 of the chain, kept for the host's statistics: deferred ops,
  uint32_t spec_nt0;     //   growth tasks,
  uint32_t spec_failed;  //   growth tasks refused (budget of tasks / arena units): their rows stay as they are, their ops stay deferred
  uint64_t spec_gu0;     //   units the growths took
  uint32_t spec_nrebal0, spec_dirfull0;
  uint32_t spec_nkind0[4];
  // ---- the far join (k_home_list / k_far_plan): rows of >= 2^HOME_LG cells and their 1024-cell units, as of the last batch that ran it
  uint32_t n_big, n_units;
  uint32_t far_nd, pad_far;      // ops in the list the join was last built for (k_far_keys): the host sizes the next table from it
};
constexpr size_t CTL_ROUND_BYTES = 64;    // one aligned fill
static_assert(offsetof(Ctl, dir_used) == CTL_ROUND_BYTES, "the per-round part of Ctl is what ctl_reset_round zeroes");

// retired blocks, one stack of block addresses per size class (device arrays grown by the host)
struct FreeLists {
  uint32_t* list[N_CLASSES];
  uint32_t cap[N_CLASSES];
};

struct GrowTask {
  uint32_t dslot;        // directory slot index
  uint32_t old_lg;
  uint32_t old_base;
  uint32_t new_base;
  uint32_t count;        // non-empty cells moved (becomes `used`, src/smatrix.c:410)
  uint32_t chunk0;       // first 64-slot chunk of the old table in the flat chunk space
  uint32_t chunk0_new;   // same for the new table
  uint32_t dup;          // the old table holds one key twice (see grow_fixdup_one)
  // chunked tasks, clustered rows (k_grow_move_home): cells of the old table's LAST run are not taken for at-home cells when
  // the run goes on round the end of the table -- the wrapped cells come earlier in old slot order and may take their places
  uint32_t wrap_from;    // the smallest old home among the wrapped cells of the table's first run (k_grow_map); none: 2^32-1
  uint32_t wrap_seen;    // the same over ALL cells, as the first pass comes across them; smaller than wrap_from (a wrapped cell
                         // behind a hole, quirk Q1/Q3) sends the row to the serial redo
  uint32_t n_disp;       // clustered rows: cells the first pass did not store (k_grow_move_home) -- what k_grow_rest_lds places, in slices
  // (round 6) the NEW keys that wait for this doubling (k_prep's records, k_pend_group): growth puts them into the new table itself
  uint32_t pend_off;     // the row's bucket in the round's key buffer,
  uint32_t pend_cap;     //   its capacity (old size / 2: the new table has no room for more; 0: none) and
  uint32_t n_pend;       //   the distinct keys that asked for a place (may exceed the capacity: the rest waits for the retry)
};

// How a row is doubled: tables whose old cells and new slots fit in LDS are rebuilt there by one wave
// (kind 0), one 256-lane workgroup (kind 1) or one 1024-lane workgroup (kind 2); larger ones go through
// the chunked global-memory passes (kind 3).  LDS per task: 16 bytes per old cell.
constexpr uint32_t GROW_LG0 = 8;      // old size <= 256 cells : 4 KB per wave
constexpr uint32_t GROW_LG1 = 11;     // old size <= 2048 cells: 32 KB per workgroup
constexpr uint32_t GROW_LG2 = 13;     // old size <= 8192 cells: 128 KB, one workgroup per CU
constexpr uint32_t GROW_CHUNKED = 3;
__host__ __device__ inline uint32_t grow_kind(uint32_t old_lg) {
  return old_lg <= GROW_LG0 ? 0u : old_lg <= GROW_LG1 ? 1u : old_lg <= GROW_LG2 ? 2u : GROW_CHUNKED;
}

// Kernel bodies are device functions over a VIRTUAL grid (workgroup `bid` of `nb`) so that several of
// them can be composed into one launch; each has a thin __global__ wrapper with the launch's own grid.
// (A persistent kernel that ran all of them as phases between grid barriers was built, measured and
// dropped -- DESIGN.md "Measured and rejected".)
struct VGrid { uint32_t bid, nb; };
#define SMX_VG (VGrid{blockIdx.x, gridDim.x})
// control-block counters are read with agent-scope loads (they are written by atomics of earlier launches)
__device__ inline uint32_t aload(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint64_t aload(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bU; h ^= h >> 13; h *= 0xc2b2ae35U; h ^= h >> 16;
  return h;
}

__device__ inline uint64_t splitmix_at(uint64_t seed, uint64_t j) {
  uint64_t z = seed + (j + 1) * 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

__device__ inline uint64_t pack_cell(uint32_t key, uint32_t value) {
  return (uint64_t)key | ((uint64_t)value << 32);   // little-endian {key,value}
}
__device__ inline uint32_t cell_key(uint64_t c) { return (uint32_t)c; }
__device__ inline uint32_t cell_val(uint64_t c) { return (uint32_t)(c >> 32); }

__device__ inline uint64_t ld_relaxed(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline uint64_t* row_cells(uint8_t* arena, uint32_t base) {
  return reinterpret_cast<uint64_t*>(arena + (uint64_t)base * UNIT_BYTES);
}
__device__ inline SubCtr* row_subs(uint8_t* arena, uint32_t base, uint32_t lg) {
  return reinterpret_cast<SubCtr*>(arena + ((uint64_t)base + units_of_lg(lg)) * UNIT_BYTES);
}
// the at-home bitmap of a row of >= 2^HOME_LG cells (one 64-bit word per 64 cells)
__device__ inline unsigned long long* row_home(uint8_t* arena, uint32_t base, uint32_t lg) {
  return reinterpret_cast<unsigned long long*>(arena + ((uint64_t)base + units_of_lg(lg) + (lg >= BIG_LG ? SUB_UNITS : 0)) * UNIT_BYTES);
}
// the same from a table's cells and mask (what a LongProbe carries)
__device__ inline const unsigned long long* cells_home(const uint64_t* cells, uint32_t mask) {
  return reinterpret_cast<const unsigned long long*>(cells + (uint64_t)mask + 1u) + (mask + 1u >= (1u << BIG_LG) ? SUBS * 8u : 0u);
}
__device__ inline uint32_t subs_sum(const SubCtr* sc) {
  uint32_t t = 0;
  for (uint32_t k = 0; k < SUBS; k++) t += __hip_atomic_load(&sc[k].cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return t;
}

// Directory lookup on a STABLE directory (no creation in flight): plain 16-byte loads.
__device__ inline DirSlot* dir_find(DirSlot* dir, uint32_t dmask, uint32_t x, uint4* snap) {
  uint32_t h = fmix32(x) & dmask;
  for (;;) {
    uint4 s = *reinterpret_cast<const uint4*>(&dir[h]);   // {meta, x, base, used}
    if (!(s.x & META_USED)) return nullptr;
    if (s.y == x) { *snap = s; return &dir[h]; }
    h = (h + 1) & dmask;
  }
}
