// smx_kernels.hpp -- gfx950 device code of the (x,y)->uint32 path.
//
// HBM layout (DESIGN.md "Data layout"):
//   directory  : power-of-two array of 16-byte DirSlot {meta, x, base, used};
//                hash = murmur3 finaliser of x, linear probing, load <= 1/2.
//                Replaces the reference's x-directory smatrix_cmap_t
//                (src/smatrix.h:51-65, src/smatrix.c:598-741) -- its layout is
//                not observable through the API, so it is free to differ.
//   row tables : one block of 16*2^k 8-byte {key,value} cells per row in one
//                contiguous arena, addressed in 128-byte units.  A row table is
//                BIT-COMPATIBLE with the reference's smatrix_rmap_t data
//                (src/smatrix.h:35-49): identity hash `y % size`, linear
//                probing, empty == (0,0), growth x2 when `used > size/2` is
//                seen by an insert (src/smatrix.c:343-416).  Keeping it makes
//                rowlen/getrow order/file blocks identical to the reference.
//
// No locks: the reference's per-row spin RW lock (src/smatrix.c:843-889) is
// replaced by 64-bit CAS slot claims + 32-bit atomics on the value word, and
// structure changes (row creation, growth, directory growth) run in their own
// launches between rounds of the op kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace smx {

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
  uint32_t pad0;
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

// ---- op kernel ----------------------------------------------------------------
//
// One lane per op.  Restates smatrix_lookup + the per-op tail
// (src/smatrix.c:174-185 get, :225-256 set/incr/decr, :258-304 lookup,
//  :363-380 rmap_probe) on the HBM tables.  Writers that would have to create a
// row, or to insert into a row that stands at the reference's growth threshold
// (`used > size/2`, src/smatrix.c:346), are DEFERRED: the structure change is
// made by prep/grow between rounds, exactly where the reference makes it.
//
//   idx   : nullptr for round 0 (op i = thread i), else the deferred op list
//   cellp : unused here (set duplicates are resolved after the rounds, k_set_locate)
// one insert ticket from a sub-counter, or nullptr when its share of the room is used up
__device__ inline uint32_t* sub_ticket(SubCtr* sc) {
  const uint2 cq = *reinterpret_cast<const uint2*>(sc);      // {cnt, quota}; quota is stable in op kernels
  if (cq.x >= cq.y) return nullptr;
  if (atomicAdd(&sc->cnt, 1u) >= cq.y) { atomicSub(&sc->cnt, 1u); return nullptr; }
  return &sc->cnt;
}
// Own share exhausted: three more at stride SUBS/4.  With >= SUBS/4 tickets of room left some share on
// that stride still has one, so a nearly full row does not bounce its ops through re-partition rounds.
// The retry is on the slow path of both op kernels (PATIENT).  In the aggregating kernel it once cost
// 0.4 ms per 2^24-op batch -- 82 SGPRs, over the residency cliff -- and is affordable since the kernel
// is pinned to 80 SGPRs (it now compiles to 78 SGPRs / 58 VGPRs, still 8 waves per SIMD): fewer ops of
// big rows are deferred for nothing, 2.71 -> 2.68 ms per step (SMX_AGG_PATIENT).
__device__ inline uint32_t* sub_ticket_elsewhere(SubCtr* subs, uint32_t k0) {
  for (uint32_t a = 1; a < 4; a++)
    if (uint32_t* t = sub_ticket(subs + ((k0 + a * (SUBS / 4u)) & (SUBS - 1u)))) return t;
  if (uint32_t* t = sub_ticket(subs)) return t;  // the endgame pool (see subs_init)
  // Still nothing: look at EVERY share before giving up.  An op of a big row is then deferred only when the row
  // really stands at the reference's threshold, so prep grows it at once -- a row that was merely unevenly drained
  // used to cost a re-partition round, then the fill round, then the growth round (three rounds per batch for the
  // ~10 big rows that cross their threshold; now two).  The scan is 64 cached 8-byte loads; once it has come up
  // empty the row is marked so that the ops behind it do not repeat it.
  if (__hip_atomic_load(&subs[0].pad[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return nullptr;
  for (uint32_t a = 1; a < SUBS; a++)
    if (uint32_t* t = sub_ticket(subs + ((k0 + a) & (SUBS - 1u)))) return t;
  __hip_atomic_store(&subs[0].pad[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return nullptr;
}

// `want` insert tickets at once (k_insert_keys: one request per row and workgroup), starting at share k0 and going round
// all of them; returns how many it got.  Same invariant as sub_ticket: no share's count ever stays above its quota.
__device__ inline uint32_t sub_tickets_bulk(SubCtr* subs, uint32_t k0, uint32_t want) {
  if (__hip_atomic_load(&subs[0].pad[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 0;     // every share is used up
  uint32_t got = 0;
  for (uint32_t a = 0; a < SUBS && got < want; a++) {
    SubCtr* sc = subs + ((k0 + a) & (SUBS - 1u));
    const uint32_t cnt = __hip_atomic_load(&sc->cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), quota = sc->quota;   // (quota is stable in op kernels)
    if (cnt >= quota) continue;
    const uint32_t take = min(want - got, quota - cnt);
    const uint32_t old = atomicAdd(&sc->cnt, take);
    const uint32_t ok = old >= quota ? 0u : min(take, quota - old);
    if (ok < take) atomicSub(&sc->cnt, take - ok);
    got += ok;
  }
  if (got == 0) __hip_atomic_store(&subs[0].pad[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (a full turn came up empty)
  return got;
}

// ---- long probe sequences: the wave-cooperative window probe ---------------------------------------------
// Row tables keep the reference's identity hash (y % size, src/smatrix.c:366) because their bytes are the file format.
// With DENSE ids that hash clusters: low ids fill a contiguous run and every id that wraps onto the run walks to its
// end (displacements of 10^3..10^4).  One lane stepping cell by cell through such a run is a chain
// of thousands of dependent loads while the other 63 lanes of its wave idle.  So a lane probes PROBE_BUDGET cells on
// its own (scrambled ids never get that far: the longest sequence in the 100 M-cell config-2 tables is ~30) and then
// hands the probe to its WAVE: 64 lanes look at 64 consecutive cells per load (one coalesced 512-byte window, four
// windows in flight), two ballots find the first cell that ends the reference's probe -- key == Y or empty
// (src/smatrix.c:369-377) -- in probe order.
// Round 4: WHERE a far-from-home key sits is remembered.  Nearly all of a dense batch's long probes are HITS on keys that sat
// thousands of cells from home the batch before as well (770 000 of 2^24 ops, ~15 000 cells each: 7 ms of wave-per-op passes
// per step).  A direct-mapped table of {y, row base, slot} entries (the matrix allocates it when its tables turn out clustered)
// is consulted when a probe has used up its budget, and written when a wave-cooperative probe has ended on the key.  An entry
// is a HINT: it counts only if the cell it names holds y in the row's CURRENT block (a doubled row has a new base; a torn or
// overwritten entry fails the same test), and a key sits in one cell of its table -- with one exception, the twins of quirk
// Q1: a (0, v) cell whose value returns to 0 becomes an empty cell, a key behind it can then be inserted a second time in
// front, and the reference's probe from home finds THAT one.  So the first write op that leaves a (0, 0) cell behind switches
// the hints off for the matrix (`y0_zeroed`, sticky): they are an accelerator for dense-id streams, not a structure.
// Unit 0 of the arena (base 0 = "no block") holds the words the kernels need for this, so that no kernel signature grows.
struct ArenaHead {
  uint32_t y0_zeroed;     // a y == 0 write has left a (0, 0) cell (see above)
  uint32_t hint_mask;     // entries - 1 of the hint table; 0: none
  uint4* hints;
  // a table may hold one key TWICE (grow_fixdup_one): only after a probe chain was cut -- a (0, v) cell zeroed (y0_zeroed) or a
  // value-0 key dropped by the loader (quirk Q4).  While neither has happened the duplicate checks of growth are skipped.
  uint32_t twins;
  // the at-home bitmaps (HOME_LG) are kept up to date by the inserting kernels: probes may use them (clustered matrices)
  uint32_t home_on;
  // the far join of a clustered write batch (see "far join" below): valid only while far_on is set -- between the scan that
  // filled the table and the first structure change of the batch
  uint32_t far_on;
  uint32_t far_mask;                          // entries - 1 of the table
  uint4* far_tab;                             // {key lo = y, key hi = row block, slot, -}
  const unsigned long long* far_occ;          // occupancy words of the indexed rows, FAR_UNIT_WORDS per unit
  const uint32_t* far_zeros;                  // free cells per unit (at the scan): a probe steps over units without any
  uint32_t far_overflow;                      // a far key did not fit the table in this batch: no claimed inserts (k_far_keys, far_claim_insert)
  unsigned long long* dbg;                    // measurement runs only (SMATRIX_REST_DBG): event counters, see smatrix_close
};
static_assert(sizeof(ArenaHead) <= 128, "unit 0 of the arena");
#ifndef SMX_HINT_BUDGET
#define SMX_HINT_BUDGET 8
#endif
constexpr uint32_t HINT_BUDGET = SMX_HINT_BUDGET;        // cells a lane probes before it asks for a hint, when the matrix has a hint table
__device__ inline uint32_t hint_index(uint32_t base, uint32_t Y, uint32_t hmask) {
  return fmix32(base * 0x9E3779B1u ^ Y * 0x85EBCA77u) & hmask;
}
// the slot of key Y in the table at `cells` (block `base`, `mask` + 1 cells), or 2^32-1 when no valid hint exists
__device__ inline uint32_t hint_find(const uint8_t* arena, const uint64_t* cells, uint32_t mask, uint32_t Y) {
  const ArenaHead* ah = reinterpret_cast<const ArenaHead*>(arena);
  const uint32_t hmask = ah->hint_mask;
  if (hmask == 0 || Y == 0 || ah->y0_zeroed) return 0xFFFFFFFFu;
  const uint32_t base = (uint32_t)((reinterpret_cast<const uint8_t*>(cells) - arena) >> 7);
  const uint4 e = ah->hints[hint_index(base, Y, hmask)];
  if (e.x != Y || e.y != base || e.z > mask) return 0xFFFFFFFFu;
  return cell_key(cells[e.z]) == Y ? e.z : 0xFFFFFFFFu;
}
__device__ inline void hint_put(const uint8_t* arena, const uint64_t* cells, uint32_t Y, uint32_t pos) {
  const ArenaHead* ah = reinterpret_cast<const ArenaHead*>(arena);
  const uint32_t hmask = ah->hint_mask;
  if (hmask == 0 || Y == 0) return;
  const uint32_t base = (uint32_t)((reinterpret_cast<const uint8_t*>(cells) - arena) >> 7);
  ah->hints[hint_index(base, Y, hmask)] = uint4{Y, base, pos, 0u};
}

constexpr uint32_t PROBE_NONE = 0xFFFFFFFFu;
// the position of the r-th (0-based) set bit of w; r < popcount(w)
__device__ inline uint32_t select_bit(unsigned long long w, uint32_t r) {
  uint32_t pos = 0;
  uint32_t lo = (uint32_t)w, c = __popc(lo);
  if (r >= c) { r -= c; pos = 32; lo = (uint32_t)(w >> 32); }
  c = __popc(lo & 0xFFFFu);
  if (r >= c) { r -= c; pos += 16; lo >>= 16; }
  c = __popc(lo & 0xFFu);
  if (r >= c) { r -= c; pos += 8; lo >>= 8; }
  c = __popc(lo & 0xFu);
  if (r >= c) { r -= c; pos += 4; lo >>= 4; }
  c = __popc(lo & 0x3u);
  if (r >= c) { r -= c; pos += 2; lo >>= 2; }
  if (r >= (lo & 1u)) pos += 1;
  return pos;
}

// ---- the far join of a clustered write batch (round 5) -------------------------------------------------------------------------
// Dense ids leave a write batch with 2-4 x 10^5 ops whose probe outruns the lane's budget: keys that wrap onto a run of cells
// at home.  Walking each of them to its end -- even a wave per op, even stepping over at-home cells by the bitmaps -- costs
// 10^9 cells per batch (4.5-5.4 ms), most of it to learn that a NEW key is absent; prep then walks the deferred ones again.
// But all big rows together are only ~45 M cells.  So, per batch, on the quiescent tables between the folding kernel and the
// wave-per-op pass:
//   1. k_far_keys   the far keys of the deferred list enter a hash table F keyed {row block, y}            (~2 x 10^5 keys)
//   2. k_far_scan   ONE streaming pass over every row of >= 2^HOME_LG cells: each displaced cell looks its key up in F and
//                   leaves its slot there; the pass also writes an OCCUPANCY word per 64 cells into a scratch bitmap
//   3. the wave-per-op pass and prep ask F: slot known -> the op goes straight to its cell; key in F without a slot -> it was
//      ABSENT when the tables were scanned, so the probe goes on by the occupancy bitmap: a cell that was taken at the scan
//      holds another key (keys never leave their cells), only cells that were empty then are looked at -- they are empty, or
//      hold a key inserted since, possibly this very one.
// Nothing persists: the table and the bitmap are rebuilt from the tables themselves in every batch and dropped (far_on = 0)
// before the first row doubles, so there is no staleness to reason about; a row or key that did not fit (capacities are
// estimates from the batch before) is simply not in F and takes the wave-cooperative walk as before.  Off once a probe chain
// may have been cut (ArenaHead::twins: a key may then sit twice and the scan cannot know which cell a probe finds first).
constexpr uint32_t FAR_UNIT_LG = 9;                       // rows are scanned in units of 512 cells (8 occupancy words)
constexpr uint32_t FAR_ROW_LG = 9;                        // ... from 512 cells up (long probes start in rows of a few hundred cells)
constexpr uint32_t FAR_UNIT_WORDS = 1u << (FAR_UNIT_LG - 6);
constexpr uint32_t FAR_NOT_FOUND = 0xFFFFFFFFu;
__device__ inline uint32_t far_hash(uint32_t base, uint32_t Y) { return fmix32(base * 0x9E3779B1u + Y * 0x85EBCA77u + 0x27d4eb2fu); }
// the entry of {base, Y}, or nullptr (linear probing; a never-used entry ends the search)
__device__ inline uint4* far_entry(uint4* tab, uint32_t tmask, uint32_t base, uint32_t Y) {
  uint32_t e = far_hash(base, Y) & tmask;
  for (uint32_t guard = 0; guard <= tmask; guard++) {
    const uint2 k = *reinterpret_cast<const uint2*>(&tab[e]);
    if (k.x == Y && k.y == base) return &tab[e];
    if (k.x == 0 && k.y == 0) return nullptr;
    e = (e + 1) & tmask;
  }
  return nullptr;
}
// insert {base, Y} (slot not known yet); false when the table is too crowded around its home
__device__ inline bool far_insert(uint4* tab, uint32_t tmask, uint32_t base, uint32_t Y, uint32_t slot) {
  const unsigned long long key = ((unsigned long long)base << 32) | Y;
  uint32_t e = far_hash(base, Y) & tmask;
  for (uint32_t guard = 0; guard < 64; guard++) {
    unsigned long long prev = *reinterpret_cast<const unsigned long long*>(&tab[e]);
    if (prev == 0ull) prev = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[e]), 0ull, key);
    if (prev == 0ull || prev == key) { if (prev == 0ull || slot != FAR_NOT_FOUND) tab[e].z = slot; return true; }
    e = (e + 1) & tmask;
  }
  return false;
}
enum { FAR_NONE = 0, FAR_FOUND = 1, FAR_ABSENT = 2 };
struct FarHit { uint32_t state, slot; const unsigned long long* occ; uint4* entry; const uint32_t* zeros; };
// what the join knows about key Y of the table at `cells` (ArenaHead::far_on must have been checked)
__device__ inline FarHit far_find(const uint8_t* arena, const uint64_t* cells, uint32_t Y) {
  const ArenaHead* ah = reinterpret_cast<const ArenaHead*>(arena);
  const uint32_t base = (uint32_t)((reinterpret_cast<const uint8_t*>(cells) - arena) >> 7);
  if (Y == 0) return FarHit{FAR_NONE, 0u, nullptr, nullptr, nullptr};
  uint4* tab = ah->far_tab;
  const uint32_t tmask = ah->far_mask;
  // (both look-ups' first entries are asked for together: two dependent round trips less)
  uint32_t er = far_hash(base, 0u) & tmask, ek = far_hash(base, Y) & tmask;
  uint4 vr = tab[er], vk = tab[ek];
  const uint4* row = nullptr;
  uint4* e = nullptr;
  for (uint32_t guard = 0; guard <= tmask; guard++) {
    if (vr.x == 0u && vr.y == base) { row = &tab[er]; break; }
    if (vr.x == 0u && vr.y == 0u) break;
    er = (er + 1) & tmask;
    vr = tab[er];
  }
  if (!row) return FarHit{FAR_NONE, 0u, nullptr, nullptr, nullptr};
  const uint32_t first_unit = vr.z;
  for (uint32_t guard = 0; guard <= tmask; guard++) {
    if (vk.x == Y && vk.y == base) { e = &tab[ek]; break; }
    if (vk.x == 0u && vk.y == 0u) break;
    ek = (ek + 1) & tmask;
    vk = tab[ek];
  }
  if (!e) return FarHit{FAR_NONE, 0u, nullptr, nullptr, nullptr};
  if (vk.z != FAR_NOT_FOUND) return FarHit{FAR_FOUND, vk.z, nullptr, e, nullptr};
  return FarHit{FAR_ABSENT, 0u, ah->far_occ + (size_t)first_unit * FAR_UNIT_WORDS, e, ah->far_zeros + first_unit};
}
// ONE LANE's probe by the occupancy words of its row (the key was absent at the scan): the first cell at/after `pos`
// (cyclically) that was free then and is empty or holds Y now; PROBE_NONE after a full turn.  Units without a free cell are
// stepped over by their counts, so a key that wraps onto a 60 000-cell run costs ~120 loads, not 60 000 -- and 64 lanes do
// their walks side by side, where the wave-cooperative probe took one op's at a time.
// bits_only: the first cell that was free at the scan, whatever it holds now (the op that inserts by rank: nobody else inserts
// its key, so the cells that others have filled since the scan -- a hot front grows by thousands of cells during the pass, and
// looking at them one by one was 400 us for the slowest lane of a wave -- need not be looked at).
__device__ inline uint32_t far_walk(const uint64_t* cells, uint32_t mask, const unsigned long long* occ, const uint32_t* zeros, uint32_t Y, uint32_t pos,
                                    bool bits_only = false) {
  const uint32_t nwords = (mask + 1u) >> 6, wmask = nwords - 1u;
  uint32_t w = pos >> 6;
  unsigned long long z = ~occ[w] & (~0ull << (pos & 63u));
  for (uint32_t walked = 0; walked <= nwords + FAR_UNIT_WORDS;) {
    if (z) {
      const uint32_t p = (w << 6) + (uint32_t)__ffsll(z) - 1u;
      if (bits_only) return p;
      const uint64_t c = ld_relaxed(&cells[p]);
      if (c == 0 || cell_key(c) == Y) return p;
      z &= z - 1;                                      // taken since the scan by another key: on
      continue;
    }
    w = (w + 1) & wmask;
    walked++;
    if ((w & (FAR_UNIT_WORDS - 1u)) == 0) {            // a unit begins: those without a free cell are stepped over whole
      while (walked <= nwords + FAR_UNIT_WORDS && zeros[w >> (FAR_UNIT_LG - 6)] == 0) { w = (w + FAR_UNIT_WORDS) & wmask; walked += FAR_UNIT_WORDS; }
    }
    z = ~occ[w];
  }
  return PROBE_NONE;
}

// CLAIMED inserts of the far join.  The new far keys of a clustered row all walk to the same free cells -- the holes of their run,
// then the cells behind it -- and each insert must see the one before it: 2 000 new keys of one row were 2 000 dependent
// compare-and-swaps on the cell at the front, the pass's critical path.  With the join such a key is known to be absent and the
// free cells of its row are the clear bits of the occupancy words, so an insert CLAIMS its cell there first: the first clear bit
// at/after the key's own first free cell that it manages to set (one atomic OR per attempt; the word the OR returns is fresh, so
// a crowded front costs one atomic per 64 cells, not one per cell) names a cell nobody else will claim; the key is then stored
// with a compare-and-swap (a cell that a plain insert took in the meantime just sends the claimer on).  The table ends as SOME
// order of the reference's inserts would leave it (src/smatrix.c:343-380): every cell between a key's home and its own was
// taken at the scan or has its bit set -- claimed by an op that holds a ticket and stores its key there, or found taken.
// One op per key does this (the claim word of the key's entry in F); another op naming the same new key is deferred to the
// retry, which finds the key in place.  The words are scratch of this batch (k_far_scan rewrites them).
template <int OP>
__device__ inline uint32_t far_claim_insert(DirSlot* d, const uint4 s, uint8_t* arena, uint32_t Y, uint32_t V, uint32_t e0,
                                            unsigned long long* occ, const uint32_t* zeros, uint64_t* cells, uint32_t mask, bool* deferred,
                                            uint32_t* where) {
  const uint32_t lg = meta_lg(s.x);
  uint32_t* ticket = nullptr;                        // src/smatrix.c:346: insert only while used <= size/2 (as in apply_row)
  if (lg >= BIG_LG) {
    SubCtr* subs = row_subs(arena, s.z, lg);
    const uint32_t k0 = (blockIdx.x * 5u + threadIdx.x) & (SUBS - 1u);
    ticket = sub_ticket(subs + k0);
    if (!ticket) ticket = sub_ticket_elsewhere(subs, k0);
    if (!ticket) { *deferred = true; return 0; }
  } else {
    if (s.w > (mask + 1u) / 2u) { *deferred = true; return 0; }
    ticket = &d->used;
    if (atomicAdd(ticket, 1u) > (mask + 1u) / 2u) { atomicSub(ticket, 1u); *deferred = true; return 0; }
  }
  const uint32_t first = OP == OP_DECR ? 0u - V : V;
  const uint32_t nwords = (mask + 1u) >> 6, wmask = nwords - 1u;
  uint32_t w = e0 >> 6;
  unsigned long long z = ~__hip_atomic_load(&occ[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (~0ull << (e0 & 63u));
  for (uint32_t walked = 0; walked <= nwords + FAR_UNIT_WORDS;) {
    if (z) {
      const uint32_t b = (uint32_t)__ffsll(z) - 1u;
      const unsigned long long bit = 1ull << b;
      const unsigned long long old = atomicOr(&occ[w], bit);
      z &= ~(old | bit);                              // (what the word really held: the bits others have set since are not tried)
      if (old & bit) continue;                        // somebody else's
      const uint32_t pos = (w << 6) + b;
      const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&cells[pos]), 0ull, (unsigned long long)pack_cell(Y, first));
      if (prev == 0) { *where = pos; return first; }
      if (cell_key(prev) == Y) {                      // (not with one op per key; kept for safety: the cell is updated, the ticket goes back)
        atomicSub(ticket, 1u);
        uint32_t* vp = reinterpret_cast<uint32_t*>(&cells[pos]) + 1;
        *where = pos;
        return OP == OP_INCR ? atomicAdd(vp, V) + V : atomicSub(vp, V) - V;
      }
      continue;                                       // a plain insert took the cell meanwhile: the claim stands for it, on
    }
    w = (w + 1) & wmask;
    walked++;
    if ((w & (FAR_UNIT_WORDS - 1u)) == 0)             // units without a free cell at the scan are full for good
      while (walked <= nwords + FAR_UNIT_WORDS && zeros[w >> (FAR_UNIT_LG - 6)] == 0) { w = (w + FAR_UNIT_WORDS) & wmask; walked += FAR_UNIT_WORDS; }
    z = ~__hip_atomic_load(&occ[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  atomicSub(ticket, 1u);
  *deferred = true;
  return 0;
}

struct LongProbe {
  bool need;
  const uint64_t* cells;
  uint32_t mask, pos;          // continue at `pos`
};
#ifndef SMX_PROBE_BUDGET
#define SMX_PROBE_BUDGET 48
#endif
constexpr uint32_t PROBE_BUDGET = SMX_PROBE_BUDGET;

// Called by ALL lanes of a wave together (convergent).  Lanes with `need` get the first slot at/after `pos`
// (cyclically, at most one full turn) whose key is Y or that is empty; PROBE_NONE if the table has neither.
// The answer is a hint for tables that are being written (the caller re-examines the slot), exact for quiescent ones.
// use_home: the matrix keeps its at-home bitmaps up to date (ArenaHead::home_on).  After the first 256 cells the probe of a
// table of >= 2^HOME_LG cells then goes on BY THE BITMAP: 64 lanes load 64 mask words (4096 cells), the cells that are not
// at home -- the only ones that can hold Y or be empty -- are numbered across the wave (prefix sums of the popcounts) and
// examined 64 at a time in probe order: lane i finds the owner of candidate i by a binary search over the prefix sums
// (six shuffles) and its bit by a select in the owner's word.  A dense run costs one mask load per 4096 cells; a pile of
// displaced cells costs what it cost before.
// occ (per lane; the far join): the key was ABSENT when the row's occupancy words were written -- the whole probe goes by those
// words (a set bit: the cell was taken then, by another key), from `pos` on.
__device__ inline uint32_t coop_probe(bool need, const uint64_t* cells, uint32_t mask, uint32_t Y, uint32_t pos, bool use_home = false,
                                      const unsigned long long* occ = nullptr) {
  const uint32_t lane = __lane_id();
  uint64_t todo = __ballot(need);
  uint32_t result = PROBE_NONE;
  while (todo) {
    const int src = __ffsll((unsigned long long)todo) - 1;
    todo &= todo - 1;
    const uint64_t* cb = reinterpret_cast<const uint64_t*>(
        ((uint64_t)(uint32_t)__shfl((int)((uint64_t)cells >> 32), src) << 32) | (uint32_t)__shfl((int)(uint64_t)cells, src));
    const uint32_t mb = (uint32_t)__shfl((int)mask, src), yb = (uint32_t)__shfl((int)Y, src), pb = (uint32_t)__shfl((int)pos, src);
    uint32_t found = PROBE_NONE;
    const unsigned long long* ob = reinterpret_cast<const unsigned long long*>(
        ((uint64_t)(uint32_t)__shfl((int)((uint64_t)occ >> 32), src) << 32) | (uint32_t)__shfl((int)(uint64_t)occ, src));
    const bool by_occ = ob != nullptr;                                       // (wave-uniform)
    const bool by_bits = by_occ || (use_home && mb + 1u >= (1u << HOME_LG));
    for (uint64_t done = 0; !by_occ && done <= mb && found == PROBE_NONE; done += 256) {          // wave-uniform
      uint64_t c[4];
      bool ok[4];
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const uint64_t off = done + (uint32_t)w * 64u + lane;
        ok[w] = off <= mb;
        c[w] = ok[w] ? cb[(pb + (uint32_t)off) & mb] : ~0ull;
      }
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const uint64_t m = __ballot(ok[w] && (cell_key(c[w]) == yb || c[w] == 0));
        if (m && found == PROBE_NONE) found = (pb + (uint32_t)done + (uint32_t)w * 64u + (uint32_t)(__ffsll((unsigned long long)m) - 1)) & mb;
      }
      if (by_bits) break;                                                    // the rest of the walk goes by the bitmap
    }
    if (by_bits && found == PROBE_NONE) {
      const unsigned long long* hb = by_occ ? ob : cells_home(cb, mb);
      const uint32_t nwords = (mb + 1u) >> 6, wmask = nwords - 1u;
      const uint32_t start = by_occ ? pb : (pb + 256u) & mb;               // (cells [pb, pb + 256) have been looked at)
      const uint32_t w0 = start >> 6;
      // one full turn: the words w0 .. w0 + nwords (the first one from bit start & 63 on, and once more whole at the end)
      for (uint32_t wd = 0; wd <= nwords && found == PROBE_NONE; wd += 64) {        // wave-uniform
        const uint32_t wi = wd + lane;
        unsigned long long cand = 0;
        if (wi <= nwords) {
          cand = ~hb[(w0 + wi) & wmask];
          if (wi == 0) cand &= ~0ull << (start & 63u);
        }
        const uint32_t cnt = (uint32_t)__popcll(cand);
        uint32_t incl = cnt;                                                 // inclusive prefix sum over the lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
          if ((int)lane >= d) incl += o;
        }
        const uint32_t excl = incl - cnt, total = (uint32_t)__shfl((int)incl, 63);
        for (uint32_t base = 0; base < total && found == PROBE_NONE; base += 64) {  // wave-uniform
          const uint32_t g = base + lane;
          const bool have = g < total;
          uint32_t own = 0;                                                  // the largest lane whose exclusive prefix is <= g
#pragma unroll
          for (int st = 32; st >= 1; st >>= 1) {
            const uint32_t v = (uint32_t)__shfl((int)excl, (int)(own + st));
            if (v <= g) own += st;
          }
          const uint32_t e_o = (uint32_t)__shfl((int)excl, (int)own);
          const unsigned long long w_o = ((unsigned long long)(uint32_t)__shfl((int)(cand >> 32), (int)own) << 32) | (uint32_t)__shfl((int)(uint32_t)cand, (int)own);
          uint32_t slot = 0;
          uint64_t c = ~0ull;
          if (have) {
            slot = ((((w0 + wd + own) & wmask) << 6) | select_bit(w_o, g - e_o)) & mb;
            c = cb[slot];
          }
          const uint64_t m = __ballot(have && (cell_key(c) == yb || c == 0));
          if (m) found = (uint32_t)__shfl((int)slot, __ffsll((unsigned long long)m) - 1);
        }
      }
    }
    if ((int)lane == src) result = found;
  }
  return result;
}

// The per-op body on a row that exists: returns the op's result (new value for writers); *deferred is set when a
// structure change must happen first.  Probing starts at `pos` (Y & mask for a fresh op).
//   MODE 0  the lane probes to the end on its own (scalar ABI kernel, CF kernel)
//   MODE 1  after PROBE_BUDGET cells the probe is handed back in *lp (lane-per-op kernels: coop_probe, then re-enter
//           at the slot it found)
template <int OP, bool PATIENT = false, int MODE = 0>
__device__ inline uint32_t apply_row(DirSlot* d, const uint4 s, uint8_t* arena, uint32_t Y, uint32_t V, uint32_t pos,
                                     bool* deferred, LongProbe* lp, bool dbg_noticket = false, bool no_ret = false,
                                     bool exists_only = false, uint64_t* where_out = nullptr, uint32_t budget = PROBE_BUDGET,
                                     bool mark_home = false) {
  // where_out (writers, y != 0): the cell the op ended at, as an index into the arena's 8-byte cells (k_set_fold)
  // budget (MODE 1): cells the lane probes on its own
  // mark_home: a key inserted into its home cell gets its bit in the row's at-home bitmap (HOME_LG; clustered matrices)
  uint32_t result = 0;
  const uint32_t lg = meta_lg(s.x);
  const uint32_t mask = (1u << lg) - 1u;
  uint64_t* cells = row_cells(arena, s.z);
  // (meta does not change while op kernels run -- structure changes have their own launches -- so every lane that
  //  marks the row stores the same word)
  if (OP != OP_GET && !(s.x & META_DIRTY)) d->meta = s.x | META_DIRTY;
  if (OP == OP_GET) {
    // src/smatrix.c:369-377 then :299: hit iff the probed slot's key == y
    for (uint32_t step = 0; step <= mask; step++) {
      uint64_t c = cells[pos];
      if (cell_key(c) == Y) { result = cell_val(c); break; }
      if (c == 0) break;
      pos = (pos + 1) & mask;
      if (MODE && step >= budget) { *lp = LongProbe{true, cells, mask, pos}; return 0; }
    }
  } else if (Y != 0) {
    uint64_t c = cells[pos];
    for (uint32_t steps = 0;;) {
      if (cell_key(c) == Y) break;                       // found
      if (c == 0) {
        // insert: reserve a place in `used` first; the reference inserts only
        // while used <= size/2 (src/smatrix.c:346), otherwise it grows first
        // (the snapshot taken with the directory slot spares a row that already stands at the
        // threshold two contended atomics per op; a stale/low snapshot only costs the atomics)
        uint32_t* ticket = nullptr;
        if (dbg_noticket) {
          // measurement builds only (SMX_AGG_DBG 5): inserts without their `used` ticket
        } else if (lg >= BIG_LG) {
          // big row: take the ticket from one of the sub-counters (its quota is a share of the room)
          // (spread by lane as well: a handful of retried ops all sit in one wave and must not
          //  queue on the single share of one sub-counter)
          SubCtr* subs = row_subs(arena, s.z, lg);
          const uint32_t k0 = (blockIdx.x * 5u + threadIdx.x) & (SUBS - 1u);
          ticket = sub_ticket(subs + k0);
          if (PATIENT && !ticket) ticket = sub_ticket_elsewhere(subs, k0);
          if (!ticket) { *deferred = true; return 0; }
        } else {
          if (s.w > (mask + 1u) / 2u) { *deferred = true; return 0; }
          ticket = &d->used;
          if (atomicAdd(ticket, 1u) > (mask + 1u) / 2u) {
            atomicSub(ticket, 1u);
            *deferred = true;
            return 0;
          }
        }
        // claim the cell AND apply the op in one CAS: the reference's insert leaves {y,0} and the
        // caller then updates the value (:354-356 then :230/:241/:252) -- 0 op v, atomically here
        const uint32_t first = OP == OP_DECR ? 0u - V : V;
        uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&cells[pos]), 0ull,
                                  (unsigned long long)pack_cell(Y, first));
        if (prev == 0) {
          if (where_out) *where_out = ((uint64_t)s.z << 4) + pos;
          if (mark_home && lg >= HOME_LG && pos == (Y & mask)) atomicOr(&row_home(arena, s.z, lg)[pos >> 6], 1ull << (pos & 63u));
          return first;
        }
        if (!dbg_noticket) atomicSub(ticket, 1u);        // lost the slot: give the ticket back
        c = prev;
        continue;                                        // re-examine what is there now
      }
      if (++steps > mask) { *deferred = true; return 0; }  // no empty cell at all: let prep grow it
      pos = (pos + 1) & mask;
      if (MODE && steps > budget) { *lp = LongProbe{true, cells, mask, pos}; return 0; }
      c = cells[pos];
    }
    uint32_t* vp = reinterpret_cast<uint32_t*>(&cells[pos]) + 1;
    if (where_out) *where_out = ((uint64_t)s.z << 4) + pos;
    if (OP == OP_INCR) result = atomicAdd(vp, V) + V;      // :241, wraps mod 2^32
    else if (OP == OP_DECR) result = atomicSub(vp, V) - V; // :252
    else { result = V; if (!exists_only) atomicExch(vp, V); }   // :230 (duplicates: see k_set_locate; exists_only: k_set_fold's
                                                                //       winners -- the passes after the rounds write the value)
  } else {
    // y == 0 (quirk Q1, src/smatrix.c:297-303,:370-374): the first slot whose KEY
    // field is 0 -- the row's own (0,v) entry or the first empty slot -- is a hit;
    // nothing is inserted and `used` is not touched.  Done with a 64-bit CAS so
    // that a concurrent claim of that empty slot by another key cannot be hit.
    // (the guard counts CELLS walked, not attempts: a CAS lost to another writer of the same cell -- every item's total
    //  lives in column 0 in the CF example, and a hot item's is written from hundreds of workgroups at once -- is retried
    //  on the value it returned and must never end the loop: somebody else made progress)
    uint64_t c = ld_relaxed(&cells[pos]);
    for (uint32_t guard = 0; guard < 4u * (mask + 1u);) {
      if (OP != OP_SET && no_ret && cell_key(c) == 0) {
        // The caller does not want the op's result (d_out == NULL; the CF import): ONE 64-bit add of V << 32 to the whole
        // cell instead of the CAS loop.  A hot item's total is written from every tile of a batch, and each lost CAS is
        // another trip to the same address: 24 ms per 2^25-op batch of the session import against 3 ms like this.
        // The add lands in the value half whatever the key half is by then: key still 0 -> done (an empty cell has just
        // become the row's (0,v) entry, exactly quirk Q1); key != 0 -> another key claimed the cell in between, the add
        // is taken back and the walk goes on.  The table's final state is exact either way; only a RESULT read from
        // that other key's cell during the few hundred ns in between would be off -- which is why this path exists for
        // callers without results only.
        const unsigned long long dv = (unsigned long long)(OP == OP_INCR ? V : 0u - V) << 32;
        const uint64_t old = atomicAdd(reinterpret_cast<unsigned long long*>(&cells[pos]), dv);
        if (cell_key(old) == 0) {
          result = cell_val(old) + (OP == OP_INCR ? V : 0u - V);
          if (result == 0) { reinterpret_cast<ArenaHead*>(arena)->y0_zeroed = 1; reinterpret_cast<ArenaHead*>(arena)->twins = 1; }   // (a (0,0) cell is an empty cell: hints off)
          break;
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(&cells[pos]), 0ull - dv);
        c = old;
      }
      if (cell_key(c) == 0) {
        uint32_t nv = OP == OP_INCR ? cell_val(c) + V : OP == OP_DECR ? cell_val(c) - V : V;
        uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&cells[pos]),
                                  (unsigned long long)c, (unsigned long long)pack_cell(0, nv));
        if (prev == c) {
          result = nv;
          if (nv == 0 && c != 0) { reinterpret_cast<ArenaHead*>(arena)->y0_zeroed = 1; reinterpret_cast<ArenaHead*>(arena)->twins = 1; }   // (0, v) -> (0, 0): hints off (ArenaHead)
          break;
        }
        c = prev;
        continue;
      }
      guard++;
      pos = (pos + 1) & mask;
      c = ld_relaxed(&cells[pos]);
    }
  }
  return result;
}

// directory lookup + the per-op body (MODE as in apply_row)
template <int OP, bool PATIENT = false, int MODE = 0>
__device__ inline uint32_t apply_one(DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t X,
                                     uint32_t Y, uint32_t V, bool* deferred, LongProbe* lp = nullptr, bool dbg_noticket = false,
                                     bool no_ret = false, bool exists_only = false, uint64_t* where_out = nullptr,
                                     uint32_t budget = PROBE_BUDGET, bool mark_home = false) {
  uint4 s;
  DirSlot* d = dir_find(dir, dmask, X, &s);
  if (!d || s.z == 0) {
    *deferred = (OP != OP_GET);     // get on an absent row: 0, creates nothing (S1)
    return 0;
  }
  return apply_row<OP, PATIENT, MODE>(d, s, arena, Y, V, Y & ((1u << meta_lg(s.x)) - 1u), deferred, lp, dbg_noticket, no_ret,
                                      exists_only, where_out, budget, mark_home);
}

#ifndef SMX_APPLY_SGPRS
#define SMX_APPLY_SGPRS 80
#endif
// WPO (wave per op): lane 0 of every wave has an op, the other 63 only help with its long probe.  The retries of a
// clustered table (dense ids) are short lists in which nearly every op walks 10^3..10^5 cells; lane per op, a wave then
// takes its 64 long probes one after the other while most of the chip has nothing to do -- the second retry of a dense
// batch took 4 ms for 4 500 ops.
// HM: 0 the matrix has no hint table (ArenaHead; the instantiation every scrambled-id stream runs: nothing of it is compiled in),
//     1 it has one, 2 look (the wave-per-op kernel: clustered tables only)
// FAR: the pass in front of prep of a clustered write batch, with the batch's far join at hand (ArenaHead::far_on)
template <int OP, bool WPO = false, int HM = 0, bool FAR = false>
__device__ __forceinline__ void apply_body(
    VGrid g, Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n, const uint32_t* idx,
    const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
    const uint32_t* __restrict__ vs, uint32_t* __restrict__ out, uint32_t* defer, uint32_t st) {
  // st: distance between consecutive ops in xs/ys/vs, in words (1 = three arrays, 2 / 3 = one array of
  // {x,y} / {x,y,v} records with xs = rec, ys = rec + 1, vs = rec + 2: what the sharded exchange delivers)
  // n == 0xFFFFFFFF: the list's length is on the device (ctl->n_prev: the host has not read the previous round back)
  if (n == 0xFFFFFFFFu) n = aload(&ctl->n_prev);
  // (64-bit trip counter: with n > 2^31 ops and a grid that covers them all, t0 + the grid's size wraps around in 32 bits
  //  and the ops at the front would be applied a SECOND time -- round 3, found by the 2^31 + 2^27-op batch test)
  const uint64_t n_lanes = WPO ? (uint64_t)n * 64u : (uint64_t)n;
  const bool has_hints = HM == 1 || (HM == 2 && reinterpret_cast<const ArenaHead*>(arena)->hint_mask != 0);       // (wave-uniform)
  // (a wave per op with the join at hand: the lane looks at the home cell only -- nine dependent loads of the lane's own probe
  //  were half of such a pass's time; the wave's first window covers them in one load)
  const uint32_t budget = WPO && FAR ? 0u : has_hints ? HINT_BUDGET : PROBE_BUDGET;
  // (clustered matrices: long probes go by the rows' at-home bitmaps, and inserts keep them up to date -- HOME_LG)
  const bool use_home = HM != 0 && reinterpret_cast<const ArenaHead*>(arena)->home_on != 0;                          // (wave-uniform)
  for (uint64_t t064 = (uint64_t)g.bid * blockDim.x; t064 < n_lanes; t064 += (uint64_t)g.nb * blockDim.x) {    // block-uniform
    const uint64_t tl = t064 + threadIdx.x;
    const uint32_t t = WPO ? (uint32_t)(tl >> 6) : (uint32_t)tl;
    const bool live = tl < n_lanes && (!WPO || (tl & 63u) == 0);
    uint32_t j = 0, r = 0, Y = 0, V = 0;
    bool deferred = false;
    LongProbe lp{false, nullptr, 0, 0};
    uint4 s = {0, 0, 0, 0};
    DirSlot* d = nullptr;
    // (measurement runs, SMATRIX_REST_DBG: where a wave-per-op pass with the far join spends its cycles)
    unsigned long long* tdbg = WPO && FAR ? reinterpret_cast<const ArenaHead*>(arena)->dbg : nullptr;
    unsigned long long* hdbg = FAR && !WPO ? reinterpret_cast<const ArenaHead*>(arena)->dbg : nullptr;     // (a lane per op: how long a wave's trip takes, log2 buckets)
    const long long h0 = hdbg ? clock64() : 0;
    long long h_find = 0, h_walk = 0, h_ins = 0;
    long long tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0;
    if (tdbg) tc0 = clock64();
    if (live) {
      j = idx ? idx[t] : t;
      const size_t at = (size_t)j * st;
      Y = ys[at];
      V = OP != OP_GET ? vs[at] : 0u;
      d = dir_find(dir, dmask, xs[at], &s);
      if (!d || s.z == 0) deferred = (OP != OP_GET);      // get on an absent row: 0, creates nothing (S1)
      else r = apply_row<OP, true, 1>(d, s, arena, Y, V, Y & ((1u << meta_lg(s.x)) - 1u), &deferred, &lp, false, false, false, nullptr, budget, use_home);
    }
    // a probe that has used up its budget: is the key's cell remembered?  (ArenaHead: dense ids)
    // was_long: the evidence for "this table is clustered" -- a probe of more than PROBE_BUDGET cells, whatever the budget was
    bool was_long = lp.need && !has_hints;
    if (has_hints && lp.need) {
      const uint32_t p = hint_find(arena, lp.cells, lp.mask, Y);
      if (p != 0xFFFFFFFFu) {
        was_long = ((p - Y) & lp.mask) > PROBE_BUDGET;
        lp.need = false;
        r = apply_row<OP, true, 1>(d, s, arena, Y, V, p, &deferred, &lp);
      }
    }
    uint32_t p_coop = PROBE_NONE;                           // where the wave-cooperative probe ended
    // the far join of this batch (the wave-per-op pass in front of prep): the key's cell is known, or the key is known to have
    // been absent when the tables were scanned and the probe goes by the occupancy words
    const unsigned long long* occ = nullptr;
    const uint32_t* zer = nullptr;
    bool ranked = false;                                    // this op may insert its (absent) key by claiming a free cell in the occupancy words: far_claim_insert
    if (tdbg) tc1 = clock64();
    if (FAR && lp.need) {
      const ArenaHead* ah = reinterpret_cast<const ArenaHead*>(arena);
      if (ah->far_on && !ah->twins) {
        const long long hf0 = hdbg ? clock64() : 0;
        const FarHit fh = far_find(arena, lp.cells, Y);
        if (hdbg) h_find = clock64() - hf0;
        if (ah->dbg && !hdbg) atomicAdd(&ah->dbg[16 + fh.state], 1ull);
        if (fh.state == FAR_FOUND) {
          if (has_hints) was_long = ((fh.slot - Y) & lp.mask) > PROBE_BUDGET;
          lp.need = false;
          r = apply_row<OP, true, 1>(d, s, arena, Y, V, fh.slot, &deferred, &lp);
          p_coop = fh.slot;
        } else if (fh.state == FAR_ABSENT) {
          occ = fh.occ;
          if ((OP == OP_INCR || OP == OP_DECR) && !ah->far_overflow) {
            // one op per new key inserts it; another one naming the same key waits for the retry (it finds the key in place)
            if (atomicCAS(&fh.entry->w, 0u, 1u) == 0u) ranked = true;
            else { lp.need = false; deferred = true; was_long = true; }
          }
          if (!WPO && lp.need) {
            // a lane per op: the lane walks by the occupancy words itself (far_walk), all lanes of the wave side by side
            lp.need = false;
            was_long = true;
            const long long hw0 = hdbg ? clock64() : 0;
            const uint32_t p = far_walk(lp.cells, lp.mask, fh.occ, fh.zeros, Y, lp.pos, ranked);
            if (hdbg) { h_walk = clock64() - hw0; atomicAdd(&hdbg[40], 1ull); atomicAdd(&hdbg[41], (unsigned long long)((p - lp.pos) & lp.mask)); atomicMax(&hdbg[42], (unsigned long long)((p - lp.pos) & lp.mask));
                        atomicAdd(&hdbg[43 + min(meta_lg(s.x) / 4u, 5u)], 1ull); }
            const long long hi0 = hdbg ? clock64() : 0;
            if (p == PROBE_NONE) { deferred = (OP != OP_GET); r = 0; }
            else if (OP != OP_GET && OP != OP_SET && ranked) {
              // (the front = the first cell that was free at the scan: the same for every op that walks up to it, whenever it comes)
              uint32_t where = p;
              r = far_claim_insert<OP == OP_DECR ? OP_DECR : OP_INCR>(d, s, arena, Y, V, p, const_cast<unsigned long long*>(fh.occ), fh.zeros, const_cast<uint64_t*>(lp.cells), lp.mask, &deferred, &where);
              p_coop = where;
            } else {
              r = apply_row<OP, true, 1>(d, s, arena, Y, V, p, &deferred, &lp);
              p_coop = p;
            }
            occ = nullptr;
            ranked = false;                                 // (whatever is left of this op walks the old way)
            if (hdbg) h_ins = clock64() - hi0;
          }
          zer = fh.zeros;
        }
      }
    }
    if (tdbg) tc2 = clock64();
    long long t_coop = 0;
    while (__any(lp.need)) {                              // wave-uniform: long probes are finished by the whole wave
      const long long ta = tdbg ? clock64() : 0;
      const uint32_t p = coop_probe(lp.need, lp.cells, lp.mask, Y, lp.pos, use_home, occ);
      if (tdbg) t_coop += clock64() - ta;
      if (lp.need) {
        lp.need = false;
        if (p == PROBE_NONE) { deferred = (OP != OP_GET); r = 0; was_long = true; }   // neither the key nor an empty cell: prep grows the row
        else {
          if (has_hints) was_long = ((p - Y) & lp.mask) > PROBE_BUDGET;
          if (OP != OP_GET && OP != OP_SET && ranked && ld_relaxed(&lp.cells[p]) == 0) {
            // the first cell that was free at the scan is still free: the key goes in by rank from this front
            ranked = false;
            uint32_t where = p;
            r = far_claim_insert<OP == OP_DECR ? OP_DECR : OP_INCR>(d, s, arena, Y, V, p, const_cast<unsigned long long*>(occ), zer, const_cast<uint64_t*>(lp.cells), lp.mask, &deferred, &where);
            p_coop = where;
          } else {
            r = apply_row<OP, true, 1>(d, s, arena, Y, V, p, &deferred, &lp);
            p_coop = p;
          }
        }
      }
    }
    if (hdbg) {
      // per wave: the longest lane of each phase, and the whole trip up to here
      long long mf = h_find, mw = h_walk, mi = h_ins;
#pragma unroll
      for (int dd = 32; dd >= 1; dd >>= 1) {
        mf = max(mf, (long long)__shfl_xor((int)mf, dd)); mw = max(mw, (long long)__shfl_xor((int)mw, dd)); mi = max(mi, (long long)__shfl_xor((int)mi, dd));
      }
      if (__lane_id() == 0) {
        const long long tot = clock64() - h0;
        atomicAdd(&hdbg[20], (unsigned long long)mf); atomicAdd(&hdbg[21], (unsigned long long)mw); atomicAdd(&hdbg[22], (unsigned long long)mi);
        atomicAdd(&hdbg[23], (unsigned long long)tot); atomicAdd(&hdbg[24], 1ull);
        atomicMax(&hdbg[25], (unsigned long long)mw); atomicMax(&hdbg[26], (unsigned long long)mi); atomicMax(&hdbg[27], (unsigned long long)tot);
      }
    }
    if (tdbg) {
      tc3 = clock64();
      if (__lane_id() == 0 && live) {
        atomicAdd(&tdbg[20], (unsigned long long)(tc1 - tc0)); atomicAdd(&tdbg[21], (unsigned long long)(tc2 - tc1));
        atomicAdd(&tdbg[22], (unsigned long long)t_coop); atomicAdd(&tdbg[23], (unsigned long long)(tc3 - tc2 - t_coop)); atomicAdd(&tdbg[24], 1ull);
      }
    }
    if (OP != OP_GET && !WPO) {                           // (the host's evidence for "this table is clustered")
      const uint64_t lm = __ballot(was_long);
      if (lm && __lane_id() == 0) atomicAdd(&ctl->n_long_ops, (uint32_t)__popcll(lm));
    }
    // (a wave per op: one op in 64 is looked at, and counts for 64 -- the evidence for "not clustered any more")
    if (OP != OP_GET && WPO && live && (t & 63u) == 0 && was_long) atomicAdd(&ctl->n_long_ops, 64u);
    // the key sits there (found, or just inserted): remembered for the next op that names it (ArenaHead)
    if (has_hints && p_coop != PROBE_NONE && !deferred && cell_key(row_cells(arena, s.z)[p_coop]) == Y) hint_put(arena, row_cells(arena, s.z), Y, p_coop);
    if (live && !deferred) out[j] = r;
    if (OP != OP_GET) {
      // one list reservation per WORKGROUP: every atomic instruction on this one word queues at the
      // memory side (~34 ns each), and a retry round has thousands of waves with a deferred op
      __shared__ uint32_t l_n, l_base;
      if (!__syncthreads_or(deferred)) continue;
      if (threadIdx.x == 0) l_n = 0;
      __syncthreads();
      const uint64_t m = __ballot(deferred);
      const uint32_t lane = __lane_id();
      uint32_t wbase = 0;
      if (m && lane == 0) wbase = atomicAdd(&l_n, (uint32_t)__popcll(m));
      wbase = __shfl(wbase, 0);
      __syncthreads();
      if (threadIdx.x == 0) l_base = atomicAdd(&ctl->n_defer, l_n);
      __syncthreads();
      if (deferred) defer[l_base + wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = j;
    }
  }
}

// (pinned to the 80-SGPR budget like k_apply_agg: the writers compiled to 97-100 SGPRs, over the residency cliff)
template <int OP, bool HINTS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(SMX_APPLY_SGPRS))) void k_apply(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n, const uint32_t* idx,
    const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
    const uint32_t* __restrict__ vs, uint32_t* __restrict__ out, uint32_t* defer, uint32_t st) {
  apply_body<OP, false, HINTS ? 1 : 0>(SMX_VG, ctl, dir, dmask, arena, n, idx, xs, ys, vs, out, defer, st);
}

// The keys of the listed ops -- DISTINCT keys (k_dedup_keys) -- inserted with value 0 where they do not exist (an incr by 0:
// src/smatrix.c:236-243 on an absent key inserts {y, 0} and adds 0).  The list of a cold start names a few thousand rows
// with up to 10^5 pending keys each, of which a row takes size/2 per round: with one `used` ticket attempt per key
// (apply_row) the hottest row's word took 3 x 10^5 refused add/sub pairs per round, 1.2 ms per launch.  Here the lanes of
// a workgroup that stand at an empty cell of the same row ask for their tickets TOGETHER: one add (and one give-back
// of what was refused) per row and workgroup; a lane with a ticket keeps it until its key is in (nobody else inserts
// that key).  Big rows (sub-counter quotas) and long probe sequences take the general path.
constexpr uint32_t INS_THREADS = 1024;
// Round 4: the keys travel PACKED -- n 64-bit keys (x << 32 | y) in `kin`, the ones that stay deferred written to `kout` the same
// way (one reservation per workgroup, as before).  A round used to read an index list and gather x and y of every listed op from
// the batch's arrays (two random 4-byte loads per key and round out of 134 MB, again in k_prep); now every round streams its input.
__global__ __launch_bounds__(INS_THREADS) void k_insert_keys(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n, const unsigned long long* __restrict__ kin,
    unsigned long long* __restrict__ kout) {
  __shared__ uint32_t l_row[2 * INS_THREADS], l_cnt[2 * INS_THREADS], l_grant[2 * INS_THREADS];
  __shared__ uint32_t l_n, l_base;
  for (uint64_t t064 = (uint64_t)blockIdx.x * INS_THREADS; t064 < n; t064 += (uint64_t)gridDim.x * INS_THREADS) {   // block-uniform
    const uint32_t t = (uint32_t)t064 + threadIdx.x;
    const bool live = t < n;
    for (uint32_t i = threadIdx.x; i < 2 * INS_THREADS; i += INS_THREADS) { l_row[i] = 0xFFFFFFFFu; l_cnt[i] = 0; }
    if (threadIdx.x == 0) l_n = 0;
    unsigned long long key = 0;
    uint32_t Y = 0, pos = 0, mask = 0, e = 0, rank = 0;
    bool deferred = false, need = false, general = false;
    uint4 s = {0, 0, 0, 0};
    DirSlot* d = nullptr;
    uint64_t* cells = nullptr;
    LongProbe lp{false, nullptr, 0, 0};
    const bool use_home = reinterpret_cast<const ArenaHead*>(arena)->home_on != 0;          // (uniform; HOME_LG)
    if (live) {
      key = kin[t];
      Y = (uint32_t)key;
      d = dir_find(dir, dmask, (uint32_t)(key >> 32), &s);
      if (!d || s.z == 0) deferred = true;                       // the row does not exist (yet): prep creates it
      else if (Y == 0) general = true;
      else if (meta_lg(s.x) < BIG_LG ? s.w > (1u << meta_lg(s.x)) / 2u
                                     : __hip_atomic_load(&row_subs(arena, s.z, meta_lg(s.x))[0].pad[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        // (round 4) the row stands at the reference's threshold (src/smatrix.c:346) -- the snapshot's count, or a big row's
        // "every share is used up" mark: the key is absent (listed keys are), so it is deferred WITHOUT walking to its empty
        // cell first.  Three quarters of a cold round's keys belong to rows that are waiting for their doubling.
        deferred = true;
      } else {
        mask = (1u << meta_lg(s.x)) - 1u;
        cells = row_cells(arena, s.z);
        if (!(s.x & META_DIRTY)) d->meta = s.x | META_DIRTY;
        pos = Y & mask;
        for (uint32_t steps = 0;; steps++) {
          const uint64_t c = cells[pos];
          if (cell_key(c) == Y) break;                           // it exists: nothing to do
          if (c == 0) { need = true; break; }
          if (steps > PROBE_BUDGET) { general = true; break; }
          pos = (pos + 1) & mask;
        }
        if (need && meta_lg(s.x) < BIG_LG && s.w > (mask + 1u) / 2u) { need = false; deferred = true; }     // (the snapshot already shows the row full)
      }
    }
    __syncthreads();
    // the tickets of this workgroup, one request per row
    const uint32_t h = (uint32_t)(d - dir);
    bool owner = false;
    if (need) {
      e = (h * 0x9E3779B1u) >> 21;                               // 11 bits
      for (;;) {
        const uint32_t prev = atomicCAS(&l_row[e], 0xFFFFFFFFu, h);
        if (prev == 0xFFFFFFFFu) { owner = true; break; }
        if (prev == h) break;
        e = (e + 1) & (2 * INS_THREADS - 1);
      }
      rank = atomicAdd(&l_cnt[e], 1u);
    }
    __syncthreads();
    if (owner) {
      // a ticket is good while the count before it is <= size/2 (src/smatrix.c:346).  A coherent look first: once the row is
      // full -- after the first few workgroups of a launch -- nobody has to add and take back any more
      const uint32_t limit = (mask + 1u) / 2u, now = aload(&d->used);
      uint32_t ok = 0;
      if (meta_lg(s.x) >= BIG_LG) {
        // big row: the room is shared out over its sub-counters (see SubCtr)
        ok = sub_tickets_bulk(row_subs(arena, s.z, meta_lg(s.x)), (blockIdx.x * 5u + (e & 7u)) & (SUBS - 1u), l_cnt[e]);
      } else if (now <= limit) {
        const uint32_t want = min(l_cnt[e], limit + 1u - now);
        const uint32_t base = atomicAdd(&d->used, want);
        ok = base > limit ? 0u : min(want, limit + 1u - base);
        if (ok < want) atomicSub(&d->used, want - ok);
      }
      l_grant[e] = ok;
    }
    __syncthreads();
    if (need) {
      if (rank >= l_grant[e]) deferred = true;
      else {
        // the ticket is this key's until it is in: a cell lost to another key only moves the walk on
        for (uint32_t guard = 0; guard <= mask; guard++) {
          const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&cells[pos]), 0ull, (unsigned long long)pack_cell(Y, 0u));
          if (prev == 0) {
            if (use_home && mask + 1u >= (1u << HOME_LG) && pos == (Y & mask))
              atomicOr(&row_home(arena, s.z, meta_lg(s.x))[pos >> 6], 1ull << (pos & 63u));
            break;
          }
          if (cell_key(prev) == Y) {                             // (not with distinct keys.  Big rows: `used` is the folded part of the
            atomicSub(&d->used, 1u);                             //  count, rowlen = used + sum(cnt) stays exact this way too)
            break;
          }
          do { pos = (pos + 1) & mask; } while (ld_relaxed(&cells[pos]) != 0 && cell_key(ld_relaxed(&cells[pos])) != Y && ++guard <= mask);
        }
      }
    }
    // the general path (big rows: sub-counter quotas; long probe sequences: the wave-cooperative probe)
    uint32_t r = 0;
    if (general) r = apply_row<OP_INCR, true, 1>(d, s, arena, Y, 0u, Y & ((1u << meta_lg(s.x)) - 1u), &deferred, &lp, false, false, false, nullptr, PROBE_BUDGET, use_home);
    while (__any(lp.need)) {
      const uint32_t p = coop_probe(lp.need, lp.cells, lp.mask, Y, lp.pos, use_home);
      if (lp.need) {
        lp.need = false;
        if (p == PROBE_NONE) deferred = true;
        else r = apply_row<OP_INCR, true, 1>(d, s, arena, Y, 0u, p, &deferred, &lp);
      }
    }
    (void)r;
    // what stays deferred: one list reservation per workgroup
    const uint64_t dm = __ballot(deferred);
    uint32_t wbase = 0;
    if (dm && __lane_id() == 0) wbase = atomicAdd(&l_n, (uint32_t)__popcll(dm));
    wbase = __shfl(wbase, 0);
    __syncthreads();
    if (threadIdx.x == 0 && l_n) l_base = atomicAdd(&ctl->n_defer, l_n);
    __syncthreads();
    if (deferred) kout[l_base + wbase + (uint32_t)__popcll(dm & ((1ull << __lane_id()) - 1ull))] = key;
    __syncthreads();                                             // the LDS tables are reused by the next trip
  }
}

// One representative op per distinct key (x, y != 0) among the listed ops: a scratch hash set of 64-bit keys (zeroed by the
// caller, >= 2 slots per op), the first op to claim a key goes to `reps`.  Representatives are collected in LDS and leave
// with ONE reservation per workgroup and DEDUP_TRIPS x 1024 ops (a reservation per wave queued 10^5 atomics on one word).
constexpr uint32_t DEDUP_THREADS = 1024, DEDUP_TRIPS = 8;
__global__ __launch_bounds__(DEDUP_THREADS) void k_dedup_keys(uint32_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ xs,
                                                             const uint32_t* __restrict__ ys, uint32_t st, unsigned long long* set,
                                                             uint64_t set_mask, unsigned long long* reps, uint32_t* n_reps) {
  __shared__ unsigned long long l_rep[DEDUP_THREADS * DEDUP_TRIPS];     // (round 4: the distinct keys themselves, x << 32 | y)
  __shared__ uint32_t l_n, l_base;
  for (uint64_t b0 = (uint64_t)blockIdx.x * DEDUP_THREADS * DEDUP_TRIPS; b0 < n; b0 += (uint64_t)gridDim.x * DEDUP_THREADS * DEDUP_TRIPS) {
    if (threadIdx.x == 0) l_n = 0;
    __syncthreads();
    for (uint32_t k = 0; k < DEDUP_TRIPS; k++) {
      const uint64_t t = b0 + (uint64_t)k * DEDUP_THREADS + threadIdx.x;
      bool won = false;
      unsigned long long key = 0;
      if (t < n) {
        const uint32_t j = idx[t];
        const uint32_t X = xs[(size_t)j * st], Y = ys[(size_t)j * st];
        if (Y != 0) {                                     // (y == 0 never inserts: quirk Q1)
          key = ((unsigned long long)X << 32) | Y;
          uint64_t h = splitmix_at(0x5eedull, key) & set_mask;
          for (;;) {
            // (a plain look first: a hot key has 10^5 duplicates, and as many CAS on its slot queue at the memory side --
            //  the kernel took 3 ms; a stale line can only show an empty slot, which the CAS then settles)
            unsigned long long prev = set[h];
            if (prev == 0ull) prev = atomicCAS(&set[h], 0ull, key);
            if (prev == 0ull) { won = true; break; }
            if (prev == key) break;
            h = (h + 1) & set_mask;
          }
        }
      }
      const uint64_t wm = __ballot(won);
      uint32_t wb = 0;
      if (wm && __lane_id() == 0) wb = atomicAdd(&l_n, (uint32_t)__popcll(wm));
      wb = __shfl(wb, 0);
      if (won) l_rep[wb + (uint32_t)__popcll(wm & ((1ull << __lane_id()) - 1ull))] = key;
    }
    __syncthreads();
    if (threadIdx.x == 0 && l_n) l_base = atomicAdd(n_reps, l_n);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < l_n; i += DEDUP_THREADS) reps[l_base + i] = l_rep[i];
    __syncthreads();
  }
}

// The pass in front of prep when the batch's far join is there: a LANE per op again.  With the join a far op is a table look-up
// and, for a new key, a look at a few occupancy words -- no walk worth a whole wave (k_apply_wpo: 26 us per op and wave).
template <int OP>
__global__ __launch_bounds__(256) void k_apply_far(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n, const uint32_t* idx,
    const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
    const uint32_t* __restrict__ vs, uint32_t* __restrict__ out, uint32_t* defer, uint32_t st) {
  apply_body<OP, false, 1, true>(SMX_VG, ctl, dir, dmask, arena, n, idx, xs, ys, vs, out, defer, st);
}

// ... and the same a wave per op (measured faster: 2.6 against 3.5-4.6 ms per dense-id batch -- a wave with 64 far ops still
// takes their cooperative walks one after the other)
template <int OP>
__global__ __launch_bounds__(256) void k_apply_wpo_far(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n, const uint32_t* idx,
    const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
    const uint32_t* __restrict__ vs, uint32_t* __restrict__ out, uint32_t* defer, uint32_t st) {
  apply_body<OP, true, 2, true>(SMX_VG, ctl, dir, dmask, arena, n, idx, xs, ys, vs, out, defer, st);
}

template <int OP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(SMX_APPLY_SGPRS))) void k_apply_wpo(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n, const uint32_t* idx,
    const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
    const uint32_t* __restrict__ vs, uint32_t* __restrict__ out, uint32_t* defer, uint32_t st) {
  apply_body<OP, true, 2>(SMX_VG, ctl, dir, dmask, arena, n, idx, xs, ys, vs, out, defer, st);
}

// ---- the scalar ABI's fast path: ONE op, arguments by value, result straight into pinned host memory
// res[0] = value, res[1] = 1 if a structure change is needed first (the host then takes the round loop)
template <int OP>
__global__ void k_scalar(DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t X, uint32_t Y, uint32_t V,
                         volatile uint32_t* res) {
  bool deferred = false;
  uint32_t r = apply_one<OP>(dir, dmask, arena, X, Y, V, &deferred);
  res[0] = r;
  res[1] = deferred ? 1u : 0u;
}

// ---- op kernel with in-tile aggregation (incr / decr) -----------------------------
//
// Under Zipf(1.1) x Zipf(1.1) 1.5 % of all ops hit ONE cell and a few dozen cells take a
// quarter of the stream; their atomics serialise at the memory side (~34 ns each, measured:
// profiles/r01_*), which alone set the un-aggregated kernel's time.  Here a workgroup first
// folds its tile of AGG_TILE ops in an LDS hash table keyed by (x,y):
// (tile = 1024 lanes x 2 ops, 52 KB of LDS, two workgroups per CU)
//   phase 1  every op CAS-claims/joins its key's LDS slot and atomically adds its value to the
//            slot's sum; the value the sum had before is the op's prefix inside the tile
//   phase 2  one lane per DISTINCT key applies the tile's total with the per-op body above
//            (directory lookup, probe, claim, ONE global atomic) and leaves the cell's old value
//   phase 3  every op returns  old + prefix + v  (incr)  /  old - prefix - v  (decr)
// -- the values a serial execution of the tile's ops in LDS-arrival order returns, i.e. a legal
// serialisation.  The all-ones key (the LDS table's empty marker) takes the per-op body.
#ifndef SMX_AGG_PATIENT
#define SMX_AGG_PATIENT true
#endif
#ifndef SMX_AGG_OPT
#define SMX_AGG_OPT 2
#endif
#ifndef SMX_AGG_THREADS
#define SMX_AGG_THREADS 1024     /* measured on config 2: 256x4 1.92 ms, 512x4 1.64, 1024x4 1.60, 1024x2 1.55 */
#endif
constexpr uint32_t AGG_OPT = SMX_AGG_OPT;          // ops per lane
constexpr uint32_t AGG_THREADS = SMX_AGG_THREADS;  // lanes per workgroup
constexpr uint32_t AGG_TILE = AGG_THREADS * AGG_OPT;   // ops per workgroup
constexpr uint32_t AGG_SLOTS = 2 * AGG_TILE;       // LDS hash slots (load <= 1/2)

// SGPR budget: gfx950 admits 8 waves per SIMD only up to 80 SGPRs (MI355X_MICROARCH.md, residency);
// at 82 a CU holds ONE 1024-lane workgroup instead of two and the kernel takes 1.84 ms instead of 1.49.
#ifndef SMX_AGG_SGPRS
#define SMX_AGG_SGPRS 80
#endif
// CLU: the instantiation for clustered tables with a hint table (ArenaHead) -- the slow path asks for the hint after HINT_BUDGET
// cells instead of walking PROBE_BUDGET dependent loads first (a tile waits for its slowest lane)
template <int OP, uint32_t ST = 1, bool RET = true, bool CLU = false>     // ST: op stride in words, compile-time here (the kernel has no SGPR to spare); RET: results wanted
__global__ __launch_bounds__(AGG_THREADS) __attribute__((amdgpu_num_sgpr(SMX_AGG_SGPRS))) void k_apply_agg(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n, const uint32_t* idx,
    const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
    const uint32_t* __restrict__ vs, uint32_t* __restrict__ out, uint32_t* defer) {
  static_assert(OP == OP_INCR || OP == OP_DECR, "aggregation is for commutative ops");
  __shared__ uint64_t l_key[AGG_SLOTS];     // (x | y<<32); after phase 2 the low word = status
  __shared__ uint32_t l_sum[AGG_SLOTS];     // running sum; after phase 2 the cell's old value
  __shared__ uint16_t l_list[AGG_TILE];     // occupied slots, compact
  __shared__ uint32_t l_n;
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < AGG_SLOTS; i += AGG_THREADS) { l_key[i] = ~0ull; l_sum[i] = 0; }
  if (tid == 0) l_n = 0;
  __syncthreads();

  const uint32_t tile0 = blockIdx.x * AGG_TILE;
  uint32_t j[AGG_OPT], V[AGG_OPT], pre[AGG_OPT], slot[AGG_OPT];
  // phase 1
#pragma unroll
  for (uint32_t k = 0; k < AGG_OPT; k++) {
    const uint32_t t = tile0 + k * AGG_THREADS + tid;
    slot[k] = ~0u;                 // ~0: no op; ~0-1: per-op path
    if (t >= n) continue;
    j[k] = idx ? idx[t] : t;
    const uint32_t X = xs[(size_t)j[k] * ST], Y = ys[(size_t)j[k] * ST];
    V[k] = vs[(size_t)j[k] * ST];
    const uint64_t key = (uint64_t)X | ((uint64_t)Y << 32);
    if (key == ~0ull) { slot[k] = ~0u - 1; continue; }      // the LDS table's empty marker: per-op path
    uint32_t h = (X * 0x9E3779B1u) ^ (Y * 0x85EBCA77u);
    h = (h ^ (h >> 15)) & (AGG_SLOTS - 1);
    for (;;) {
      uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&l_key[h]), ~0ull,
                                (unsigned long long)key);
      if (prev == ~0ull) l_list[atomicAdd(&l_n, 1u)] = (uint16_t)h;     // first of its key
      if (prev == ~0ull || prev == key) break;
      h = (h + 1) & (AGG_SLOTS - 1);
    }
    pre[k] = atomicAdd(&l_sum[h], V[k]);
    slot[k] = h;
  }
  __syncthreads();
  // phase 2: a lane owns up to AGG_OPT distinct keys.  The common case -- directory hit on the
  // first probe, cell hit on the first probe -- is software-pipelined over the lane's keys (all
  // directory loads in flight, then all cell loads, then all atomics) so that the three dependent
  // memory round trips of one key overlap with those of the others; anything else (collision,
  // insert, missing row) falls back to the generic per-op body.
  const uint32_t nd = l_n;
#ifdef SMX_AGG_DBG
  // measurement builds only (tools/probe/agg_phases.sh): once the host has set ctl->pad1, part of phase 2 is left out
  // so that its share of the kernel's time can be read off (the tables are wrong afterwards: timing runs only)
  const uint32_t dbg = aload(&ctl->pad1) ? SMX_AGG_DBG : 0;
#else
  constexpr uint32_t dbg = 0;
#endif
  {
    uint32_t hh[AGG_OPT], tot[AGG_OPT], old[AGG_OPT];
    uint64_t kk[AGG_OPT];
    uint4 ds[AGG_OPT];
    uint64_t cc[AGG_OPT];
    uint64_t* cp[AGG_OPT];
    uint32_t fast = 0, have = 0;
#pragma unroll
    for (uint32_t q = 0; q < AGG_OPT; q++) {
      const uint32_t i = tid + q * AGG_THREADS;
      if (i < nd) {
        have |= 1u << q;
        hh[q] = l_list[i];
        kk[q] = l_key[hh[q]];
        tot[q] = l_sum[hh[q]];
        if (dbg != 2) ds[q] = *reinterpret_cast<const uint4*>(&dir[fmix32((uint32_t)kk[q]) & dmask]);
      }
    }
    if (dbg == 2 || dbg == 3) {                  // 2: no global access at all in phase 2; 3: directory loads only
#pragma unroll
      for (uint32_t q = 0; q < AGG_OPT; q++) {
        if (!(have & (1u << q))) continue;
        l_sum[hh[q]] = dbg == 3 ? ds[q].w : 0u;
        reinterpret_cast<uint32_t*>(&l_key[hh[q]])[0] = 0u;
      }
      have = 0;
    }
#pragma unroll
    for (uint32_t q = 0; q < AGG_OPT; q++) {
      if (!(have & (1u << q))) continue;
      // (y == 0 is folded like any key -- the CF example keeps every item's total there, examples/cf_recommender.c:38 --
      //  but never takes this pipelined path: its cell is found and updated by the quirk branch of the per-op body,
      //  with a 64-bit CAS, because "the first slot whose key field is 0" may be an empty slot another key is claiming)
      if ((ds[q].x & META_USED) && ds[q].y == (uint32_t)kk[q] && ds[q].z != 0 && (uint32_t)(kk[q] >> 32) != 0) {
        const uint32_t Y = (uint32_t)(kk[q] >> 32);
        cp[q] = row_cells(arena, ds[q].z) + (Y & ((1u << meta_lg(ds[q].x)) - 1u));
        cc[q] = *cp[q];
        fast |= 1u << q;
      }
    }
    // (round 4) The commonest INSERT rides the same pipeline: the key's home cell is EMPTY and the row is a small one (its
    // `used` word is the ticket counter).  Its ticket add is issued beside the hits' adds -- all returning atomics of the lane in
    // flight together -- and the claim follows in the next stage; the per-op body did the same steps one dependent round trip after
    // the other, after a second directory look-up and a second load of the cell, with a quarter of the lanes active.  Same
    // protocol as apply_row: a snapshot that shows the row at the reference's threshold defers at once (src/smatrix.c:346), a
    // ticket above the threshold is given back and defers, a claim lost to another tile gives the ticket back and takes the
    // general path (the cell may hold this very key by now).
    uint32_t ins = 0, full = 0;
#pragma unroll
    for (uint32_t q = 0; q < AGG_OPT; q++) {
      if (!(fast & (1u << q))) continue;
      if (cell_key(cc[q]) == (uint32_t)(kk[q] >> 32)) {
        uint32_t* vp = reinterpret_cast<uint32_t*>(cp[q]) + 1;
        if (dbg == 1) old[q] = cell_val(cc[q]);            // hits without their atomic
        else old[q] = OP == OP_INCR ? atomicAdd(vp, tot[q]) : atomicSub(vp, tot[q]);
        if (!(ds[q].x & META_DIRTY)) dir[fmix32((uint32_t)kk[q]) & dmask].meta = ds[q].x | META_DIRTY;
      } else {
        fast &= ~(1u << q);
        if (dbg == 0 && cc[q] == 0 && meta_lg(ds[q].x) < BIG_LG) {
          if (ds[q].w > (1u << meta_lg(ds[q].x)) / 2u) full |= 1u << q;
          else { old[q] = atomicAdd(&dir[fmix32((uint32_t)kk[q]) & dmask].used, 1u); ins |= 1u << q; }
        }
      }
    }
#pragma unroll
    for (uint32_t q = 0; q < AGG_OPT; q++) {
      if (!(ins & (1u << q))) continue;
      DirSlot* d = &dir[fmix32((uint32_t)kk[q]) & dmask];
      if (old[q] > (1u << meta_lg(ds[q].x)) / 2u) {
        atomicSub(&d->used, 1u);
        full |= 1u << q;
      } else {
        // claim the cell AND apply the tile's total in one CAS (apply_row: :354-356 then :241 / :252)
        const uint32_t first = OP == OP_DECR ? 0u - tot[q] : tot[q];
        const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(cp[q]), 0ull, (unsigned long long)pack_cell((uint32_t)(kk[q] >> 32), first));
        if (prev == 0) {
          old[q] = 0;                                        // the cell's value before the tile
          fast |= 1u << q;
          if (!(ds[q].x & META_DIRTY)) d->meta = ds[q].x | META_DIRTY;
          if (CLU && meta_lg(ds[q].x) >= HOME_LG) {          // (its home cell: the row's at-home bitmap, HOME_LG)
            const uint32_t hp = (uint32_t)(kk[q] >> 32) & ((1u << meta_lg(ds[q].x)) - 1u);
            atomicOr(&row_home(arena, ds[q].z, meta_lg(ds[q].x))[hp >> 6], 1ull << (hp & 63u));
          }
        } else {
          atomicSub(&d->used, 1u);
        }
      }
    }
#pragma unroll
    for (uint32_t q = 0; q < AGG_OPT; q++) {
      if (!(have & (1u << q))) continue;
      bool deferred = false;
      if (dbg == 4 && !(fast & (1u << q))) { old[q] = 0; fast |= 1u << q; }     // 4: the slow path (inserts, collisions) left out
      if (full & (1u << q)) {
        deferred = true;                                     // the row stands at its threshold: prep doubles it
      } else if (!(fast & (1u << q))) {
        // a probe that outruns the budget (clustered dense ids) is not walked here, one lane at a time: the op is
        // deferred and the lane-per-op kernel finishes it with the wave-cooperative window probe
        LongProbe lp{false, nullptr, 0, 0};
        uint32_t res = apply_one<OP, SMX_AGG_PATIENT, 1>(dir, dmask, arena, (uint32_t)kk[q], (uint32_t)(kk[q] >> 32), tot[q], &deferred, &lp, dbg == 5, !RET,
                                                         false, nullptr, CLU ? HINT_BUDGET : PROBE_BUDGET, CLU);
        if (lp.need) {
          // (round 4) ... unless the key's cell is remembered (ArenaHead): then this is a hit like any other.  One hinted key in
          // 256 counts for 256 long probes: the host's evidence that the table is still clustered
          const uint32_t p = CLU ? hint_find(arena, lp.cells, lp.mask, (uint32_t)(kk[q] >> 32)) : 0xFFFFFFFFu;
          if (CLU && p != 0xFFFFFFFFu) {
            uint32_t* vp = reinterpret_cast<uint32_t*>(const_cast<uint64_t*>(&lp.cells[p])) + 1;
            res = OP == OP_INCR ? atomicAdd(vp, tot[q]) + tot[q] : atomicSub(vp, tot[q]) - tot[q];
            if (((tid ^ blockIdx.x) & 255u) == 0) atomicAdd(&ctl->n_long_ops, 256u);
          } else { deferred = true; ctl->n_long = 1; }
        }
        old[q] = OP == OP_INCR ? res - tot[q] : res + tot[q];
      }
      l_sum[hh[q]] = old[q];                                  // the cell's value before the tile
      reinterpret_cast<uint32_t*>(&l_key[hh[q]])[0] = deferred ? 1u : 0u;
    }
  }
  __syncthreads();
  // phase 3
  uint32_t dmask_k = 0;          // which of this lane's ops are deferred
#pragma unroll
  for (uint32_t k = 0; k < AGG_OPT; k++) {
    bool deferred = false;
    if (slot[k] == ~0u - 1) {
      LongProbe lp{false, nullptr, 0, 0};
      uint32_t r = apply_one<OP, false, 1>(dir, dmask, arena, xs[(size_t)j[k] * ST], ys[(size_t)j[k] * ST], V[k], &deferred, &lp);
      if (lp.need) { deferred = true; ctl->n_long = 1; }
      if (!deferred && RET) out[j[k]] = r;
    } else if (slot[k] != ~0u) {
      deferred = reinterpret_cast<uint32_t*>(&l_key[slot[k]])[0] != 0;
      if (!deferred && RET) {                                  // (!RET: the caller does not want the results)
        const uint32_t old = l_sum[slot[k]];
        out[j[k]] = OP == OP_INCR ? old + pre[k] + V[k] : old - pre[k] - V[k];
      }
    }
    if (deferred) dmask_k |= 1u << k;
  }
  // deferred ops: ONE global atomic per workgroup (a per-wave atomic on the single list
  // counter was the kernel's critical path when a few % of the ops defer)
  __syncthreads();                       // everybody is done with l_sum / l_n
  if (tid == 0) l_n = 0;
  __syncthreads();
  uint32_t mine = __popc(dmask_k), at = 0;
  if (mine) at = atomicAdd(&l_n, mine);
  __syncthreads();
  if (tid == 0 && l_n) l_sum[0] = atomicAdd(&ctl->n_defer, l_n);
  __syncthreads();
  if (mine) {
    at += l_sum[0];
#pragma unroll
    for (uint32_t k = 0; k < AGG_OPT; k++)
      if (dmask_k & (1u << k)) defer[at++] = j[k];
  }
}

// ---- set batches: the same fold, keeping each key's LAST op --------------------------------------------------
// A set batch resolves duplicates highest-index-wins (include/smatrix_batch.h).  One atomicExch per op serialises on
// the hot cells exactly like un-folded incrs did (13 ms per 2^24 Zipf ops), and five passes over ALL ops then put the
// right values in.  Here a tile first reduces its ops to one WINNER per distinct key (LDS claim + LDS atomicMax on the op
// index); only winners touch the table -- found or inserted like any write, their value lands for now -- and only
// winners enter the passes that settle the order ACROSS tiles after the rounds (k_set_*_e below: locate, clear, rank by
// atomicMax of the op index, pick, store): ~0.7 n entries, at most one per tile on a hot cell.
// set returns the value it was given (src/smatrix.c:230): out[i] = v[i], written at once.
// LDS empty marker: key 0 = (x 0, y 0), which never enters the table (y == 0 ops take the per-op body: quirk Q1).
template <uint32_t ST = 1>
__global__ __launch_bounds__(AGG_THREADS) __attribute__((amdgpu_num_sgpr(SMX_AGG_SGPRS))) void k_set_fold(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n,
    const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys, const uint32_t* __restrict__ vs,
    uint32_t* __restrict__ out, uint32_t* defer, uint32_t* __restrict__ ent_idx, uint64_t* __restrict__ ent_cell) {
  // ent_cell[e]: the cell the entry's key lives in, as found (or created) HERE, its value word cleared for the ranking pass.
  // The address holds while no row is created or doubled: a batch that round 0 completes -- every key present, or
  // inserted without a structure change -- goes straight to the ranking pass and spares k_set_locate_e, the most
  // expensive of the entry passes (0.93 of 2.67 ms per 2^24 sets on present keys).  Clearing early is harmless: every
  // cell a set op names ends the batch with its winner's value, a (key, 0) cell stays a live cell for every probe and
  // rehash, and all clears of this kernel are over before the first atomicMax of the next one.
  __shared__ uint64_t l_key[AGG_SLOTS];     // (x | y<<32), 0 = empty
  __shared__ uint32_t l_win[AGG_SLOTS];     // highest op index + 1 among the tile's ops on the key
  __shared__ uint16_t l_list[AGG_TILE];
  __shared__ uint32_t l_n, l_base;
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < AGG_SLOTS; i += AGG_THREADS) { l_key[i] = 0ull; l_win[i] = 0; }
  if (tid == 0) l_n = 0;
  __syncthreads();
  const uint32_t tile0 = blockIdx.x * AGG_TILE;
  uint32_t own = 0;                         // this lane's ops with y == 0
#pragma unroll
  for (uint32_t k = 0; k < AGG_OPT; k++) {
    const uint32_t t = tile0 + k * AGG_THREADS + tid;
    if (t >= n) continue;
    const uint32_t X = xs[(size_t)t * ST], Y = ys[(size_t)t * ST];
    out[t] = vs[(size_t)t * ST];
    if (Y == 0) { own |= 1u << k; continue; }
    const uint64_t key = (uint64_t)X | ((uint64_t)Y << 32);
    uint32_t h = (X * 0x9E3779B1u) ^ (Y * 0x85EBCA77u);
    h = (h ^ (h >> 15)) & (AGG_SLOTS - 1);
    for (;;) {
      const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&l_key[h]), 0ull, (unsigned long long)key);
      if (prev == 0ull) l_list[atomicAdd(&l_n, 1u)] = (uint16_t)h;
      if (prev == 0ull || prev == key) break;
      h = (h + 1) & (AGG_SLOTS - 1);
    }
    atomicMax(&l_win[h], t + 1u);
  }
  __syncthreads();
  const uint32_t nd = l_n;
  uint32_t dm = 0, wj[AGG_OPT];             // winners that could not be applied: they go to the round loop
#pragma unroll
  for (uint32_t q = 0; q < AGG_OPT; q++) {
    const uint32_t i = tid + q * AGG_THREADS;
    uint32_t e = 0;
    if (i < nd) {
      const uint32_t h = l_list[i];
      const uint64_t key = l_key[h];
      const uint32_t w = l_win[h] - 1u;
      bool deferred = false;
      LongProbe lp{false, nullptr, 0, 0};
      uint64_t where = ~0ull;
      apply_one<OP_SET, true, 1>(dir, dmask, arena, (uint32_t)key, (uint32_t)(key >> 32), vs[(size_t)w * ST], &deferred, &lp,
                                 false, false, true, &where);
      if (lp.need) { deferred = true; ctl->n_long = 1; }
      if (deferred) { dm |= 1u << q; wj[q] = w; where = ~0ull; }
      e = w + 1u;
      if (where != ~0ull) reinterpret_cast<uint32_t*>(arena)[where * 2 + 1] = 0;
      ent_cell[tile0 + i] = where;
    }
    ent_idx[tile0 + i] = e;                 // (the entry arrays hold gridDim.x * AGG_TILE slots)
  }
#pragma unroll
  for (uint32_t k = 0; k < AGG_OPT; k++) {
    if (!(own & (1u << k))) continue;
    const uint32_t t = tile0 + k * AGG_THREADS + tid;
    bool deferred = false;
    LongProbe lp{false, nullptr, 0, 0};
    apply_one<OP_SET, false, 1>(dir, dmask, arena, xs[(size_t)t * ST], 0u, vs[(size_t)t * ST], &deferred, &lp);
    if (lp.need) { deferred = true; ctl->n_long = 1; }
    if (deferred) { dm |= 1u << (AGG_OPT + k); }
  }
  // deferred ops: one reservation per workgroup
  __syncthreads();
  if (tid == 0) l_n = 0;
  __syncthreads();
  const uint32_t mine = __popc(dm);
  uint32_t at = 0;
  if (mine) at = atomicAdd(&l_n, mine);
  __syncthreads();
  if (tid == 0 && l_n) l_base = atomicAdd(&ctl->n_defer, l_n);
  __syncthreads();
  if (mine) {
    at += l_base;
#pragma unroll
    for (uint32_t q = 0; q < AGG_OPT; q++)
      if (dm & (1u << q)) defer[at++] = wj[q];
#pragma unroll
    for (uint32_t k = 0; k < AGG_OPT; k++)
      if (dm & (1u << (AGG_OPT + k))) defer[at++] = tile0 + k * AGG_THREADS + tid;
  }
}

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
    FreeLists fl, uint32_t st, uint32_t create_only, uint32_t wpo_max) {
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
        else if (fh.state == FAR_ABSENT) occ = fh.occ;
      }
    }
    while (__any(lp.need)) {                        // long sequences (dense ids): the wave finishes them (coop_probe)
      const uint32_t p = coop_probe(lp.need, lp.cells, lp.mask, Y, lp.pos, reinterpret_cast<const ArenaHead*>(arena)->home_on != 0, occ);
      if (lp.need) {
        lp.need = false;
        absent = p == PROBE_NONE || cell_key(lp.cells[p]) != Y;     // the table is quiescent here: the answer is final
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
    FreeLists fl, uint32_t st, uint32_t create_only, uint32_t wpo_max) {
  prep_body(SMX_VG, ctl, dir, dmask, dir_limit, arena, arena_cap_units, defer, xs, ys, tasks, klist, kcap, rebal, fl, st, create_only, wpo_max);
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
__device__ __forceinline__ void grow_plan_body(VGrid g, Ctl* ctl, GrowTask* tasks, uint64_t arena_cap_units, FreeLists fl,
                                               uint32_t task_budget, uint32_t chunk_cap) {
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
    if (threadIdx.x == 0) { l_chunks = 0; l_units = 0; l_chunk_ok = 1; l_unit_ok = 1; }
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
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_grow_plan(Ctl* ctl, GrowTask* tasks, uint64_t arena_cap_units, FreeLists fl,
                                                   uint32_t task_budget, uint32_t chunk_cap) {
  grow_plan_body(SMX_VG, ctl, tasks, arena_cap_units, fl, task_budget, chunk_cap);
}

// chunk -> task maps, filled one wave per CHUNKED task (prep lists them: a steady batch has ~50 of them among 60 000
// tasks, and a wave per task of ALL kinds made this trivial pass 40 us of the growth round's critical path)
// arena != nullptr (clustered rows, k_grow_move_home): GrowTask::wrap_from is worked out as well
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
    if (arena && k.new_base != 0) {
      // GrowTask::wrap_from: the table's first run, window by window up to its first empty slot
      __shared__ uint32_t l_wrap, l_end;
      if (threadIdx.x == 0) { l_wrap = 0xFFFFFFFFu; l_end = 0xFFFFFFFFu; }
      __syncthreads();
      const uint64_t* O = row_cells(arena, k.old_base);
      const uint32_t old_size = 1u << k.old_lg;
      for (uint32_t b0 = 0; b0 < old_size; b0 += blockDim.x) {                 // block-uniform
        const uint32_t p = b0 + threadIdx.x;
        const uint64_t c = O[p];
        if (c == 0) atomicMin(&l_end, p);
        __syncthreads();
        if (c != 0 && p < l_end && (cell_key(c) & (old_size - 1u)) > p) atomicMin(&l_wrap, cell_key(c) & (old_size - 1u));
        const bool done = l_end != 0xFFFFFFFFu;                                   // (uniform: read between two barriers)
        __syncthreads();
        if (done) break;
      }
      __syncthreads();
      if (threadIdx.x == 0) tasks[t].wrap_from = l_wrap;
      __syncthreads();
    }
  }
}
__global__ __launch_bounds__(256) void k_grow_map(const Ctl* ctl, GrowTask* tasks, const uint32_t* list,
                                                  uint32_t* map_old, uint32_t* map_new, uint8_t* arena) {
  grow_map_body(SMX_VG, ctl, tasks, list, map_old, map_new, arena);
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
template <typename S>
__device__ __forceinline__ void grow_lds_task(GrowTask* task, uint8_t* arena, uint64_t* l_old, uint32_t* l_tab,
                                              uint32_t* l_cd) {
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
      const uint64_t c = r == NONE ? 0ull : l_old[r];
      T[q] = c;
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

// one workgroup (THREADS = 64: one wave) per task of the given kind
template <int THREADS, uint32_t MAX_LG>
__global__ __launch_bounds__(THREADS) void k_grow_lds(const Ctl* ctl, GrowTask* tasks, const uint32_t* list,
                                                      uint32_t kind, uint8_t* arena) {
  extern __shared__ uint64_t l_dyn[];                               // 2^MAX_LG cells ...
  uint32_t* l_tab = reinterpret_cast<uint32_t*>(l_dyn + (1u << MAX_LG));   // ... and 2^(MAX_LG+1) slot indices
  __shared__ uint32_t l_cd[2];
  const uint32_t n = ctl->n_kind[kind];
  for (uint32_t li = blockIdx.x; li < n; li += gridDim.x)                     // block-uniform
    grow_lds_task<BlockScope<THREADS>>(&tasks[list[li]], arena, l_dyn, l_tab, l_cd);
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
__device__ __forceinline__ void grow_move_home_body(VGrid g, const Ctl* ctl, GrowTask* tasks, const uint32_t* map_old,
                                                    uint8_t* arena) {
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
    if (lane == 0) {
      // the masks ARE the new table's at-home bitmap (HOME_LG): they stay behind the block for the op kernels' probes
      unsigned long long* hb = row_home(arena, k.new_base, k.old_lg + 1);
      hb[c] = lo_m;
      hb[c + (old_size >> 6)] = hi_m;
    }
  }
}
__global__ __launch_bounds__(256) void k_grow_move_home(const Ctl* ctl, GrowTask* tasks, const uint32_t* map_old, uint8_t* arena) {
  grow_move_home_body(SMX_VG, ctl, tasks, map_old, arena);
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
    if (by_lds && rest_by_lds(k)) continue;                                      // (k_grow_rest_lds places this row's displaced cells)
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
  return ((size_t)1 << (REST_LDS_MAX_LG - 3)) + ((size_t)1 << (REST_LDS_MAX_LG - 9)) + (size_t)REST_WAVES * REST_STAGE * 8 + (size_t)REST_WAVES * REST_BUCKETS * 4 + (REST_WAVES + 2) * 4;
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

static_assert(((size_t)1 << (REST_LDS_MAX_LG - 3)) + ((size_t)1 << (REST_LDS_MAX_LG - 9)) + (size_t)REST_WAVES * REST_STAGE * 8 + (size_t)REST_WAVES * REST_BUCKETS * 4 + (REST_WAVES + 2) * 4 <= 160 * 1024,
              "k_grow_rest_lds: the LDS of one CU");
// dbg (measurement runs only, SMATRIX_REST_DBG): counters {steps, rounds, cells, most steps of one wave, trips, most trips of one
// wave}; bit 0 of dbg_mode: the staged cells are dropped instead of placed (what the loads alone cost: tables wrong afterwards)
__global__ __launch_bounds__(REST_THREADS) void k_grow_rest_lds(const Ctl* ctl, GrowTask* tasks, const uint32_t* list, uint8_t* arena,
                                                                unsigned long long* dbg, uint32_t dbg_mode) {
  extern __shared__ unsigned long long l_rest[];
  unsigned long long* B = l_rest;                                           // 2^(REST_LDS_MAX_LG - 6) words
  unsigned long long* S = B + (1u << (REST_LDS_MAX_LG - 6));                // 2^(REST_LDS_MAX_LG - 12) words
  uint64_t* stage_all = reinterpret_cast<uint64_t*>(S + (1u << (REST_LDS_MAX_LG - 12)));
  uint32_t* scratch_all = reinterpret_cast<uint32_t*>(stage_all + REST_WAVES * REST_STAGE);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint64_t* stage = stage_all + wave * REST_STAGE;                          // {key, old slot + 1} of this wave's pending displaced cells, in old slot order
  uint32_t* bucket = scratch_all + wave * REST_BUCKETS;
  uint32_t* bound = scratch_all + REST_WAVES * REST_BUCKETS;                // where each wave's range of old slots begins
  const uint32_t n = aload(&ctl->n_kind[GROW_CHUNKED]);
  for (uint32_t li = blockIdx.x; li < n; li += gridDim.x) {                 // block-uniform
    const GrowTask k = tasks[list[li]];
    if (k.new_base == 0 || grow_kind(k.old_lg) != GROW_CHUNKED || k.chunk0 == CHUNK_NONE || !rest_by_lds(k)) continue;
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
    // this wave's range of old slots: from the first empty old slot at/after its nominal start to the one of the next wave
    {
      uint32_t b = wave * (old_size / REST_WAVES);
      if (wave != 0) {
        for (bool found = false; !found;) {                                 // (wave-uniform; eight 64-cell windows in flight)
          uint64_t c[8];
#pragma unroll
          for (int q = 0; q < 8; q++) { const uint32_t p = b + (uint32_t)q * 64u + lane; c[q] = p < old_size ? O[p] : 1ull; }
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const uint64_t m = __ballot(c[q] == 0);
            if (m && !found) { b += (uint32_t)q * 64u + (uint32_t)__ffsll((unsigned long long)m) - 1u; found = true; }
          }
          if (!found) { b += 512; if (b >= old_size) { b = old_size; found = true; } }
        }
      }
      if (lane == 0) bound[wave] = b;
      if (threadIdx.x == 0) { bound[REST_WAVES] = old_size; bound[REST_WAVES + 1] = 0; }
    }
    __syncthreads();
    const uint32_t lo = bound[wave], hi = bound[wave + 1];
    // A table whose first run continues its last one round the end (wrapped cells: GrowTask::wrap_from): the wrapped cells sit
    // in the FIRST piece and come first in old slot order, but land among the cells of the LAST piece -- so the wave that holds
    // the end of the table starts only when wave 0 is through (bound[REST_WAVES + 1]); all other pieces stay independent.
    if (k.wrap_seen != 0xFFFFFFFFu && wave != 0 && hi == old_size && lo < hi)
      while (__hip_atomic_load(&bound[REST_WAVES + 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(8);
    uint32_t n_st = 0;                                                      // staged cells (wave-uniform)
    uint32_t d_steps = 0, d_rounds = 0, d_trips = 0;
    // a step: the first `cnt` staged cells (cnt <= 64), lane l = the l-th of them in old slot order
    auto place = [&](uint32_t cnt) {
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
    if (dbg && lane == 0) {
      atomicAdd(&dbg[0], (unsigned long long)d_steps); atomicAdd(&dbg[1], (unsigned long long)d_rounds);
      atomicMax(&dbg[3], (unsigned long long)d_steps); atomicAdd(&dbg[4], (unsigned long long)d_trips); atomicMax(&dbg[5], (unsigned long long)d_trips);
      atomicMax(&dbg[6], (unsigned long long)d_rounds);
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
      if (c != 0) {
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
                                                 FreeLists fl) {
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
                                                     FreeLists fl) {
  grow_commit_body(SMX_VG, ctl, tasks, dir, arena, fl);
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

// ---- the far join's kernels (see "far join" above) ---------------------------------------------------------------------------
// k_far_rows: every row of >= 2^FAR_ROW_LG cells takes its units (one atomic add: the order does not matter), fills the unit ->
// row map and enters F as {row block, 0} -> first unit.  A row that does not fit the capacities is left out.
__global__ __launch_bounds__(256) void k_far_rows(Ctl* ctl, const DirSlot* dir, uint32_t dir_size, uint32_t* unit_row, uint32_t cap_units, uint4* tab,
                                                  uint32_t tmask) {
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t h0 = blockIdx.x * blockDim.x; h0 < dir_size; h0 += gridDim.x * blockDim.x) {      // (block-uniform: dir_size is a multiple of 256)
    const uint32_t h = h0 + threadIdx.x;
    const DirSlot d = dir[h];
    const bool big = (d.meta & META_USED) && d.base != 0 && meta_lg(d.meta) >= FAR_ROW_LG;
    const uint32_t units = big ? 1u << (meta_lg(d.meta) - FAR_UNIT_LG) : 0u;
    // one reservation per WAVE (10^5 rows adding to one word one by one were 1 ms of every batch)
    uint32_t incl = units;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, dd);
      if ((int)lane >= dd) incl += o;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    const uint64_t bm = __ballot(big);
    // ... and per WORKGROUP (every atomic on these two words queues at the memory side: 65 000 wave-level adds were still 1 ms)
    __shared__ uint32_t l_tot[4], l_big[4], l_base;
    const uint32_t wv = threadIdx.x >> 6;
    if (lane == 0) { l_tot[wv] = total; l_big[wv] = (uint32_t)__popcll(bm); }
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t t4 = l_tot[0] + l_tot[1] + l_tot[2] + l_tot[3], b4 = l_big[0] + l_big[1] + l_big[2] + l_big[3];
      l_base = t4 ? atomicAdd(&ctl->n_units, t4) : 0u;
      if (b4) atomicAdd(&ctl->n_big, b4);
    }
    __syncthreads();
    uint32_t base = l_base;
    for (uint32_t q = 0; q < wv; q++) base += l_tot[q];
    __syncthreads();                                                       // (the scratch is reused by the next trip)
    if (!bm) continue;
    if (!unit_row) continue;                                               // (unit_row == nullptr: counting only, the host sizes its buffers)
    const uint32_t first = base + incl - units;
    // the unit -> row map, a row at a time with the whole wave (the lane of a 2^21-cell row wrote its 4096 entries alone: 1 ms)
    for (uint64_t todo = bm; todo; todo &= todo - 1) {                     // (wave-uniform)
      const int src = __ffsll((unsigned long long)todo) - 1;
      const uint32_t f = (uint32_t)__shfl((int)first, src), n = (uint32_t)__shfl((int)units, src), hh = h0 + (threadIdx.x & ~63u) + (uint32_t)src;
      for (uint32_t u = lane; u < n && (uint64_t)f + u < cap_units; u += 64) unit_row[f + u] = hh;           // (every unit below the capacity names ITS row)
    }
    if (big && (uint64_t)first + units <= cap_units) far_insert(tab, tmask, d.base, 0u, first);
  }
}

// k_far_keys: the deferred ops whose probe outruns the lane's budget on a row of >= 2^HOME_LG cells (what the wave-per-op pass is
// going to find out again: nothing changes in between) enter F.  `limit`: ops beyond it are not entered (the table would fill up).
__global__ __launch_bounds__(256) void k_far_keys(Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, const uint32_t* idx,
                                                  const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys, uint32_t st, uint4* tab,
                                                  uint32_t tmask, uint32_t limit) {
  const uint32_t n = min(aload(&ctl->n_prev), limit);
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const size_t at = (size_t)idx[t] * st;
    const uint32_t Y = ys[at];
    uint4 s;
    if (Y == 0 || !dir_find(dir, dmask, xs[at], &s) || s.z == 0 || meta_lg(s.x) < FAR_ROW_LG) continue;
    const uint32_t mask = (1u << meta_lg(s.x)) - 1u;
    const uint64_t* cells = row_cells(arena, s.z);
    uint32_t pos = Y & mask;
    bool far = true;
    for (uint32_t step = 0; step <= HINT_BUDGET; step++) {
      const uint64_t c = cells[pos];
      if (cell_key(c) == Y || c == 0) { far = false; break; }
      pos = (pos + 1) & mask;
    }
    if (far && !far_insert(tab, tmask, s.z, Y, FAR_NOT_FOUND)) reinterpret_cast<ArenaHead*>(arena)->far_overflow = 1;
  }
  // (ops beyond the limit are not in the table: two ops naming one new key could then take different paths -- no claimed inserts)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->far_nd = aload(&ctl->n_prev);
    if (aload(&ctl->n_prev) > limit) reinterpret_cast<ArenaHead*>(arena)->far_overflow = 1;
  }
}

// k_far_scan: a wave per unit: the occupancy words (a (0, v) cell counts as free: it may turn back into an empty one, quirk Q1),
// the unit's count of free cells, and every displaced cell's slot into its key's entry of F, if it has one.
__global__ __launch_bounds__(256) void k_far_scan(const Ctl* ctl, const DirSlot* dir, const uint32_t* unit_row, uint32_t cap_units, uint8_t* arena,
                                                  uint4* tab, uint32_t tmask, unsigned long long* occ, uint32_t* zeros) {
  const uint32_t n_units = min(aload(&ctl->n_units), cap_units);
  const uint32_t lane = threadIdx.x & 63u, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t u = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; u < n_units; u += nwaves) {     // (wave-uniform)
    const DirSlot d = dir[unit_row[u]];
    const uint32_t mask = (1u << meta_lg(d.meta)) - 1u;
    const uint4* row = far_entry(tab, tmask, d.base, 0u);                  // (a row that did not fit whole has no entry: its units are skipped)
    if (!row) { if (lane == 0) zeros[u] = 0xFFFFFFFFu; continue; }
    const uint32_t p0 = (u - row->z) << FAR_UNIT_LG;
    const uint64_t* cells = row_cells(arena, d.base) + p0;
    uint64_t c[FAR_UNIT_WORDS];
#pragma unroll
    for (uint32_t q = 0; q < FAR_UNIT_WORDS; q++) c[q] = cells[q * 64u + lane];
    uint32_t free_cells = 0;
#pragma unroll
    for (uint32_t q = 0; q < FAR_UNIT_WORDS; q++) {
      const uint32_t p = p0 + q * 64u + lane, key = cell_key(c[q]);
      const bool taken = c[q] != 0 && key != 0;
      const uint64_t m = __ballot(taken);
      free_cells += 64u - (uint32_t)__popcll(m);
      if (lane == 0) occ[(size_t)u * FAR_UNIT_WORDS + q] = m;
      if (taken && (key & mask) != p) {
        uint4* e = far_entry(tab, tmask, d.base, key);
        if (e) e->z = p;
      }
    }
    if (lane == 0) zeros[u] = free_cells;
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
  const uint32_t keep_long = ctl->n_long, keep_oom = ctl->arena_oom, keep_long_ops = ctl->n_long_ops;
  uint64_t* z = reinterpret_cast<uint64_t*>(ctl);
  for (uint32_t i = 0; i < CTL_ROUND_BYTES / 8; i++) z[i] = 0;
  ctl->n_long = keep_long;                           // (sticky for the batch: the host switches the retries to lane-per-op)
  ctl->n_long_ops = keep_long_ops;                   // (summed over the rounds of a chain)
  ctl->arena_oom = keep_oom;
}

// ---- set: duplicates of one cell inside a batch resolve highest-index-wins ----
// (the reference's threads would leave "some" value; the batch contract pins it)
// After the rounds (structure final): where does each set's cell live?  y==0 sets were
// applied in place (quirk Q1 path) and take no part.
__global__ __launch_bounds__(256) void k_set_locate(DirSlot* dir, uint32_t dmask, uint8_t* arena,
                                                    uint32_t n, const uint32_t* __restrict__ xs,
                                                    const uint32_t* __restrict__ ys, uint64_t* cellp, uint32_t st) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = j < n;
  uint64_t where = ~0ull;
  const uint32_t Y = live ? ys[(size_t)j * st] : 0u;
  uint4 s = {0, 0, 0, 0};
  DirSlot* d = live && Y ? dir_find(dir, dmask, xs[(size_t)j * st], &s) : nullptr;
  LongProbe lp{false, nullptr, 0, 0};
  if (d && s.z) {
    const uint32_t mask = (1u << meta_lg(s.x)) - 1u;
    const uint64_t* cells = row_cells(arena, s.z);
    uint32_t pos = Y & mask;
    for (uint32_t step = 0; step <= mask; step++) {
      uint64_t c = cells[pos];
      if (cell_key(c) == Y) { where = (((uint64_t)s.z) << 4) + pos; break; }
      if (c == 0) break;
      pos = (pos + 1) & mask;
      if (step >= PROBE_BUDGET) { lp = LongProbe{true, cells, mask, pos}; break; }
    }
  }
  while (__any(lp.need)) {
    const uint32_t p = coop_probe(lp.need, lp.cells, lp.mask, Y, lp.pos);
    if (lp.need) {
      lp.need = false;
      if (p != PROBE_NONE && cell_key(lp.cells[p]) == Y) where = (((uint64_t)s.z) << 4) + p;
    }
  }
  if (live) cellp[j] = where;
}
__global__ __launch_bounds__(256) void k_set_clear(uint32_t n, const uint64_t* cellp, uint8_t* arena) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n && cellp[j] != ~0ull)
    reinterpret_cast<uint32_t*>(arena)[cellp[j] * 2 + 1] = 0;
}
__global__ __launch_bounds__(256) void k_set_rank(uint32_t n, const uint64_t* cellp, uint8_t* arena) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n && cellp[j] != ~0ull)
    atomicMax(&reinterpret_cast<uint32_t*>(arena)[cellp[j] * 2 + 1], j + 1);
}
__global__ __launch_bounds__(256) void k_set_pick(uint32_t n, uint64_t* cellp, uint8_t* arena) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n && cellp[j] != ~0ull)
    if (reinterpret_cast<uint32_t*>(arena)[cellp[j] * 2 + 1] != j + 1) cellp[j] = ~0ull;  // loser
}
__global__ __launch_bounds__(256) void k_set_store(uint32_t n, const uint64_t* cellp,
                                                   const uint32_t* vs, uint8_t* arena, uint32_t st) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n && cellp[j] != ~0ull)
    reinterpret_cast<uint32_t*>(arena)[cellp[j] * 2 + 1] = vs[(size_t)j * st];
}

// the same five passes over the ENTRIES of k_set_fold (ent_idx[e] = winner's op index + 1, 0 = no entry)
__global__ __launch_bounds__(256) void k_set_locate_e(DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n_ent,
                                                      const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
                                                      uint32_t* ent_idx, uint64_t* ent_cell, uint32_t st) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t w1 = e < n_ent ? ent_idx[e] : 0u;
  const bool live = w1 != 0;
  const uint32_t j = w1 - 1u;
  uint64_t where = ~0ull;
  const uint32_t Y = live ? ys[(size_t)j * st] : 0u;
  uint4 s = {0, 0, 0, 0};
  DirSlot* d = live ? dir_find(dir, dmask, xs[(size_t)j * st], &s) : nullptr;
  LongProbe lp{false, nullptr, 0, 0};
  if (d && s.z) {
    const uint32_t mask = (1u << meta_lg(s.x)) - 1u;
    const uint64_t* cells = row_cells(arena, s.z);
    uint32_t pos = Y & mask;
    for (uint32_t step = 0; step <= mask; step++) {
      const uint64_t c = cells[pos];
      if (cell_key(c) == Y) { where = (((uint64_t)s.z) << 4) + pos; break; }
      if (c == 0) break;
      pos = (pos + 1) & mask;
      if (step >= PROBE_BUDGET) { lp = LongProbe{true, cells, mask, pos}; break; }
    }
  }
  while (__any(lp.need)) {
    const uint32_t p = coop_probe(lp.need, lp.cells, lp.mask, Y, lp.pos);
    if (lp.need) {
      lp.need = false;
      if (p != PROBE_NONE && cell_key(lp.cells[p]) == Y) where = (((uint64_t)s.z) << 4) + p;
    }
  }
  if (live) {
    ent_cell[e] = where;
    if (where == ~0ull) ent_idx[e] = 0;
    else reinterpret_cast<uint32_t*>(arena)[where * 2 + 1] = 0;          // (k_set_clear's job, done here)
  }
}
__global__ __launch_bounds__(256) void k_set_rank_e(uint32_t n_ent, const uint32_t* ent_idx, const uint64_t* ent_cell, uint8_t* arena) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_ent && ent_idx[e]) atomicMax(&reinterpret_cast<uint32_t*>(arena)[ent_cell[e] * 2 + 1], ent_idx[e]);
}
__global__ __launch_bounds__(256) void k_set_pick_e(uint32_t n_ent, uint32_t* ent_idx, const uint64_t* ent_cell, uint8_t* arena) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_ent && ent_idx[e] && reinterpret_cast<uint32_t*>(arena)[ent_cell[e] * 2 + 1] != ent_idx[e]) ent_idx[e] = 0;   // loser
}
__global__ __launch_bounds__(256) void k_set_store_e(uint32_t n_ent, const uint32_t* ent_idx, const uint64_t* ent_cell,
                                                     const uint32_t* __restrict__ vs, uint8_t* arena, uint32_t st) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_ent && ent_idx[e]) reinterpret_cast<uint32_t*>(arena)[ent_cell[e] * 2 + 1] = vs[(size_t)(ent_idx[e] - 1u) * st];
}
// (Round 3 tried three passes instead -- every entry stores its id in the cell's value word, one 64-bit atomicMax of
//  {op index, value} on a side slot of the id that stayed, that entry writes the winner's value -- and reverted: with plain
//  stores of the ids two entries of one key on different XCDs each read THEIR id back in the next kernel (conflicting
//  plain stores to one word are not reconciled by a kernel boundary on this chip: two representatives per key, ~100
//  wrong cells per 1.5 M-op Zipf batch, caught by tests/soak.py), and with agent-scope atomic stores the passes cost
//  3.5 ms per 2^24 sets against 2.6 for the four below.  DESIGN.md "Measured and rejected".)

// the deferred list of a batch into an EMPTY matrix: every op, in order (run_write)
__global__ __launch_bounds__(256) void k_iota(uint32_t* out, uint32_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (uint32_t)i;
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

// ---- random-access probes (include/smx_probe.h) ------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_probe_random(uint64_t* buf, uint64_t words, uint64_t touches,
                                                      uint64_t seed, unsigned long long* sink) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t acc = 0;
  for (uint64_t i = t; i < touches; i += 4 * stride) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint64_t j = i + k * stride;
      if (j >= touches) break;
      const uint64_t w = splitmix_at(seed, j) % words;
      if (MODE == 0) acc += buf[w];
      else if (MODE == 1) acc += atomicAdd(reinterpret_cast<uint32_t*>(&buf[w]), 1u);
      else if (MODE == 2) atomicAdd(reinterpret_cast<uint32_t*>(&buf[w]), 1u);
      else {
        const uint4 a = *reinterpret_cast<const uint4*>(&buf[w & ~1ull]);
        const uint64_t w2 = (splitmix_at(seed ^ a.x, j) + a.y) % words;
        acc += buf[w2];
      }
    }
  }
  if (MODE != 2 && acc == 0x1234567deadbeefULL) *sink = acc;   // keeps the loads alive
}

// ---- stream generator (include/smx_stream.h) -----------------------------------------
__device__ inline uint32_t draw_id(int dist, uint32_t n_ids, const double* cdf, int scramble, uint64_t r) {
  uint32_t id;
  if (dist == 0) {
    id = 1u + (uint32_t)(r % n_ids);
  } else {
    double u = (double)(r >> 11) * 0x1.0p-53;
    uint32_t lo = 0, hi = n_ids - 1;
    while (lo < hi) {
      uint32_t mid = lo + (hi - lo) / 2;
      if (cdf[mid] < u) lo = mid + 1; else hi = mid;
    }
    id = lo + 1;
  }
  return scramble ? fmix32(id) : id;
}

__global__ __launch_bounds__(256) void k_stream_fill(int dist, uint64_t seed, uint32_t n_ids,
                                                     const double* cdf, int scramble, uint64_t first,
                                                     uint64_t n, uint32_t* x, uint32_t* y, uint64_t per_row) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t op = first + i;
  if (dist == 2) {                                   // SMX_DIST_CF: row 1 + op / per_row, one uniform column draw per op
    const uint32_t row = 1u + (uint32_t)(op / per_row), col = 1u + (uint32_t)(splitmix_at(seed, op) % n_ids);
    x[i] = scramble ? fmix32(row) : row;
    y[i] = scramble ? fmix32(col) : col;
    return;
  }
  x[i] = draw_id(dist, n_ids, cdf, scramble, splitmix_at(seed, 2 * op));
  y[i] = draw_id(dist, n_ids, cdf, scramble, splitmix_at(seed, 2 * op + 1));
}

}  // namespace smx
