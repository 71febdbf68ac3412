// kernels/ops.hpp -- the op kernels: per-op body, long probes (wave-cooperative window probe, hint table, far join), lane-per-op / wave-per-op / folding kernels, set passes.
// A fragment of smx_kernels.hpp (round 5: the 4 500-line header split by concern, no kernel changed): included there, in order,
// INSIDE namespace smx; not a header of its own.

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
// per step).  A table of {y, row base, slot} entries (round 6: slots of two {tag, slot} entries; the matrix allocates it when its
// tables turn out clustered) is consulted when a probe has used up its budget, and written when a wave-cooperative probe has ended on the key.  An entry
// is a HINT: it counts only if the cell it names holds y in the row's CURRENT block (a doubled row has a new base; a torn or
// overwritten entry fails the same test), and a key sits in one cell of its table -- with one exception, the twins of quirk
// Q1: a (0, v) cell whose value returns to 0 becomes an empty cell, a key behind it can then be inserted a second time in
// front, and the reference's probe from home finds THAT one.  So the first write op that leaves a (0, 0) cell behind switches
// the hints off for the matrix (`y0_zeroed`, sticky): they are an accelerator for dense-id streams, not a structure.
// Unit 0 of the arena (base 0 = "no block") holds the words the kernels need for this, so that no kernel signature grows.
struct ArenaHead {
  uint32_t y0_zeroed;     // a y == 0 write has left a (0, 0) cell (see above)
  uint32_t hint_mask;     // slots - 1 of the hint table; 0: none
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
  // claimed inserts by RANK (far_claim_insert): the occupancy words as the scan left them, and per word how many of its free
  // cells have been handed out -- parallel to far_occ
  const unsigned long long* far_occ0;
  uint32_t* far_clm;
  // (round 5) the clustered folding kernel keeps the ops it defers in TWO lists while this is set: ops whose probe outran the
  // budget in the usual one -- the pass in front of prep finishes most of them --, ops whose key is known to be ABSENT from a row
  // that stands at its threshold (or from no row at all) here: they wait for prep whatever that pass does, which therefore
  // never sees them (Ctl::n_absent, k_round_advance)
  uint32_t* absent_list;
};
static_assert(sizeof(ArenaHead) <= 128, "unit 0 of the arena");
#ifndef SMX_HINT_BUDGET
#define SMX_HINT_BUDGET 8
#endif
constexpr uint32_t HINT_BUDGET = SMX_HINT_BUDGET;        // cells a lane probes before it asks for a hint, when the matrix has a hint table
__device__ inline uint32_t hint_index(uint32_t base, uint32_t Y, uint32_t hmask) {
  return fmix32(base * 0x9E3779B1u ^ Y * 0x85EBCA77u) & hmask;
}
// (round 6) A slot holds TWO entries {tag, cell}: the tag is a second hash of (block, key), never 0.  A tag that matches by
// accident costs a look at a cell that holds another key -- the cell decides, never the tag.  (One entry {key, block, cell} per
// slot until round 6: of the 50 000 gets per dense-id batch that found no hint and walked, 36 000 had lost theirs to another key.)
__device__ inline uint32_t hint_tag(uint32_t base, uint32_t Y) {
  return fmix32(base * 0xC2B2AE35u + Y * 0x27D4EB2Fu + 0x165667B1u) | 1u;
}
// the cell a slot's entries name for `tag` in a table of mask + 1 cells, or 2^32-1
__device__ inline uint32_t hint_way(uint4 e, uint32_t tag, uint32_t mask) {
  return e.x == tag && e.y <= mask ? e.y : e.z == tag && e.w <= mask ? e.w : 0xFFFFFFFFu;
}
// the slot of key Y in the table at `cells` (block `base`, `mask` + 1 cells), or 2^32-1 when no valid hint exists
__device__ inline uint32_t hint_find(const uint8_t* arena, const uint64_t* cells, uint32_t mask, uint32_t Y) {
  const ArenaHead* ah = reinterpret_cast<const ArenaHead*>(arena);
  const uint32_t hmask = ah->hint_mask;
#ifdef SMX_DBG_NO_Y0_GUARD     /* (tests of the tests: the bounded soak's quirk episode must FAIL in a build without the guard) */
  if (hmask == 0 || Y == 0) return 0xFFFFFFFFu;
#else
  if (hmask == 0 || Y == 0 || ah->y0_zeroed) return 0xFFFFFFFFu;
#endif
  const uint32_t base = (uint32_t)((reinterpret_cast<const uint8_t*>(cells) - arena) >> 7);
  const uint32_t p = hint_way(ah->hints[hint_index(base, Y, hmask)], hint_tag(base, Y), mask);
  if (p == 0xFFFFFFFFu) return p;
  return cell_key(cells[p]) == Y ? p : 0xFFFFFFFFu;
}
// (the newest entry in front, the one it pushes back stays, the third is forgotten; racing writers may lose an entry, never
//  invent one: a slot is written with one 16-byte store)
__device__ inline void hint_put(const uint8_t* arena, const uint64_t* cells, uint32_t Y, uint32_t pos) {
  const ArenaHead* ah = reinterpret_cast<const ArenaHead*>(arena);
  const uint32_t hmask = ah->hint_mask;
  if (hmask == 0 || Y == 0) return;
  const uint32_t base = (uint32_t)((reinterpret_cast<const uint8_t*>(cells) - arena) >> 7);
  const uint32_t tag = hint_tag(base, Y);
  uint4* slot = &ah->hints[hint_index(base, Y, hmask)];
  const uint4 e = *slot;
  *slot = e.x == tag ? uint4{tag, pos, e.z, e.w} : uint4{tag, pos, e.x, e.y};
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
#ifndef SMX_FAR_ROW_LG
#define SMX_FAR_ROW_LG 13
#endif
constexpr uint32_t FAR_ROW_LG = SMX_FAR_ROW_LG;           // ... of rows from 8192 cells up: measured 9..18 on the dense-id stream (10.7 / 10.4 / 10.2 / 10.1 / 10.1 ms at 9 / 11 / 12 / 13 / 14, 10.3 at 16, 11.8 at 18)
constexpr uint32_t FAR_UNIT_WORDS = 1u << (FAR_UNIT_LG - 6);
constexpr uint32_t FAR_NOT_FOUND = 0xFFFFFFFFu;
__device__ inline uint32_t far_hash(uint32_t base, uint32_t Y) { return fmix32(base * 0x9E3779B1u + Y * 0x85EBCA77u + 0x27d4eb2fu); }
constexpr uint32_t FAR_BLOOM_LG = 23;
__device__ inline uint32_t far_bloom_bit(uint32_t base, uint32_t Y) { return fmix32(base * 0x85EBCA77u ^ Y * 0xC2B2AE3Du ^ 0x165667B1u) >> (32 - FAR_BLOOM_LG); }
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
__device__ inline bool far_insert(uint4* tab, uint32_t tmask, uint32_t base, uint32_t Y, uint32_t slot, uint32_t* created_at = nullptr) {
  // created_at: the entry's index when THIS call created it, else 2^32-1
  const unsigned long long key = ((unsigned long long)base << 32) | Y;
  uint32_t e = far_hash(base, Y) & tmask;
  if (created_at) *created_at = 0xFFFFFFFFu;
  for (uint32_t guard = 0; guard < 64; guard++) {
    unsigned long long prev = *reinterpret_cast<const unsigned long long*>(&tab[e]);
    if (prev == 0ull) prev = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[e]), 0ull, key);
    if (prev == 0ull || prev == key) { if (prev == 0ull || slot != FAR_NOT_FOUND) tab[e].z = slot; if (prev == 0ull && created_at) *created_at = e; return true; }
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
// CLAIMED inserts of the far join.  The new far keys of a clustered row all walk to the same free cells -- the holes of their run,
// then the cells behind it -- and each insert must see the one before it: 2 000 new keys of one row were 2 000 dependent
// compare-and-swaps on the cell at the front, the pass's critical path.  With the join such a key is known to be absent and the
// free cells of its row are the clear bits of the occupancy words, so an insert CLAIMS its cell there first: walking the words
// from the key's own first free cell on, it takes a RANK in the first word that has free cells left (one fetch-add; the r-th
// claimer owns the r-th cell that was free at the scan -- see the body), sets that cell's bit in the live word and stores the key
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
  unsigned long long* cdbg = reinterpret_cast<const ArenaHead*>(arena)->dbg;     // (measurement runs: where a claimed insert spends its clock ticks; one key in 64)
  if ((fmix32(Y) & 63u) != 0) cdbg = nullptr;
  const long long c_t0 = cdbg ? clock64() : 0;
  uint32_t c_words = 0, c_tries = 0;
  uint32_t* ticket = nullptr;                        // src/smatrix.c:346: insert only while used <= size/2 (as in apply_row)
  if (lg >= BIG_LG) {
    SubCtr* subs = row_subs(arena, s.z, lg);
    const uint32_t k0 = (blockIdx.x * 5u + threadIdx.x) & (SUBS - 1u);
    ticket = sub_ticket(subs + k0);
    if (!ticket) ticket = sub_ticket_elsewhere(subs, k0);
    if (!ticket) {
      *deferred = true;
      if (cdbg) { const long long c_t2 = clock64(); atomicAdd(&cdbg[113], 1ull); atomicAdd(&cdbg[114], (unsigned long long)(c_t2 - c_t0)); atomicMax(&cdbg[115], (unsigned long long)(c_t2 - c_t0)); }
      return 0;
    }
  } else {
    if (s.w > (mask + 1u) / 2u) { *deferred = true; return 0; }
    ticket = &d->used;
    if (atomicAdd(ticket, 1u) > (mask + 1u) / 2u) { atomicSub(ticket, 1u); *deferred = true; return 0; }
  }
  const long long c_t1 = cdbg ? clock64() : 0;
  const auto c_done = [&](uint32_t how) {
    if (!cdbg) return;
    const long long c_t2 = clock64();
    atomicAdd(&cdbg[96], (unsigned long long)(c_t1 - c_t0)); atomicAdd(&cdbg[97], (unsigned long long)(c_t2 - c_t1)); atomicAdd(&cdbg[98], 1ull);
    atomicMax(&cdbg[99], (unsigned long long)(c_t1 - c_t0)); atomicMax(&cdbg[100], (unsigned long long)(c_t2 - c_t1));
    atomicAdd(&cdbg[101], (unsigned long long)c_words); atomicAdd(&cdbg[102], (unsigned long long)c_tries); atomicMax(&cdbg[103], (unsigned long long)c_words);
    atomicAdd(&cdbg[104 + how], 1ull);
    if (c_t2 - c_t0 > 100000) { atomicAdd(&cdbg[108], 1ull); atomicAdd(&cdbg[109], (unsigned long long)(c_t1 - c_t0)); atomicAdd(&cdbg[110], (unsigned long long)(c_t2 - c_t1)); atomicAdd(&cdbg[111], (unsigned long long)c_words); atomicAdd(&cdbg[112], (unsigned long long)c_tries); }
  };
  const uint32_t first = OP == OP_DECR ? 0u - V : V;
  const uint32_t nwords = (mask + 1u) >> 6, wmask = nwords - 1u;
  // (round 5) Which free cell of a word a claimer gets is decided by RANK: one fetch-add on the word's counter, and the r-th
  // claimer owns the r-th cell that was free at the scan (far_occ0) -- one atomic per claimer and word, winners in first-free
  // order.  Going for the lowest clear bit with the atomic OR alone made every claimer of a word try the SAME bit: one winner per
  // round trip, up to 64 round trips per word and claimer.  The new keys of a young dense-id table's hottest rows made 13 attempts
  // on average and 827 at most, each queued behind hundreds of others on one address: single ops of 3-4 ms, which the whole pass
  // waited for.  The first word of a claimer whose front is not that word's first free cell -- the cells below belong to the keys
  // of an earlier run -- keeps the OR protocol, on the bits from its front up: a rank naming a cell the claimer may not take
  // would leave that cell empty for good.  The OR still marks every claimed cell in the live words (the probes read them).
  const ArenaHead* ah = reinterpret_cast<const ArenaHead*>(arena);
  const unsigned long long* occ0 = ah->far_occ0 + (occ - ah->far_occ);
  uint32_t* clm = ah->far_clm + (occ - ah->far_occ);
  uint32_t w = e0 >> 6;
  unsigned long long from = ~0ull << (e0 & 63u);     // (the first word counts from the front's bit only)
  for (uint32_t walked = 0; walked <= nwords + FAR_UNIT_WORDS;) {
    c_words++;
    unsigned long long z = ~__hip_atomic_load(&occ[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & from;
    const unsigned long long z0 = ~occ0[w];
    const bool by_or = (z0 & ~from) != 0;
    const uint32_t nfree0 = (uint32_t)__popcll(z0);
    if (z && (by_or || __hip_atomic_load(&clm[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nfree0)) {
      for (;;) {
        uint32_t b;
        if (by_or) {
          if (!z) break;
          b = (uint32_t)__ffsll(z) - 1u;
        } else {
          const uint32_t rk = atomicAdd(&clm[w], 1u);
          if (rk >= nfree0) break;                    // every cell of this word that was free at the scan has its claimer
          b = select_bit(z0, rk);
        }
        c_tries++;
        const unsigned long long bit = 1ull << b;
        const unsigned long long old = atomicOr(&occ[w], bit);
        z &= ~(old | bit);                            // (what the word really held: the bits others have set since are not tried)
        if (old & bit) continue;                      // somebody else's (a claimer by OR)
        // (round 6) the unit's count of free cells is kept LIVE by the claims: the walk below steps over a unit whose cells
        // have all been handed out since the scan as it steps over one that had none -- the 5 000th new key behind a hot
        // front walked 100 used-up words, one round trip each (30 000 trips of 0.5 M clock ticks per dense-id batch)
        atomicSub(const_cast<uint32_t*>(&zeros[w >> (FAR_UNIT_LG - 6)]), 1u);
        const uint32_t pos = (w << 6) + b;
        const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&cells[pos]), 0ull, (unsigned long long)pack_cell(Y, first));
        if (prev == 0) { *where = pos; c_done(0); return first; }
        if (cell_key(prev) == Y) {                    // (not with one op per key; kept for safety: the cell is updated, the ticket goes back)
          atomicSub(ticket, 1u);
          uint32_t* vp = reinterpret_cast<uint32_t*>(&cells[pos]) + 1;
          *where = pos;
          return OP == OP_INCR ? atomicAdd(vp, V) + V : atomicSub(vp, V) - V;
        }
        // a plain insert took the cell meanwhile (or it holds the row's (0, v) entry): the claim stands for it, on
      }
    }
    from = ~0ull;
    w = (w + 1) & wmask;
    walked++;
    if ((w & (FAR_UNIT_WORDS - 1u)) == 0)             // units without a free cell left (at the scan, or handed out since) are full for good
      while (walked <= nwords + FAR_UNIT_WORDS && __hip_atomic_load(&zeros[w >> (FAR_UNIT_LG - 6)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) { w = (w + FAR_UNIT_WORDS) & wmask; walked += FAR_UNIT_WORDS; }
  }
  atomicSub(ticket, 1u);
  *deferred = true;
  c_done(1);
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
// U: groups of 64 candidates whose cells are loaded together (k_get_clu: 4 -- a walk over 65 000 cells of a big row's run took
// 336 load latencies one after the other, 400 us, and the kernel's last wave is always one of those).
template <int U = 1>
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
        for (uint32_t base = 0; base < total && found == PROBE_NONE; base += 64 * U) {  // wave-uniform
          uint32_t slot[U];
          uint64_t c[U];
          bool have[U];
#pragma unroll
          for (int u = 0; u < U; u++) {
            const uint32_t g = base + (uint32_t)u * 64u + lane;
            have[u] = g < total;
            uint32_t own = 0;                                                // the largest lane whose exclusive prefix is <= g
#pragma unroll
            for (int st = 32; st >= 1; st >>= 1) {
              const uint32_t v = (uint32_t)__shfl((int)excl, (int)(own + st));
              if (v <= g) own += st;
            }
            const uint32_t e_o = (uint32_t)__shfl((int)excl, (int)own);
            const unsigned long long w_o = ((unsigned long long)(uint32_t)__shfl((int)(cand >> 32), (int)own) << 32) | (uint32_t)__shfl((int)(uint32_t)cand, (int)own);
            slot[u] = 0;
            c[u] = ~0ull;
            if (have[u]) {
              slot[u] = ((((w0 + wd + own) & wmask) << 6) | select_bit(w_o, g - e_o)) & mb;
              c[u] = cb[slot[u]];
            }
          }
#pragma unroll
          for (int u = 0; u < U; u++) {
            const uint64_t m = __ballot(have[u] && (cell_key(c[u]) == yb || c[u] == 0));
            if (m && found == PROBE_NONE) found = (uint32_t)__shfl((int)slot[u], __ffsll((unsigned long long)m) - 1);
          }
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
// SHORT: a lane per op that does NOT finish long probes: an op whose probe outruns the lane's budget (and the hint table) is
//        deferred -- the first half of a clustered table's retry, see k_apply_short
// WPK (round 6): ops per wave of a WPO kernel -- lanes 0, 64/WPK, ... hold one each.  What an op costs in such a pass is the
// chain of dependent loads its lane makes alone (directory, the join's table, ticket, claim: ~20 000 clock ticks of 23 500 per
// op, measured), the wave's cooperative probe is the smaller part: WPK lanes make their chains side by side and the wave takes
// their probes one after the other.
#ifndef SMX_WPO_OPS
#define SMX_WPO_OPS 4
#endif
template <int OP, bool WPO = false, int HM = 0, bool FAR = false, bool SHORT = false, int WPK = 1>
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
  constexpr uint32_t OP_LANES = WPO ? 64u / (uint32_t)WPK : 1u;          // lanes per op
  const uint64_t n_lanes = (uint64_t)n * OP_LANES;
  const bool has_hints = HM == 1 || (HM == 2 && reinterpret_cast<const ArenaHead*>(arena)->hint_mask != 0);       // (wave-uniform)
  // (a wave per op with the join at hand: the lane looks at the home cell only -- nine dependent loads of the lane's own probe
  //  were half of such a pass's time; the wave's first window covers them in one load)
  const uint32_t budget = WPO && FAR ? 0u : has_hints ? HINT_BUDGET : PROBE_BUDGET;
  // (clustered matrices: long probes go by the rows' at-home bitmaps, and inserts keep them up to date -- HOME_LG)
  const bool use_home = HM != 0 && reinterpret_cast<const ArenaHead*>(arena)->home_on != 0;                          // (wave-uniform)
  for (uint64_t t064 = (uint64_t)g.bid * blockDim.x; t064 < n_lanes; t064 += (uint64_t)g.nb * blockDim.x) {    // block-uniform
    const uint64_t tl = t064 + threadIdx.x;
    const uint32_t t = (uint32_t)(tl / OP_LANES);
    const bool live = tl < n_lanes && (tl & (OP_LANES - 1u)) == 0;
    uint32_t j = 0, r = 0, Y = 0, V = 0;
    bool deferred = false;
    LongProbe lp{false, nullptr, 0, 0};
    uint4 s = {0, 0, 0, 0};
    DirSlot* d = nullptr;
    // (measurement runs, SMATRIX_REST_DBG: where a wave-per-op pass with the far join spends its cycles)
    unsigned long long* tdbg = WPO && FAR ? reinterpret_cast<const ArenaHead*>(arena)->dbg : nullptr;
    const long long h0r = (WPO && !FAR && reinterpret_cast<const ArenaHead*>(arena)->dbg) ? clock64() : 0;
    long long tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0;
    if (tdbg) tc0 = clock64();
    if (live) {
      j = idx ? idx[t] : t;
      const size_t at = (size_t)j * st;
      Y = ys[at];
      V = OP != OP_GET ? vs[at] : 0u;
      d = dir_find(dir, dmask, xs[at], &s);
      if (!d || s.z == 0) deferred = (OP != OP_GET);      // get on an absent row: 0, creates nothing (S1)
      else if (WPO && FAR && OP != OP_GET && Y != 0) {
        // (the pass with the join: the lane does not even look at the home cell -- the join's table is asked first, and what it
        //  does not know goes to the hint table and the wave's probe FROM the home cell: one dependent load less per op, two for
        //  the keys the join knows.  What apply_row does first for every writer is done here: the row is marked dirty.)
        if (!(s.x & META_DIRTY)) d->meta = s.x | META_DIRTY;
        const uint32_t mask = (1u << meta_lg(s.x)) - 1u;
        lp = LongProbe{true, row_cells(arena, s.z), mask, Y & mask};
      }
      else r = apply_row<OP, true, 1>(d, s, arena, Y, V, Y & ((1u << meta_lg(s.x)) - 1u), &deferred, &lp, false, false, false, nullptr, budget, use_home);
    }
    // a probe that has used up its budget: is the key's cell remembered?  (ArenaHead: dense ids)
    // was_long: the evidence for "this table is clustered" -- a probe of more than PROBE_BUDGET cells, whatever the budget was
    bool was_long = lp.need && !has_hints;
    bool far_none = true;                                   // (WPO && FAR) the join's table does not know this op's key
    const auto ask_hints = [&] {
      const uint32_t p = hint_find(arena, lp.cells, lp.mask, Y);
      if (p != 0xFFFFFFFFu) {
        was_long = ((p - Y) & lp.mask) > PROBE_BUDGET;
        lp.need = false;
        r = apply_row<OP, true, 1>(d, s, arena, Y, V, p, &deferred, &lp);
      }
    };
    if (!(WPO && FAR) && has_hints && lp.need) ask_hints();
    if constexpr (SHORT) {
      if (lp.need) { lp.need = false; deferred = true; was_long = true; }
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
        const FarHit fh = far_find(arena, lp.cells, Y);
        if (ah->dbg && (t & 63u) == 0) atomicAdd(&ah->dbg[16 + fh.state], 64ull);
        far_none = fh.state != FAR_FOUND && fh.state != FAR_ABSENT;
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
            else {
              if (WPO && has_hints) ask_hints();              // (the op that holds the claim may have put the key in already)
              if (lp.need) { lp.need = false; deferred = true; was_long = true; }
            }
          }
          zer = fh.zeros;
        }
      }
    }
    if (WPO && FAR && has_hints && lp.need && far_none) ask_hints();
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
          if (OP != OP_GET && OP != OP_SET && ranked) {
            // the first cell that was free at the scan -- free still, or taken since (by a claim the probe's copy of the word did
            // not show yet, or by a plain insert): either way this key, absent at the scan and inserted by this op alone, goes
            // into the first cell it can CLAIM from here on.  (Round 5: a taken cell used to send the op back to the lane's own
            // walk, 48 dependent loads, then to another cooperative probe: the latest of thousands of new keys behind one run
            // took a hundred such turns -- single trips of 3-4 ms, which the whole pass waited for.)
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
    if (WPO && !FAR) {
      // (measurement runs: the retry's wave-per-op pass -- ops, clock ticks, the trips above 10^5 ticks by row size 2^(4k..))
      unsigned long long* rdbg = reinterpret_cast<const ArenaHead*>(arena)->dbg;
      if (rdbg && live) {
        const long long tt = clock64() - h0r;
        atomicAdd(&rdbg[116], 1ull); atomicAdd(&rdbg[117], (unsigned long long)tt); atomicMax(&rdbg[118], (unsigned long long)tt);
        if (tt > 100000) { atomicAdd(&rdbg[119], 1ull); atomicAdd(&rdbg[120], (unsigned long long)tt); atomicAdd(&rdbg[121 + min(meta_lg(s.x) / 4u, 5u)], 1ull); }
        if (deferred) atomicAdd(&rdbg[127], 1ull);
      }
    }
    if (tdbg) {
      tc3 = clock64();
      if (__lane_id() == 0 && live && tc3 - tc0 > 100000) {   // the long trips: how many, the longest, by row size (2^(4k..)), how it ended
        atomicAdd(&tdbg[28], 1ull); atomicMax(&tdbg[29], (unsigned long long)(tc3 - tc0)); atomicAdd(&tdbg[30], (unsigned long long)(tc3 - tc0));
        atomicAdd(&tdbg[32 + min(meta_lg(s.x) / 4u, 5u)], 1ull);
        atomicAdd(&tdbg[38 + (deferred ? 1 : 0)], 1ull);
        atomicMax(&tdbg[31], (unsigned long long)t_coop);
        atomicAdd(&tdbg[40], (unsigned long long)(tc1 - tc0)); atomicAdd(&tdbg[41], (unsigned long long)(tc2 - tc1)); atomicAdd(&tdbg[42], (unsigned long long)t_coop);
        atomicAdd(&tdbg[43], (unsigned long long)(tc3 - tc2 - t_coop));
      }
      if (__lane_id() == 0 && live && (t & 63u) == 0) {      // (one op in 64: the counters' own atomics must not be what is measured)
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
// wpo (round 5, clustered tables): a WAVE per key -- lane 0 holds it, the wave finishes its long probe, as in k_apply_wpo.  The late
// rounds of a dense-id cold start are a few 10^5 keys of the hottest rows, every one a walk to the end of a long run: a lane per key,
// a wave took its 64 walks one after the other (8, 16, 30, 49 ms for the last four rounds of the stream's first batch).
// mode (round 5, clustered tables): the order in which a batch's new keys go in is the library's to choose (any order is one the
// reference's threads could have taken), and the identity hash rewards ONE order: a key below the table's size sits at home
// whatever else is in, as long as no key that wraps (y >= size) got there first.  So a cold round runs twice over its keys:
//   INS_SMALL_ONLY  keys below their row's size whose home cell is free only (one compare-and-swap, no walk); the others stay listed;
//   INS_HOME_ONLY   then, over what that left (INS_FROM_PREV: ctl->n_prev keys, k_list_advance): any key whose home cell is free -- rows
//                   the first launch filled to their threshold refuse by the snapshot, so only rows that ran out of small keys
//                   take wrapping ones, and first those that need no walk (the hottest row of the dense stream's first batch has
//                   180 000 keys below 2^19 and must hold 262 145 before it may double again: 82 000 of its 120 000 larger keys
//                   go in at that size, and more than half of them find their home cell free);
//                   a key whose home cell holds another key and whose row still has room goes into a list of its own (`kwalk`,
//                   counted in ctl->n_absent): the WALKERS -- a fraction of what a round leaves listed, most of which waits for
//                   its row to double;
//   (neither)       last, over the walkers (k_walk_advance: their number becomes ctl->n_prev), everything admitted: the walks, a
//                   wave per key -- or the batch's far join (insert_pending_keys); what stays deferred joins the round's list.
// A dense-id row then grows through all its doublings with (next to) no displaced cell -- its 10^5 new keys used to go in in list
// order, most of them wrapped onto the run of the keys before them and queued at its end, round after round, and every doubling
// moved them again: rounds 6-10 of the dense stream's first batch took 4.5 + 6.4 + 10.7 + 17.6 + 19.3 ms.
// INS_DROP_EXISTING: nothing is inserted -- every key is looked for (the whole probe, the wave's cooperative one where it is long), a
// key that EXISTS leaves the list, the others stay.  The rounds below and their prep take a listed key for absent (a row at its
// threshold defers without a look, prep grows it without one); that holds for what the op kernels defer, but not for what the
// pass in front of prep of a clustered table leaves: of two ops that name one new far key one claims and inserts it, the other
// waits for the retry -- its key is in the table by the time the cold start reads the list (a row that ended a batch at exactly
// size/2 + 1 keys was doubled for a key it already held: tests/cold_soak.py, seed 12).
constexpr uint32_t INS_SMALL_ONLY = 1u, INS_FROM_PREV = 2u, INS_HOME_ONLY = 4u, INS_DROP_EXISTING = 8u;
__global__ __launch_bounds__(INS_THREADS) void k_insert_keys(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n_host, const unsigned long long* __restrict__ kin,
    unsigned long long* __restrict__ kout, uint32_t wpo, uint32_t mode, unsigned long long* __restrict__ kwalk) {
  __shared__ uint32_t l_row[2 * INS_THREADS], l_cnt[2 * INS_THREADS], l_grant[2 * INS_THREADS];
  __shared__ uint32_t l_n, l_base, l_nw, l_basew;
  const uint32_t n = (mode & INS_FROM_PREV) ? min(n_host, aload(&ctl->n_prev)) : n_host;     // (uniform)
  const uint64_t n_lanes = wpo ? (uint64_t)n * 64u : (uint64_t)n;
  const uint32_t lane_budget = wpo ? 4u : PROBE_BUDGET;
  for (uint64_t t064 = (uint64_t)blockIdx.x * INS_THREADS; t064 < n_lanes; t064 += (uint64_t)gridDim.x * INS_THREADS) {   // block-uniform
    const uint64_t tl = t064 + threadIdx.x;
    const uint32_t t = wpo ? (uint32_t)(tl >> 6) : (uint32_t)tl;
    const bool live = tl < n_lanes && (!wpo || (tl & 63u) == 0);
    for (uint32_t i = threadIdx.x; i < 2 * INS_THREADS; i += INS_THREADS) { l_row[i] = 0xFFFFFFFFu; l_cnt[i] = 0; }
    if (threadIdx.x == 0) { l_n = 0; l_nw = 0; }
    unsigned long long key = 0;
    uint32_t Y = 0, pos = 0, mask = 0, e = 0, rank = 0;
    bool deferred = false, need = false, general = false, walker = false;
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
      else if (Y == 0) general = !(mode & INS_DROP_EXISTING), deferred = (mode & INS_DROP_EXISTING) != 0;
      else if (mode & INS_DROP_EXISTING) {
        mask = (1u << meta_lg(s.x)) - 1u;
        cells = row_cells(arena, s.z);
        pos = Y & mask;
        for (uint32_t steps = 0;; steps++) {
          const uint64_t c = cells[pos];
          if (cell_key(c) == Y) break;                           // it exists: it leaves the list
          if (c == 0) { deferred = true; break; }                // absent: it stays
          if (steps > lane_budget) { lp = LongProbe{true, cells, mask, pos}; break; }
          pos = (pos + 1) & mask;
        }
      }
      else if (meta_lg(s.x) < BIG_LG ? s.w > (1u << meta_lg(s.x)) / 2u
                                     : __hip_atomic_load(&row_subs(arena, s.z, meta_lg(s.x))[0].pad[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        // (round 4) the row stands at the reference's threshold (src/smatrix.c:346) -- the snapshot's count, or a big row's
        // "every share is used up" mark: the key is absent (listed keys are), so it is deferred WITHOUT walking to its empty
        // cell first.  Three quarters of a cold round's keys belong to rows that are waiting for their doubling.
        deferred = true;
      } else if ((mode & INS_SMALL_ONLY) && Y > ((1u << meta_lg(s.x)) - 1u)) {
        deferred = true;                                         // (a key that wraps: the second launch of the round)
      } else {
        mask = (1u << meta_lg(s.x)) - 1u;
        cells = row_cells(arena, s.z);
        if (!(s.x & META_DIRTY)) d->meta = s.x | META_DIRTY;
        pos = Y & mask;
        for (uint32_t steps = 0;; steps++) {
          const uint64_t c = cells[pos];
          if (cell_key(c) == Y) break;                           // it exists: nothing to do
          if (c == 0) { need = true; break; }
          if (mode & (INS_SMALL_ONLY | INS_HOME_ONLY)) {             // (its home cell holds another key: a walk, the last launch's)
            if ((mode & INS_HOME_ONLY) && kwalk) walker = true;      // (... which takes only these: they go into a list of their own, ctl->n_absent counts them)
            else deferred = true;
            break;
          }
          if (steps > lane_budget) { general = true; break; }
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
    {
      // (the host's evidence for "this table is clustered", as in the op kernels: the cold start of a dense-id stream must find
      //  out in its first rounds -- its last four doublings of the hot rows took 8, 16, 30 and 49 ms by priority probing)
      const uint64_t lm = __ballot(lp.need);
      if (lm && __lane_id() == 0) atomicAdd(&ctl->n_long_ops, (uint32_t)__popcll(lm));
    }
    while (__any(lp.need)) {
      const uint32_t p = coop_probe(lp.need, lp.cells, lp.mask, Y, lp.pos, use_home);
      if (lp.need) {
        lp.need = false;
        if (p == PROBE_NONE) deferred = true;
        else if (mode & INS_DROP_EXISTING) deferred = cell_key(lp.cells[p]) != Y;      // (the probe ended on the key, or on an empty cell)
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
    if (kwalk) {                                                 // (uniform)
      const uint64_t wm = __ballot(walker);
      uint32_t wb = 0;
      if (wm && __lane_id() == 0) wb = atomicAdd(&l_nw, (uint32_t)__popcll(wm));
      wb = __shfl(wb, 0);
      __syncthreads();
      if (threadIdx.x == 0 && l_nw) l_basew = atomicAdd(&ctl->n_absent, l_nw);
      __syncthreads();
      if (walker) kwalk[l_basew + wb + (uint32_t)__popcll(wm & ((1ull << __lane_id()) - 1ull))] = key;
    }
    __syncthreads();                                             // the LDS tables are reused by the next trip
  }
}

// The walks of a cold round through the batch's far join (insert_pending_keys): the packed keys are handed to k_apply_wpo_far as ops
// -- x and y are the high and the low word of a key (stride 2), the amounts a zeroed array, the list 0..n-1 -- and the indices it
// leaves deferred are turned back into packed keys for prep and the next round.
__global__ void k_iota(uint32_t* out, uint32_t n);           // (defined below: the list 0..n-1)
__global__ __launch_bounds__(256) void k_gather_keys(const Ctl* ctl, const uint32_t* __restrict__ idx, const unsigned long long* __restrict__ kin,
                                                     unsigned long long* __restrict__ kout, uint32_t n_max) {
  // (the pass appended its indices behind the round's list: entries n_absent .. n_defer of both, see k_walk_advance)
  const uint32_t n0 = aload(&ctl->n_absent), n = min(aload(&ctl->n_defer), n_max);
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x + n0; i < n; i += (uint64_t)gridDim.x * 256) kout[i] = kin[idx[i]];
}

// One representative op per distinct key (x, y != 0) among the listed ops: a scratch hash set of 64-bit keys (zeroed by the
// caller, >= 2 slots per op), the first op to claim a key goes to `reps`.  Representatives are collected in LDS and leave
// with ONE reservation per workgroup and DEDUP_TRIPS x 1024 ops (a reservation per wave queued 10^5 atomics on one word).
constexpr uint32_t DEDUP_THREADS = 1024, DEDUP_TRIPS = 8;
__global__ __launch_bounds__(DEDUP_THREADS) void k_dedup_keys(uint32_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ xs,
                                                             const uint32_t* __restrict__ ys, uint32_t st, unsigned long long* set,
                                                             uint64_t set_mask, unsigned long long* reps, uint32_t* n_reps) {
  // n_reps[1]: how many of the distinct keys have y < n (the length of the list).  Dense ids -- the ids of a batch about as many as
  // its keys -- put nearly all of them there, scrambled or hashed ids one in 2^32 / n: the host's evidence for taking the keys of
  // the cold rounds smallest first (k_insert_keys: mode) before a single long probe has been seen.
  __shared__ unsigned long long l_rep[DEDUP_THREADS * DEDUP_TRIPS];     // (round 4: the distinct keys themselves, x << 32 | y)
  __shared__ uint32_t l_n, l_base, l_small;
  for (uint64_t b0 = (uint64_t)blockIdx.x * DEDUP_THREADS * DEDUP_TRIPS; b0 < n; b0 += (uint64_t)gridDim.x * DEDUP_THREADS * DEDUP_TRIPS) {
    if (threadIdx.x == 0) { l_n = 0; l_small = 0; }
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
      const uint64_t sm = __ballot(won && (uint32_t)key < n);
      uint32_t wb = 0;
      if (sm && __lane_id() == 0) atomicAdd(&l_small, (uint32_t)__popcll(sm));
      if (wm && __lane_id() == 0) wb = atomicAdd(&l_n, (uint32_t)__popcll(wm));
      wb = __shfl(wb, 0);
      if (won) l_rep[wb + (uint32_t)__popcll(wm & ((1ull << __lane_id()) - 1ull))] = key;
    }
    __syncthreads();
    if (threadIdx.x == 0 && l_n) { l_base = atomicAdd(n_reps, l_n); if (l_small) atomicAdd(n_reps + 1, l_small); }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < l_n; i += DEDUP_THREADS) reps[l_base + i] = l_rep[i];
    __syncthreads();
  }
}

// (round 6) GET on a clustered matrix.  With dense ids most ops end at their home cell -- two dependent loads -- and the rest
// walked on cell by cell, asked the hint table, walked with their wave: up to twelve DEPENDENT loads, and every wave waited for
// its few far lanes (0.90 ms per 2^24 gets against 0.40 on scrambled ids).  Here a lane that misses at home asks for the next
// HINT_BUDGET cells AND its hint entry in one trip (the cells share a line or two; nothing depends on the order they arrive in),
// looks through them in probe order, then at the hinted cell: four trips for a far key.  Same answers: the cells are examined
// in the order of the reference's probe (src/smatrix.c:369-377, :299), a hint only names a cell that holds the key.
// (Measured first: the unsettled ops compacted in LDS and run through the generic body by the first wave -- 1.2 ms: a quarter
//  of the ops miss at home, and the second half made all their loads again.)
#ifndef SMX_GET_WALK_U
#define SMX_GET_WALK_U 4
#endif
__global__ __launch_bounds__(256) void k_get_clu(DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n,
                                                 const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys, uint32_t* __restrict__ out, uint32_t st) {
  const ArenaHead* ah = reinterpret_cast<const ArenaHead*>(arena);
  const uint32_t hmask = ah->hint_mask;
  const bool hints_ok = hmask != 0 && !ah->y0_zeroed;                      // (uniform; see hint_find)
  const bool use_home = ah->home_on != 0;
  for (uint64_t t0 = (uint64_t)blockIdx.x * 256u; t0 < n; t0 += (uint64_t)gridDim.x * 256u) {      // block-uniform
    const uint64_t t = t0 + threadIdx.x;
    const bool live = t < n;
    uint32_t r = 0, Y = 0;
    LongProbe lp{false, nullptr, 0, 0};
    if (live) {
      const size_t at = (size_t)t * st;
      Y = ys[at];
      uint4 s;
      DirSlot* d = dir_find(dir, dmask, xs[at], &s);
      if (d && s.z != 0) {                                                   // (get on an absent row: 0, creates nothing, S1)
        const uint32_t mask = (1u << meta_lg(s.x)) - 1u, home = Y & mask;
        const uint64_t* cells = row_cells(arena, s.z);
        const uint64_t c0 = cells[home];
        if (cell_key(c0) == Y) r = cell_val(c0);
        else if (c0 != 0) {
          uint64_t c[HINT_BUDGET];
#pragma unroll
          for (uint32_t i = 0; i < HINT_BUDGET; i++) c[i] = cells[(home + 1u + i) & mask];
          const bool ask = hints_ok && Y != 0;
          uint4 e = {0, 0, 0, 0};
          if (ask) e = ah->hints[hint_index(s.z, Y, hmask)];
          bool settled = false;
#pragma unroll
          for (uint32_t i = 0; i < HINT_BUDGET; i++)
            if (!settled && (cell_key(c[i]) == Y || c[i] == 0)) { settled = true; r = cell_key(c[i]) == Y ? cell_val(c[i]) : 0u; }
          if (const uint32_t hp = !settled && ask ? hint_way(e, hint_tag(s.z, Y), mask) : 0xFFFFFFFFu; hp != 0xFFFFFFFFu) {
            const uint64_t ch = cells[hp];
            if (cell_key(ch) == Y) { settled = true; r = cell_val(ch); }
          }
          if (!settled) lp = LongProbe{true, cells, mask, (home + 1u + HINT_BUDGET) & mask};
        }
      }
    }
#ifdef SMX_GET_WALK_TIMES     /* (tools/probe/get_walkers.py: a walker's result is replaced by how long its walk took) */
    const long long w0 = wall_clock64();
#endif
    // what is left walks with the wave.  (Measured, tools/probe/get_walkers.py: 19 000 walkers per 2^24-get batch of the dense-id
    // stream, 4 us at the median -- but 200 walks above 50 us and some of 400, by cold keys far down the run of a hot row whose
    // hint had been pushed out; the kernel's last wave is always one of them: 0.80 ms against 0.40 with no walker at all.  Two
    // entries per hint slot, four times the slots and four groups of candidates per trip of the walk: 0.54 ms.)
    while (__any(lp.need)) {                                                // (wave-uniform)
      const uint32_t p = coop_probe<SMX_GET_WALK_U>(lp.need, lp.cells, lp.mask, Y, lp.pos, use_home, nullptr);
      if (lp.need) {
        lp.need = false;
        if (p != PROBE_NONE) {
          const uint64_t c = lp.cells[p];
          if (cell_key(c) == Y) { r = cell_val(c); hint_put(arena, lp.cells, Y, p); }   // remembered for the next op that names it
#ifdef SMX_GET_WALK_TIMES
          r = 0xFFF00000u + (uint32_t)min((long long)0xFFFFF, wall_clock64() - w0);
#endif
        }
      }
    }
    if (live) out[t] = r;
  }
}

// The pass in front of prep with the batch's far join at hand: a wave per op, four ops to a wave.  (A LANE per op, every lane walking
// the occupancy words itself, was measured in round 5 -- 3.5-4.6 against 2.6 ms per dense-id batch: a wave with 64 far ops still
// takes their cooperative walks one after the other -- and is gone with its switch.)
#ifndef SMX_WPO_FAR_WAVES
#define SMX_WPO_FAR_WAVES 5     /* 96 VGPRs = 5 waves per SIMD: the pass is bound by the latency of its dependent loads (4 waves 1.80 ms, 5 waves 1.52 ms, 6 waves with spills 1.82 ms) */
#endif
template <int OP>
__global__ __launch_bounds__(256)
#if SMX_WPO_FAR_WAVES
__attribute__((amdgpu_waves_per_eu(SMX_WPO_FAR_WAVES, SMX_WPO_FAR_WAVES)))
#endif
void k_apply_wpo_far(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n, const uint32_t* idx,
    const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
    const uint32_t* __restrict__ vs, uint32_t* __restrict__ out, uint32_t* defer, uint32_t st) {
  apply_body<OP, true, 2, true, false, SMX_WPO_OPS>(SMX_VG, ctl, dir, dmask, arena, n, idx, xs, ys, vs, out, defer, st);
}

// (round 5) The retry of a clustered table in two halves.  After the growth round most ops of the list are short again -- a key
// that was absent from a full row goes into a table that has just doubled -- but a wave per op is priced for the long ones (2.8 ns
// per op: 4 ms for the 1.4 M-op lists of a young dense-id table), and a lane per op finishing its long probes inside the wave makes
// the 63 others wait for each.  So: k_apply_short takes everything a lane's budget and the hint table settle and DEFERS the rest;
// k_apply_wpo then runs over what is left.
template <int OP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(SMX_APPLY_SGPRS))) void k_apply_short(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n, const uint32_t* idx,
    const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
    const uint32_t* __restrict__ vs, uint32_t* __restrict__ out, uint32_t* defer, uint32_t st) {
  apply_body<OP, false, 2, false, true>(SMX_VG, ctl, dir, dmask, arena, n, idx, xs, ys, vs, out, defer, st);
}
// a cold round's walkers become the list the last launch reads; the round's deferred list goes on behind what it holds (n_defer
// stays), and where that was is kept in n_absent for k_gather_keys
__global__ void k_walk_advance(Ctl* ctl) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  ctl->n_prev = ctl->n_absent;
  ctl->n_absent = ctl->n_defer;
}
// between the two: the list the first half wrote becomes the list the second half reads (nothing else of the round's state moves)
__global__ void k_list_advance(Ctl* ctl) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  ctl->n_prev = ctl->n_defer;
  ctl->n_defer = 0;
}

template <int OP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(SMX_APPLY_SGPRS))) void k_apply_wpo(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n, const uint32_t* idx,
    const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
    const uint32_t* __restrict__ vs, uint32_t* __restrict__ out, uint32_t* defer, uint32_t st) {
  apply_body<OP, true, 2, false, false, SMX_WPO_OPS>(SMX_VG, ctl, dir, dmask, arena, n, idx, xs, ys, vs, out, defer, st);
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
#ifndef SMX_AGG_CLU_VGPRS
#define SMX_AGG_CLU_VGPRS 64
#endif
// CLU: the instantiation for clustered tables with a hint table (ArenaHead) -- the slow path asks for the hint after HINT_BUDGET
// cells instead of walking PROBE_BUDGET dependent loads first (a tile waits for its slowest lane)
template <int OP, uint32_t ST, bool RET, bool CLU>     // ST: op stride in words, compile-time here (the kernel has no SGPR to spare); RET: results wanted
__device__ __forceinline__ void apply_agg_body(
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
    // CLU: the 16 cells of one 128-byte line of a row take 16 consecutive slots, and the list below is in slot order (see there)
    uint32_t h = (X * 0x9E3779B1u) ^ ((CLU ? Y >> 4 : Y) * 0x85EBCA77u);
    h = ((h ^ (h >> 15)) + (CLU ? Y & 15u : 0u)) & (AGG_SLOTS - 1);
    for (;;) {
      uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&l_key[h]), ~0ull,
                                (unsigned long long)key);
      if (!CLU && prev == ~0ull) l_list[atomicAdd(&l_n, 1u)] = (uint16_t)h;     // first of its key
      if (prev == ~0ull || prev == key) break;
      h = (h + 1) & (AGG_SLOTS - 1);
    }
    pre[k] = atomicAdd(&l_sum[h], V[k]);
    slot[k] = h;
  }
  __syncthreads();
  if constexpr (CLU) {
    // (round 5) Dense ids put the hottest cells of a hot row side by side: the 16 hottest cells of the hottest row are ONE 128-byte
    // line, and every tile adds to each of them.  Returning atomics on one line are served one REQUEST at a time (tools/probe/
    // atomic_line.hip: 5.7 ns per request from lanes of different waves, 11 ns for a whole wave instruction whose 16 lanes hit 16
    // cells of the line): with the keys in arrival order a tile sent 16 requests to that line, 1.3 x 10^5 per batch, and the kernel
    // took 2.9 ms against 1.7 without its atomics.  In slot order the keys of a line sit in adjacent lanes of phase 2 -- one
    // request per tile and line for the cell loads and for the atomics.
    __shared__ uint32_t l_wave[AGG_THREADS / 64];
    constexpr uint32_t PER = AGG_SLOTS / AGG_THREADS;
    uint32_t occ = 0;
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) occ |= (l_key[tid * PER + q] != ~0ull ? 1u : 0u) << q;
    uint32_t incl = (uint32_t)__popc(occ);
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
      if ((tid & 63u) >= d) incl += o;
    }
    if ((tid & 63u) == 63u) l_wave[tid >> 6] = incl;
    __syncthreads();
    uint32_t at = incl - (uint32_t)__popc(occ);
    for (uint32_t w = 0; w < (tid >> 6); w++) at += l_wave[w];
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) if (occ & (1u << q)) l_list[at++] = (uint16_t)(tid * PER + q);
    if (tid == AGG_THREADS - 1) l_n = at;
    __syncthreads();
  }
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
      bool deferred = false, waits = false;                  // waits: deferred for a structure change, not for a long probe
      if (dbg == 4 && !(fast & (1u << q))) { old[q] = 0; fast |= 1u << q; }     // 4: the slow path (inserts, collisions) left out
      if (full & (1u << q)) {
        deferred = waits = true;                             // the row stands at its threshold: prep doubles it
      } else if (!(fast & (1u << q))) {
        // a probe that outruns the budget (clustered dense ids) is not walked here, one lane at a time: the op is
        // deferred and the lane-per-op kernel finishes it with the wave-cooperative window probe
        LongProbe lp{false, nullptr, 0, 0};
        uint32_t res = apply_one<OP, SMX_AGG_PATIENT, 1>(dir, dmask, arena, (uint32_t)kk[q], (uint32_t)(kk[q] >> 32), tot[q], &deferred, &lp, dbg == 5, !RET,
                                                         false, nullptr, CLU ? HINT_BUDGET : PROBE_BUDGET, CLU);
        waits = deferred;                                    // (no row, no ticket, no empty cell: apply_row)
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
      reinterpret_cast<uint32_t*>(&l_key[hh[q]])[0] = deferred ? (CLU && waits ? 2u : 1u) : 0u;
    }
  }
  __syncthreads();
  // phase 3
  uint32_t dmask_k = 0;          // which of this lane's ops are deferred
  uint32_t amask_k = 0;          // (CLU) ... into the list of the ops that wait for prep (ArenaHead::absent_list)
  uint32_t* alist = nullptr;
  if constexpr (CLU) alist = reinterpret_cast<const ArenaHead*>(arena)->absent_list;
#pragma unroll
  for (uint32_t k = 0; k < AGG_OPT; k++) {
    bool deferred = false;
    if (slot[k] == ~0u - 1) {
      LongProbe lp{false, nullptr, 0, 0};
      uint32_t r = apply_one<OP, false, 1>(dir, dmask, arena, xs[(size_t)j[k] * ST], ys[(size_t)j[k] * ST], V[k], &deferred, &lp);
      if (lp.need) { deferred = true; ctl->n_long = 1; }
      if (!deferred && RET) out[j[k]] = r;
    } else if (slot[k] != ~0u) {
      const uint32_t status = reinterpret_cast<uint32_t*>(&l_key[slot[k]])[0];
      deferred = status != 0;
      if (!deferred && RET) {                                  // (!RET: the caller does not want the results)
        const uint32_t old = l_sum[slot[k]];
        out[j[k]] = OP == OP_INCR ? old + pre[k] + V[k] : old - pre[k] - V[k];
      }
      if (CLU && status == 2u && alist) { amask_k |= 1u << k; deferred = false; }
    }
    if (deferred) dmask_k |= 1u << k;
  }
  // deferred ops: ONE global atomic per workgroup (a per-wave atomic on the single list
  // counter was the kernel's critical path when a few % of the ops defer)
  __syncthreads();                       // everybody is done with l_sum / l_n
  if (tid == 0) l_n = 0;
  uint32_t* l_abs = &l_sum[1];           // (CLU) [0] the tile's ops for the second list, [1] their place in it
  if (CLU && tid == 0) l_abs[0] = 0;
  __syncthreads();
  uint32_t mine = __popc(dmask_k), at = 0;
  if (mine) at = atomicAdd(&l_n, mine);
  uint32_t amine = 0, aat = 0;
  if constexpr (CLU) { amine = __popc(amask_k); if (amine) aat = atomicAdd(&l_abs[0], amine); }
  __syncthreads();
  if (tid == 0 && l_n) l_sum[0] = atomicAdd(&ctl->n_defer, l_n);
  if (CLU && tid == 64 && l_abs[0]) l_abs[1] = atomicAdd(&ctl->n_absent, l_abs[0]);
  __syncthreads();
  if (mine) {
    at += l_sum[0];
#pragma unroll
    for (uint32_t k = 0; k < AGG_OPT; k++)
      if (dmask_k & (1u << k)) defer[at++] = j[k];
  }
  if constexpr (CLU) {
    if (amine) {
      aat += l_abs[1];
#pragma unroll
      for (uint32_t k = 0; k < AGG_OPT; k++)
        if (amask_k & (1u << k)) alist[aat++] = j[k];
    }
  }
}

template <int OP, uint32_t ST = 1, bool RET = true, bool CLU = false>
__global__ __launch_bounds__(AGG_THREADS) __attribute__((amdgpu_num_sgpr(SMX_AGG_SGPRS))) void k_apply_agg(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n, const uint32_t* idx,
    const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
    const uint32_t* __restrict__ vs, uint32_t* __restrict__ out, uint32_t* defer) {
  static_assert(!CLU, "clustered tables: k_apply_agg_clu");
  apply_agg_body<OP, ST, RET, false>(ctl, dir, dmask, arena, n, idx, xs, ys, vs, out, defer);
}
// The clustered instantiation as a kernel of its own, held to 64 VGPRs (it compiled to 68, two spills now): a CU then holds TWO
// 1024-lane workgroups like the scrambled-id kernel does, 2.04 -> 1.66 ms per launch on the dense-id stream.  (A kernel of its
// own because the attribute changes the code of every instantiation it is put on, and the headline's is frozen.)
template <int OP>
__global__ __launch_bounds__(AGG_THREADS) __attribute__((amdgpu_num_sgpr(SMX_AGG_SGPRS))) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_apply_agg_clu(
    Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, uint32_t n, const uint32_t* idx,
    const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
    const uint32_t* __restrict__ vs, uint32_t* __restrict__ out, uint32_t* defer) {
  apply_agg_body<OP, 1, true, true>(ctl, dir, dmask, arena, n, idx, xs, ys, vs, out, defer);
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

// ---- the far join's kernels (see "far join" above) ---------------------------------------------------------------------------
// k_far_rows: every row of >= 2^FAR_ROW_LG cells takes its units (one atomic add: the order does not matter), fills the unit ->
// row map and enters F as {row block, 0} -> first unit.  A row that does not fit the capacities is left out.
// unit_info (round 6): per unit {row block, log2 size << 24 | the unit's place in its row} -- what k_far_scan needs of a unit's row in
// ONE load (it used to go unit -> directory slot -> the row's entry of F before it could ask for a cell); y = 2^32-1: the row did not fit
__global__ __launch_bounds__(256) void k_far_rows(Ctl* ctl, const DirSlot* dir, uint32_t dir_size, uint32_t* unit_row, uint32_t cap_units, uint4* tab,
                                                  uint32_t tmask, uint2* unit_info, uint32_t* big_list, uint32_t big_cap, uint32_t from_list) {
  // big_list (round 6): the directory slots of the rows of >= 2^FAR_ROW_LG cells, [0] = their number.  The pass over the whole
  // directory (64 MB per dense-id batch to find 4 000 rows: 90 us) REBUILDS it; from then on k_grow_commit appends every row that
  // reaches 2^FAR_ROW_LG cells (rows never shrink or go away) and this kernel walks the list (from_list) -- until the directory
  // is rebuilt (its slots move) or a file is loaded.  A row the list misses is only not in the join: its far ops walk as before.
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n_src = from_list ? min(big_list[0], big_cap) : dir_size;
  for (uint32_t h0 = blockIdx.x * blockDim.x; h0 < n_src; h0 += gridDim.x * blockDim.x) {      // (block-uniform)
    const uint32_t i_src = h0 + threadIdx.x;
    const uint32_t h = from_list ? (i_src < n_src ? big_list[1u + i_src] : 0xFFFFFFFFu) : i_src;
    DirSlot d = {0, 0, 0, 0};
    if (h != 0xFFFFFFFFu) d = dir[h];
    const bool big = (d.meta & META_USED) && d.base != 0 && meta_lg(d.meta) >= FAR_ROW_LG;
    if (big && !from_list && big_list) { const uint32_t at = atomicAdd(&big_list[0], 1u); if (at < big_cap) big_list[1u + at] = h; }
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
      const uint32_t f = (uint32_t)__shfl((int)first, src), n = (uint32_t)__shfl((int)units, src), hh = (uint32_t)__shfl((int)h, src);
      const uint32_t rb = (uint32_t)__shfl((int)d.base, src), rlg = (uint32_t)__shfl((int)meta_lg(d.meta), src);
      const bool fits = (uint64_t)f + n <= cap_units;
      for (uint32_t u = lane; u < n && (uint64_t)f + u < cap_units; u += 64) {
        unit_row[f + u] = hh;           // (every unit below the capacity names ITS row)
        unit_info[f + u] = uint2{rb, fits ? (rlg << 24) | u : 0xFFFFFFFFu};
      }
    }
    if (big && (uint64_t)first + units <= cap_units) far_insert(tab, tmask, d.base, 0u, first);
  }
}

// k_far_keys: the deferred ops whose probe outruns the lane's budget on a row of >= 2^HOME_LG cells (what the wave-per-op pass is
// going to find out again: nothing changes in between) enter F.  `limit`: ops beyond it are not entered (the table would fill up).
__global__ __launch_bounds__(256) void k_far_keys(Ctl* ctl, DirSlot* dir, uint32_t dmask, uint8_t* arena, const uint32_t* idx,
                                                  const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys, uint32_t st, uint4* tab,
                                                  uint32_t tmask, uint32_t limit, uint32_t all_far, uint32_t* bloom) {
  // bloom (round 6): one bit per key of F in a table of 2^FAR_BLOOM_LG bits (1 MB: it stays in L2) -- k_far_scan asks it before
  // it looks a displaced cell's key up in F (9 M look-ups in a 64 MB table per dense-id batch, 24 of 25 for keys F does not hold)
  // all_far (the walkers of a cold round: keys known absent whose home cell is taken): every listed key of an indexed row enters F,
  // however short its probe is now -- thousands of new keys of one hot row end their walks on the same few empty cells, and the
  // ones that were not in F took them by compare-and-swap, one winner per cell and turn: 20 000 trips of 1.3 M clock ticks on
  // average (the longest 10 M) in the six passes of the dense stream's first batch.  In F they claim their cell by rank.
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
    for (uint32_t step = 0; step <= HINT_BUDGET && !all_far; step++) {
      const uint64_t c = cells[pos];
      if (cell_key(c) == Y || c == 0) { far = false; break; }
      pos = (pos + 1) & mask;
    }
    if (far) {
      if (!far_insert(tab, tmask, s.z, Y, FAR_NOT_FOUND)) reinterpret_cast<ArenaHead*>(arena)->far_overflow = 1;
      const uint32_t hb = far_bloom_bit(s.z, Y);
      atomicOr(&bloom[hb >> 5], 1u << (hb & 31u));
    }
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
                                                  uint4* tab, uint32_t tmask, unsigned long long* occ, uint32_t* zeros, unsigned long long* occ0, uint32_t* clm,
                                                  uint32_t* rcnt, const uint32_t* bloom, const uint2* unit_info) {
  // rcnt (round 6, k_far_absent / k_far_place): per unit, the absent keys of the row that begins there -- zeroed here
  const uint32_t n_units = min(aload(&ctl->n_units), cap_units);
  const uint32_t lane = threadIdx.x & 63u, nwaves = (gridDim.x * blockDim.x) >> 6;
  if (rcnt && blockIdx.x == 0 && threadIdx.x == 0) rcnt[cap_units] = 0;
  for (uint32_t u = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; u < n_units; u += nwaves) {     // (wave-uniform)
    const uint2 ui = unit_info[u];
    struct { uint32_t base; } d = {ui.x};
    const uint32_t mask = (1u << (ui.y >> 24)) - 1u;
    if (rcnt && lane == 0) rcnt[u] = 0;
    if (ui.y == 0xFFFFFFFFu) { if (lane == 0) zeros[u] = 0xFFFFFFFFu; continue; }      // (a row that did not fit whole has no entry in F: its units are skipped)
    const uint32_t p0 = (ui.y & 0xFFFFFFu) << FAR_UNIT_LG;
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
      if (lane == 0) { occ[(size_t)u * FAR_UNIT_WORDS + q] = m; occ0[(size_t)u * FAR_UNIT_WORDS + q] = m; clm[(size_t)u * FAR_UNIT_WORDS + q] = 0; }
      if (taken && (key & mask) != p) {
        const uint32_t hb = far_bloom_bit(d.base, key);
        if ((bloom[hb >> 5] >> (hb & 31u)) & 1u) {
          uint4* e = far_entry(tab, tmask, d.base, key);
          if (e) e->z = p;
        }
      }
    }
    if (lane == 0) zeros[u] = free_cells;
  }
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
