/*
 * smatrix_batch.h -- additive batched entry points (no reference counterpart as
 * functions; each applies the reference's per-call semantics, src/smatrix.c:174-256,
 * to n (x,y[,value]) triples at once -- the form the HIP kernels consume).
 *
 * Contract (SURVEY.md 8a "batch semantics"):
 *  - the final table state equals applying the n ops one by one in SOME order the
 *    reference's threads could have produced; row sizes and `used` counters are
 *    exactly the reference's (they do not depend on the order);
 *  - incr/decr: out[i] is the value after op i in that order (exact for unique
 *    keys; for duplicates the largest out equals the final value);
 *  - set: duplicates of one (x,y) inside a batch resolve to the HIGHEST index;
 *    out[i] = v[i];
 *  - a batch of one op, or any stream applied one op per call, reproduces the
 *    reference's table bytes slot for slot.
 *
 * Two flavours: host pointers (staged through HBM by the library; calls above two chunks of 2^21 ops -- SMATRIX_HOST_CHUNK_LG --
 * run as a three-stage pipeline: the caller's arrays are copied into pinned memory by a pool of threads and uploaded while the
 * kernels of the previous chunk run and the results of the one before travel back; a write batch's chunks are applied in order,
 * so the call behaves like ONE batch -- set: the later op still wins; a chunk whose amounts v[] are all one value is filled on
 * the device instead of uploaded.  2^24-op calls: 3.1-3.5 G incr/s, 4.4-5.0 G get/s, PCIe
 * included; this is the shape a JNI / Ruby batch binding has, cf. src/smatrix_jni.c:95-111) and `_dev`
 * (pointers are device memory on the matrix's GPU; work is enqueued on
 * `hip_stream` -- a hipStream_t passed as void*; NULL = the legacy default stream, i.e. what
 * `torch.cuda.current_stream().cuda_stream` is unless the caller switched streams, so inputs
 * produced by earlier work on that stream are ordered before the kernels that read them.
 * Results are complete in stream order: writers return after their last round has
 * finished on that stream, get/rowlen/getrow may return as soon as they are enqueued;
 * with hip_stream == NULL every call synchronises before returning).
 * ORDERING: the library's lock serialises host code only.  All calls on one handle must reach the GPU in one order:
 * use ONE stream per handle, or order the streams yourself (events) -- a get/rowlen/getrow enqueued on stream A
 * and a later write on stream B would otherwise race on the tables (a write may grow rows, recycle their old
 * blocks and replace the directory).  hip_stream == NULL and the host-pointer calls always synchronise.
 * Return value: 0 on success; failures abort like the scalar API.
 * n must be < 2^32.
 */
#ifndef SMATRIX_BATCH_H
#define SMATRIX_BATCH_H

#include "smatrix.h"

#ifdef __cplusplus
extern "C" {
#endif

/* op codes for smatrix_apply_batch* */
enum { SMATRIX_OP_GET = 0, SMATRIX_OP_SET = 1, SMATRIX_OP_INCR = 2, SMATRIX_OP_DECR = 3 };

int smatrix_apply_batch(smatrix_t* self, int op, size_t n, const uint32_t* x,
                        const uint32_t* y, const uint32_t* v, uint32_t* out);
int smatrix_get_batch(smatrix_t* self, size_t n, const uint32_t* x, const uint32_t* y, uint32_t* out);
int smatrix_set_batch(smatrix_t* self, size_t n, const uint32_t* x, const uint32_t* y,
                      const uint32_t* v, uint32_t* out);
int smatrix_incr_batch(smatrix_t* self, size_t n, const uint32_t* x, const uint32_t* y,
                       const uint32_t* v, uint32_t* out);
int smatrix_decr_batch(smatrix_t* self, size_t n, const uint32_t* x, const uint32_t* y,
                       const uint32_t* v, uint32_t* out);
int smatrix_rowlen_batch(smatrix_t* self, size_t n, const uint32_t* x, uint32_t* out);
/* row r writes at most offsets[r+1]-offsets[r] pairs at ret + 2*offsets[r] (uint32 units),
 * in table slot order; counts[r] = pairs written.  offsets has n+1 entries. */
int smatrix_getrow_batch(smatrix_t* self, size_t n, const uint32_t* x, const uint64_t* offsets,
                         uint32_t* ret, uint32_t* counts);

/* device-pointer flavours (d_v is not read by get).  out / d_out may be NULL in every write call when the results are not
 * wanted: the table ends in exactly the same state, the result stores are skipped, and updates of a column-0 cell (the CF
 * example's per-item totals, examples/cf_recommender.c:38) are made with one 64-bit add instead of a compare-and-swap
 * loop -- under heavy contention on a hot item's total that is the difference between 24 ms and 3 ms per 2^25 ops. */
int smatrix_apply_batch_dev(smatrix_t* self, int op, size_t n, const uint32_t* d_x,
                            const uint32_t* d_y, const uint32_t* d_v, uint32_t* d_out,
                            void* hip_stream);

/* The same on ONE device array of n records {x, y} (width 2, get only) or {x, y, v} (width 3): the form
 * in which the sharded exchange delivers a batch (include/smatrix_shard.h) -- no unpacking pass. */
int smatrix_apply_packed_dev(smatrix_t* self, int op, size_t n, const uint32_t* d_records, uint32_t width,
                             uint32_t* d_out, void* hip_stream);
int smatrix_rowlen_batch_dev(smatrix_t* self, size_t n, const uint32_t* d_x, uint32_t* d_out,
                             void* hip_stream);
int smatrix_getrow_batch_dev(smatrix_t* self, size_t n, const uint32_t* d_x,
                             const uint64_t* d_offsets, uint32_t* d_ret, uint32_t* d_counts,
                             void* hip_stream);

/* CF-recommender read path, fused (the reference's documented production use,
 * examples/cf_recommender.c:50-86): for each item a, every neighbour (b, cc) of getrow(a) gets
 *   score = cc / (sqrt(get(a,0)) * sqrt(get(b,0)))      in double,
 * with the example's guards (get(b,0) == 0 -> 1; den == 0 -> 0; cc > den -> 0).  Neighbours come in
 * table slot order; item i writes at most offsets[i+1]-offsets[i] of them at ids/scores + offsets[i];
 * counts[i] = neighbours written.  Column 0 holds the per-item totals, so quirks Q1/Q2 apply. */
int smatrix_cf_neighbors_batch(smatrix_t* self, size_t n, const uint32_t* items, const uint64_t* offsets,
                               uint32_t* ids, double* scores, uint32_t* counts);
int smatrix_cf_neighbors_batch_dev(smatrix_t* self, size_t n, const uint32_t* d_items,
                                   const uint64_t* d_offsets, uint32_t* d_ids, double* d_scores,
                                   uint32_t* d_counts, void* hip_stream);

/* The same read path, but only the k <= 64 BEST neighbours of each item leave the GPU (what a recommender serves):
 * best score first, equal scores in table slot order; the candidates and the score are exactly those above (the
 * (0,total) entry of the row included, as in the example's loop).  ids / scores hold n*k entries, item i's at
 * [i*k, i*k + counts[i]); counts[i] = min(k, entries of the row).  Returns -1 for k == 0 or k > 64. */
int smatrix_cf_topk_batch(smatrix_t* self, size_t n, const uint32_t* items, uint32_t k, uint32_t* ids, double* scores,
                          uint32_t* counts);
int smatrix_cf_topk_batch_dev(smatrix_t* self, size_t n, const uint32_t* d_items, uint32_t k, uint32_t* d_ids,
                              double* d_scores, uint32_t* d_counts, void* hip_stream);

/* CF-recommender write path, on the device (examples/cf_recommender.c:36-47 import_preference_set): session s is
 * ids[offsets[s] .. offsets[s+1]); for every position n of a session  incr(ids[n], 0, 1)  and, for every OTHER position i,
 * incr(ids[n], ids[i], 1) -- L*L ops for a session of L ids, generated on the GPU and applied as incr batches (the
 * batch contract above: any order, same final state).  offsets has n_sessions+1 entries.
 * _dev: all arrays in device memory; d_op_offsets[s] = sum of L*L over the sessions before s (n_sessions+1 entries),
 * total_ops = its last entry. */
int smatrix_cf_import_sessions(smatrix_t* self, size_t n_sessions, const uint64_t* offsets, const uint32_t* ids);
int smatrix_cf_import_sessions_dev(smatrix_t* self, size_t n_sessions, const uint64_t* d_offsets, const uint32_t* d_ids,
                                   const uint64_t* d_op_offsets, uint64_t total_ops, void* hip_stream);

/* Capacity hint, like vector::reserve: map at least `bytes` of device memory for row tables now instead of in growth
 * steps later (each step is a call into the driver, normally ~0.3 ms, but one that can block for seconds while the driver
 * still has freed memory to wipe).  Nothing observable changes.  Returns 0. */
int smatrix_reserve(smatrix_t* self, uint64_t bytes);

/* smatrix_close keeps up to SMATRIX_CHUNK_POOL_GB (default 64, 0 = nothing) of the closed matrix's device memory for the
 * next matrix this process opens (memory handed back to the driver is wiped before reuse, and allocating into that wipe
 * blocks for seconds); this gives it back at once. */
void smatrix_release_cached_memory(void);

/* ---- introspection (tests, bench) ---------------------------------------- */
typedef struct {
  uint64_t rows;            /* rows in the directory */
  uint64_t dir_slots;       /* directory capacity */
  uint64_t arena_units;     /* 128-byte units handed out (incl. retired blocks) */
  uint64_t arena_mapped;    /* bytes of HBM mapped for row tables */
  uint64_t arena_free_units;/* units in retired blocks waiting for reuse */
  uint64_t batches;         /* write batches executed */
  uint64_t rounds;          /* op-kernel rounds over all write batches */
  uint64_t deferred_ops;    /* ops re-run after a structure change */
  uint64_t rows_grown;      /* row doublings */
  uint64_t dir_grown;       /* directory rebuilds */
  uint64_t rows_rebalanced; /* big rows whose insert quotas were re-partitioned */
  uint64_t long_probe_rounds; /* rounds in which ops were handed to the wave-cooperative window probe (clustered ids) */
  uint64_t scalar_cache_hits;    /* scalar-ABI calls answered from the host-side cell mirror (no device round trip) */
  uint64_t scalar_cache_flushes; /* write-backs of mirrored values (one batched set each) */
  uint64_t scalar_cache_flushed_cells;
  uint64_t bulk_rounds;          /* write batches whose deferred ops were grouped by row (the bulk path, k_fix_*) */
  uint64_t bulk_ops;             /* ops finished on that path */
  uint64_t file_flushes;         /* file mode: write-outs of dirty rows (smatrix_flush, SMATRIX_FLUSH_EVERY, close) */
  uint64_t file_rows_written;    /* rows those write-outs wrote (a clean row is never rewritten) */
  uint64_t file_leaked_bytes;    /* row blocks that grown rows left behind in the file since it was opened */
  uint64_t file_compactions;     /* smatrix_compact runs */
  uint64_t spec_chains;          /* write batches whose rounds 0 and 1 were enqueued at once, with one read-back (run_write) */
  uint64_t spec_refused;         /* of those: chains in which a growth task did not fit the estimates and was left to the host-driven loop */
  uint64_t file_bg_flushes;      /* of file_flushes: those the background flusher made (SMATRIX_FLUSH_MS, default 100; 0 = off) */
  uint64_t cold_starts;          /* write batches whose large remainder was reduced to its distinct keys before the rounds went on (insert_pending_keys) */
  uint64_t cold_keys;            /* distinct keys those inserted */
  uint64_t clustered_mode;       /* 1 once a write batch had >= 1/64 of its ops finished by the wave-cooperative probe (unscrambled ids: long runs
                                    under the reference's identity hash): large rows are then doubled in two passes, retry lists run a wave per op */
  uint64_t set_located_by_fold;  /* set batches that round 0 completed: their entries' cells were the ones k_set_fold had found (no locate pass) */
  uint64_t flush_snapshots_refused; /* flushes that could not get their snapshot buffer on the device and wrote under the matrix lock instead */
  /* profiling (smatrix_profile): HIP-event time, launches and ops of the round-0 op kernel,
   * indexed by op code (SMATRIX_OP_GET/SET/INCR/DECR) */
  double   kernel_ms[4];
  uint64_t kernel_launches[4];
  uint64_t kernel_ops[4];
  /* (round 6; new fields are APPENDED from here on -- see smatrix_stats_sz) what the write batches cost the calling thread:
   * wall time inside the write path, of it blocked on the device (read-backs between rounds: kernels were running), of it
   * inside device allocations / frees / address-range maps; call - wait - alloc is host work with the device idle or ahead.
   * Totals since open, then the same for the most recent write batch. */
  double   write_call_ms, write_wait_ms, write_alloc_ms;
  double   last_write_call_ms, last_write_wait_ms, last_write_alloc_ms;
} smatrix_stats_t;

void smatrix_stats(smatrix_t* self, smatrix_stats_t* out);
/* The same for callers that may have been compiled against an older (shorter) version of the struct: at most `size` bytes
 * are written (pass sizeof(smatrix_stats_t) as the caller's header defines it).  Fields are only ever appended. */
void smatrix_stats_sz(smatrix_t* self, smatrix_stats_t* out, size_t size);
/* File mode: writes every row that changed since the last flush to the backing file NOW (dirty rows only -- in
 * place when the row's table still has its on-disk size, else as a fresh block whose CMAP entry is re-pointed, the
 * reference's own scheme, src/smatrix.c:418-496); row blocks first, then the entries that publish them.  smatrix_close
 * does the same one last time.  The reference has no such call: its IO thread flushes continuously (:929-960) and
 * close is its only barrier (:113-133).  Like the reference's IO thread (100 ms poll, :945) a background flusher of this
 * library writes dirty rows every SMATRIX_FLUSH_MS milliseconds (default 100, 0 = off), so a process that dies without
 * close loses about that much.  Both -- this call and the background flusher -- hold the matrix lock only while the dirty
 * rows are collected, laid out and SNAPSHOT on the device (at most SMATRIX_FLUSH_SNAPSHOT_MB = 2048 MB of row tables at a
 * time; a larger backlog goes out in several such steps); the copies to the host and the writes run without it, as the
 * reference's IO thread writes under per-row read locks only (:929-960): callers on other threads keep their latency
 * (measured: a 1 GB flush, batch gets p99 23 -> 25 us).  A row that changes while a flush writes goes out with the next.
 * SMATRIX_FLUSH_EVERY=N checkpoints after every N-th write batch (by the call that made it, once it has released the matrix
 * lock; a host-pointer call that runs in chunks is ONE batch), SMATRIX_FSYNC=1 adds fsync() after the row blocks and after the
 * entries.  Memory mode: no-op.  Returns 0.
 * Lock order (round 6): the FILE lock first, then the matrix lock, everywhere -- smatrix_flush, the background flusher, a
 * SMATRIX_FLUSH_EVERY checkpoint, smatrix_compact, smatrix_close.  A flush keeps the file lock while it writes and holds the
 * matrix lock only for its snapshot; whoever wants the file next queues for it WITHOUT the matrix lock, so no caller of the
 * handle ever stands behind a thread that waits for a write in flight
 * (tests/test_gpu_round6.py::test_a_second_flush_does_not_hold_up_the_callers). */
int smatrix_flush(smatrix_t* self);
/* EXPERIMENTAL, no reference counterpart (the reference's files only grow: resized rows leave their old block behind,
 * src/smatrix.c:430-436, and so do this library's -- same format, same leak): rewrites the backing file without those
 * blocks, all rows into a new file next to it, fsync, rename over the old one.  Needs room for a second copy while it
 * runs.  SMATRIX_COMPACT_AT_CLOSE=1 does it at close.  Not part of the drop-in surface; may change.  Returns 0. */
#ifdef SMATRIX_EXPERIMENTAL
/* (only for callers that define SMATRIX_EXPERIMENTAL, and only active in a process run with SMATRIX_EXPERIMENTAL=1 in its
 *  environment: otherwise the call prints a note, leaves the file alone and returns -1) */
int smatrix_compact(smatrix_t* self);
#endif
/* on: time every round-0 op kernel with HIP events on its stream (adds one sync per
 * batch); resets the kernel_* accumulators. */
void smatrix_profile(smatrix_t* self, int on);
/* returns 1 if the row exists; size = slots, used = rowlen */
int smatrix_row_info(smatrix_t* self, uint32_t x, uint32_t* size, uint32_t* used);
/* copies the row's raw {key,value} slots (slot order); returns its size, 0 if absent */
uint32_t smatrix_row_slots(smatrix_t* self, uint32_t x, uint32_t* kv, uint32_t cap_slots);
/* 1 if a HIP device is usable; the library never falls back to a CPU path */
int smatrix_device_available(void);

#ifdef __cplusplus
}
#endif
#endif
