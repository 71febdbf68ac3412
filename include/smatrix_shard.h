/*
 * smatrix_shard.h -- device-side helpers for the row-hash sharded (multi-GPU) path.
 *
 * No reference counterpart: the reference is one process.  Rows are
 * independent (there is no cross-row operation in src/smatrix.h:87-94), so a node's GPUs
 * each own the rows with smatrix_shard_of(x, nshards) == rank; ops are routed to the owner
 * with one all-to-all and results come back with a second one (libsmatrix_amd/sharded.py).
 * These entry points do the on-GPU part: partition a batch by owner and un-permute results.
 * All pointers named d_* are device memory; hip_stream is a hipStream_t passed as void*.
 */
#ifndef SMATRIX_SHARD_H
#define SMATRIX_SHARD_H

#include <stddef.h>
#include <stdint.h>

#include "smatrix.h"

#ifdef __cplusplus
extern "C" {
#endif

/* hash owner of row x: floor(mix(x) * nshards / 2^32), nshards <= 64 */
uint32_t smatrix_shard_of(uint32_t x, uint32_t nshards);

/* PLACEMENT (optional; planned by the caller, libsmatrix_amd/sharded.py).  Under Zipf the hottest row
 * alone is 12 % of all ops, so equal hash ranges leave its owner with 1.9x the mean load of 8 shards.
 *   d_cuts  : nshards - 1 ascending cut points of the hash space (device memory): shard r owns the rows
 *             with cuts[r-1] <= smatrix_shard_mix(x) < cuts[r] (cuts[-1] = 0, cuts[nshards-1] = 2^32).
 *             NULL = equal ranges = smatrix_shard_of.
 *   d_place : the few hot rows placed one by one: `place_slots` (a power of two <= 1024, or 0 for none)
 *             entries {x, owner + 1} in device memory, open addressing, slot of x =
 *             smatrix_place_slot(x, place_slots), linear probing, owner + 1 == 0 marks an empty slot.
 * A row's owner is its d_place entry if it has one, else the range its hash falls in. */
uint32_t smatrix_shard_mix(uint32_t x);
uint32_t smatrix_place_slot(uint32_t x, uint32_t place_slots);

/* rows held by this matrix whose EQUAL-RANGE hash owner is not `rank` (lets files written without a
 * stored placement be reopened).  Writes up to cap ids to out_x (host); returns how many there are. */
size_t smatrix_displaced_rows(smatrix_t* self, uint32_t rank, uint32_t nshards, uint32_t* out_x, size_t cap);

/* Reorders the n ops shard by shard.  counts_host[s] (host) = ops owned by shard s;
 * d_perm[i] = position of op i in the reordered arrays d_xo/d_yo/d_vo (d_v, d_vo may be NULL);
 * d_work: >= 1024 bytes of device scratch.  Synchronises hip_stream (once).  Returns 0 on success. */
int smatrix_partition_dev(size_t n, const uint32_t* d_x, const uint32_t* d_y, const uint32_t* d_v,
                          uint32_t nshards, uint64_t* counts_host, void* d_work, uint32_t* d_perm,
                          uint32_t* d_xo, uint32_t* d_yo, uint32_t* d_vo, const uint32_t* d_place,
                          uint32_t place_slots, const uint32_t* d_cuts, void* hip_stream);

/* Same, but the reordered ops are written as one record {x,y,v} (d_v != NULL, width 3) or {x,y}
 * (width 2) per op into d_packed, so that a single all-to-all moves them. */
int smatrix_partition_packed_dev(size_t n, const uint32_t* d_x, const uint32_t* d_y, const uint32_t* d_v,
                                 uint32_t nshards, uint64_t* counts_host, void* d_work, uint32_t* d_perm,
                                 uint32_t* d_packed, const uint32_t* d_place, uint32_t place_slots,
                                 const uint32_t* d_cuts, void* hip_stream);

/* records of `width` words -> separate x/y[/v] arrays (the form the op kernels consume) */
int smatrix_unpack_dev(size_t n, uint32_t width, const uint32_t* d_packed, uint32_t* d_x, uint32_t* d_y,
                       uint32_t* d_v, void* hip_stream);

/* d_out[i] = d_src[d_perm[i]] : routes results back into op order */
int smatrix_gather_dev(size_t n, const uint32_t* d_src, const uint32_t* d_perm, uint32_t* d_out,
                       void* hip_stream);

/* ---- the router: the multi-GPU path behind the C boundary -------------------------------------------------
 * One process per GPU.  Every rank opens its shard with the SAME 128-byte id (made once by
 * smatrix_shard_unique_id on rank 0 and handed to the others by whatever launched the processes -- MPI, a file,
 * torch.distributed's store); the library then does the exchange itself: per batch the per-peer counts, the packed
 * {x,y[,v]} records, the results (an alltoallv: grouped ncclSend/ncclRecv over RCCL/xGMI, each peer pair on its own
 * link).  RCCL is loaded at run time (librccl.so.1, or the copy the process already holds); with nranks == 1 it is not
 * needed at all.
 * TRANSPORT.  SMATRIX_SHARD_TRANSPORT=shm replaces RCCL by a host-staged exchange through one POSIX shared-memory
 * object (the id is then its name, "/..."; SMATRIX_SHARD_SHM_MB sizes it, default 256).  It exists so that the N > 1
 * logic can run with several ranks on ONE GPU (tests); RCCL refuses that.
 * All calls on a shard handle are COLLECTIVE: every rank calls them in the same order, each with its own n (0 is
 * fine).  d_* are device pointers on this rank's GPU; work is enqueued on hip_stream (NULL = the legacy default
 * stream, then the call also waits for it).  Results are in the caller's op order.  Semantics of one batch: the
 * batch contract of smatrix_batch.h over the UNION of all ranks' ops (ops of different ranks on one key are
 * "concurrent callers").  Arguments that could differ between ranks (n >= 2^32, NULL arrays, a bad op) are checked
 * before anything is exchanged and take the library's error path (message on stdout, abort): a rank that merely
 * returned an error would leave its peers waiting in the exchange.
 * PLACEMENT.  With more than one rank the first write batch of an EMPTY matrix is also a sample: every rank's 256 most
 * frequent row ids are gathered, the hottest rows are given to shards one by one and the rest hash ranges of unequal
 * width (PLACEMENT above).  Rows never move afterwards: the plan is stored next to the shard file (<file>.placement,
 * the JSON libsmatrix_amd/sharded.py writes) and taken over at reopen; shard files that hold rows away from their
 * equal-range owner but lack that file are refused.  SMATRIX_SHARD_PLACE=0: equal ranges. */
typedef struct smatrix_shard smatrix_shard_t;
#define SMATRIX_SHARD_ID_BYTES 128

int smatrix_shard_unique_id(void* id128);                       /* 0 on success (needs RCCL unless the transport is shm) */
/* fname: this rank's backing file or NULL; returns NULL on failure (no device, RCCL missing, bad rank) */
smatrix_shard_t* smatrix_shard_open(const char* fname, int rank, int nranks, const void* id128);
void smatrix_shard_close(smatrix_shard_t* sh);                  /* collective; closes (and persists) the local shard */
/* this rank's shard as an ordinary handle: rowlen/getrow scans of the rows it owns, stats, smatrix_flush */
smatrix_t* smatrix_shard_local(smatrix_shard_t* sh);
int smatrix_shard_rank(smatrix_shard_t* sh);
int smatrix_shard_nranks(smatrix_shard_t* sh);
uint64_t smatrix_shard_ops_applied(smatrix_shard_t* sh);        /* ops this rank's shard has applied (load balance) */
const char* smatrix_shard_transport(smatrix_shard_t* sh);       /* "self" (one rank), "rccl", "shm" */
/* The RCCL the router uses in this process: the path it is mapped from and ncclGetVersion() (e.g. 22203).  The copy the
 * process already holds wins (a torch process: torch's own librccl), else SMATRIX_RCCL_LIB, librccl.so.1, librccl.so.
 * NULL when no RCCL can be loaded.  Needs no GPU and no open shard. */
const char* smatrix_shard_rccl_library(int* version);

/* a placement chosen by the caller instead (see PLACEMENT above), identical on every rank, before the first batch:
 * cuts: nranks-1 host words or NULL; place_pairs: place_slots x {x, owner+1} host words (open addressing as above) */
int smatrix_shard_set_placement(smatrix_shard_t* sh, const uint32_t* cuts, const uint32_t* place_pairs, uint32_t place_slots);
/* the placement in force: cuts_out (nranks-1 words, *n_cuts = 0 for equal ranges), up to cap_rows {x, owner} pairs of
 * the rows placed one by one (*n_rows = how many there are).  Returns 1 once the placement is settled, else 0. */
int smatrix_shard_get_placement(smatrix_shard_t* sh, uint32_t* cuts_out, uint32_t* n_cuts, uint32_t* place_pairs_out,
                                uint32_t cap_rows, uint32_t* n_rows);

/* the planner and the <file>.placement format without a device or a handle (tests; tools that want to inspect or prepare a
 * placement): n samples {xs[i], counts[i] ops} of a stream of `total` ops -> the plan for `world` shards as JSON in out
 * (cap bytes).  reparse != 0: parsed back and re-serialised first.  Returns the text's length, -1 on error. */
int smatrix_shard_plan_json(const uint32_t* xs, const uint64_t* counts, size_t n, uint64_t total, int world, int reparse,
                            char* out, size_t cap);

/* op = SMATRIX_OP_* (smatrix_batch.h); d_v may be NULL for get */
int smatrix_shard_apply_dev(smatrix_shard_t* sh, int op, size_t n, const uint32_t* d_x, const uint32_t* d_y,
                            const uint32_t* d_v, uint32_t* d_out, void* hip_stream);
/* a write batch and a get on the SAME keys right behind it (the "mixed incr+get" step of the benchmark,
 * src/smatrix_benchmark.c:226-230) with ONE partition and ONE record exchange instead of two */
int smatrix_shard_apply_then_get_dev(smatrix_shard_t* sh, int op, size_t n, const uint32_t* d_x, const uint32_t* d_y,
                                     const uint32_t* d_v, uint32_t* d_out, uint32_t* d_out_get, void* hip_stream);

/* SPLIT PHASES: the same batch in four calls, so that the exchange of one batch runs under the op kernels of another.
 *   h = route(op, ...)        asynchronous: partition + count exchange + record exchange, issued by the library's
 *                             communication thread on its own stream (inputs_ready != 0: the arrays are complete
 *                             already -- else the exchange first waits for what hip_stream holds at this moment)
 *   apply_routed(h, get)      the owner's op kernels on what arrived, on hip_stream (blocks the caller's thread for the
 *                             host-driven rounds of a write batch, like smatrix_apply_batch_dev); then_get != 0: a get
 *                             on the same keys right behind the write
 *   finish(h, out, out_get)   results back and into the caller's order (communication thread; returns when issued)
 *   wait(h, stream)           `stream` waits for the results; h is released (at most 4 batches may be in flight)
 * The communication thread works in call order, and every rank must make these calls in the same order.
 * Threading: a shard handle is driven by ONE caller thread at a time -- "the same order on every rank" has no meaning
 * otherwise.  The blocking calls serialise on the handle's lock; the split phases do not take it (a phase may block while
 * another is issued for the next batch), only the pool of routed handles has a lock of its own.
 * The pipeline of the benchmark:  h1 = route(s+1);  apply_routed(h0);  finish(h0);  wait(h0);  h0 = h1. */
typedef struct smatrix_routed smatrix_routed_t;
smatrix_routed_t* smatrix_shard_route_dev(smatrix_shard_t* sh, int op, size_t n, const uint32_t* d_x, const uint32_t* d_y,
                                          const uint32_t* d_v, int inputs_ready, void* hip_stream);
int smatrix_shard_apply_routed(smatrix_shard_t* sh, smatrix_routed_t* h, int then_get, void* hip_stream);
int smatrix_shard_finish(smatrix_shard_t* sh, smatrix_routed_t* h, uint32_t* d_out, uint32_t* d_out_get);
int smatrix_shard_wait(smatrix_shard_t* sh, smatrix_routed_t* h, void* hip_stream);

/* rowlen / getrow of ARBITRARY rows (src/smatrix.c:212-223 / :189-210, batched): every id is routed to its owner, read
 * there (getrow: in slot order) and sent back.  d_offsets / d_ret / d_counts as in smatrix_getrow_batch_dev: row i may
 * receive d_offsets[i+1] - d_offsets[i] pairs at d_ret + 2 * d_offsets[i], d_counts[i] = pairs written.  A scan of a
 * rank's OWN rows needs no exchange: smatrix_shard_local + smatrix_getrow_batch_dev. */
int smatrix_shard_rowlen_dev(smatrix_shard_t* sh, size_t n, const uint32_t* d_x, uint32_t* d_out, void* hip_stream);
int smatrix_shard_getrow_dev(smatrix_shard_t* sh, size_t n, const uint32_t* d_x, const uint64_t* d_offsets, uint32_t* d_ret,
                             uint32_t* d_counts, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
