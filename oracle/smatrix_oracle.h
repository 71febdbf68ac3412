/*
 * smatrix_oracle.h -- CPU restatement of libsmatrix's (x,y)->uint32 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load or call it, and only as the checker / the CPU baseline.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 *   - against the transcripts of the real reference compiled in this container
 *     (oracle/_ref, recipe in oracle/Makefile), committed as JSON fixtures under tests/golden/
 *     by oracle/gen_golden.py;
 *   - against the known-answer cases of the reference's only test-suite
 *     (src/java/test/TestSparseMatrix.java:22-166) replayed at the C ABI;
 *   - against SURVEY.md Appendix A (slot dumps, stream checksums).
 *
 * Every function cites the reference file:line whose behaviour it restates.
 * All symbols are prefixed ora_ so the library can be loaded next to the real
 * reference (smatrix_*) and next to the product library in one process.
 */
#ifndef SMATRIX_ORACLE_H
#define SMATRIX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_matrix ora_matrix_t;

/* src/smatrix.h:87-94 -- the eight public entry points */
ora_matrix_t* ora_open(const char* fname);
void          ora_close(ora_matrix_t* m);
uint32_t      ora_get(ora_matrix_t* m, uint32_t x, uint32_t y);
uint32_t      ora_set(ora_matrix_t* m, uint32_t x, uint32_t y, uint32_t value);
uint32_t      ora_incr(ora_matrix_t* m, uint32_t x, uint32_t y, uint32_t value);
uint32_t      ora_decr(ora_matrix_t* m, uint32_t x, uint32_t y, uint32_t value);
uint32_t      ora_rowlen(ora_matrix_t* m, uint32_t x);
uint32_t      ora_getrow(ora_matrix_t* m, uint32_t x, uint32_t* ret, size_t ret_len);

/* ---- checker conveniences (no reference counterpart) -------------------- */

/* op codes shared with the product's batch API and the golden fixtures */
enum { ORA_OP_GET = 0, ORA_OP_SET = 1, ORA_OP_INCR = 2, ORA_OP_DECR = 3 };

/* apply n ops of one kind in index order; out (may be NULL) gets each return */
void ora_apply(ora_matrix_t* m, int op, size_t n, const uint32_t* x,
               const uint32_t* y, const uint32_t* v, uint32_t* out);

/* sum of get(x_i, y_i) over a stream (SURVEY.md A.4 checksum) */
uint64_t ora_sum_get(ora_matrix_t* m, size_t n, const uint32_t* x, const uint32_t* y);

/* CF-recommender read path as the example intends it (examples/cf_recommender.c:50-86): neighbours of
 * `item` in getrow (slot) order with their cosine scores; returns the count (<= cap) */
uint32_t ora_cf_neighbors(ora_matrix_t* m, uint32_t item, uint32_t* ids, double* scores, uint32_t cap);
/* examples/cf_recommender.c:36-47 */
void ora_cf_import_preference_set(ora_matrix_t* m, const uint32_t* ids, uint32_t num_ids);

/* introspection: row table geometry and raw slots (slot order) */
uint64_t ora_num_rows(ora_matrix_t* m);              /* cmap.used  */
uint64_t ora_dir_size(ora_matrix_t* m);              /* cmap.size  */
uint64_t ora_mem(ora_matrix_t* m);                   /* self->mem  */
uint64_t ora_nnz(ora_matrix_t* m);                   /* non-empty slots over all rows */
/* returns 1 if the row exists; fills size/used */
int      ora_row_info(ora_matrix_t* m, uint32_t x, uint32_t* size, uint32_t* used);
/* copies min(size, cap_slots) raw {key,value} slots; returns the row's size (0 if absent) */
uint32_t ora_row_slots(ora_matrix_t* m, uint32_t x, uint32_t* kv, uint32_t cap_slots);
/* lists row ids in directory slot order; returns count written (<= cap) */
uint64_t ora_list_rows(ora_matrix_t* m, uint32_t* xs, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
