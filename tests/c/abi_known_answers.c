/* tests/c/abi_known_answers.c -- a plain C caller of the drop-in ABI (include/smatrix.h) and of the additive batch /
 * flush / shard entry points, linked against lib/smatrix.so: the known answers of SURVEY.md A.1 (measured on the
 * unmodified reference) through the eight reference calls, then the batch API on the same handle.
 * Built and run by tests/test_gpu_configs.py::test_c_program_known_answers (gcc is present on the GPU box);
 * tests/test_abi.py compiles it (and every header as C99) on the CPU box.  Exit code 0 = all answers matched. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smatrix.h"
#include "smatrix_batch.h"
#include "smatrix_shard.h"
#include "smx_probe.h"
#include "smx_stream.h"

#define CHECK(expr)                                                        \
  do {                                                                     \
    if (!(expr)) {                                                         \
      printf("FAILED %s (line %d)\n", #expr, __LINE__);                    \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main(int argc, char** argv) {
  const char* file = argc > 1 ? argv[1] : NULL;
  smatrix_t* db = smatrix_open(file);
  uint32_t buf[64], i;
  if (!db) { printf("smatrix_open failed (no HIP device?)\n"); return 2; }
  /* S1: get on an absent cell is 0 and creates nothing */
  CHECK(smatrix_get(db, 7, 7) == 0 && smatrix_rowlen(db, 7) == 0);
  /* S2: new value returned, arithmetic wraps */
  CHECK(smatrix_decr(db, 1, 2, 30) == 4294967266u);
  CHECK(smatrix_incr(db, 1, 2, 5) == 4294967271u);
  CHECK(smatrix_set(db, 1, 2, 17) == 17 && smatrix_get(db, 1, 2) == 17);
  /* Q1: y = 0 is stored but not counted */
  CHECK(smatrix_incr(db, 2, 0, 1) == 1 && smatrix_rowlen(db, 2) == 0);
  CHECK(smatrix_incr(db, 2, 16, 1) == 1 && smatrix_incr(db, 2, 32, 1) == 1);
  CHECK(smatrix_rowlen(db, 2) == 2 && smatrix_get(db, 2, 0) == 1);
  /* S5: growth on the 10th distinct key; getrow in slot order; S4: ret_len in bytes, rounded up to a pair */
  for (i = 1; i <= 12; i++) CHECK(smatrix_incr(db, 3, i, 1) == 1);
  CHECK(smatrix_rowlen(db, 3) == 12);
  memset(buf, 0, sizeof buf);
  CHECK(smatrix_getrow(db, 3, buf, 256) == 12);
  for (i = 0; i < 12; i++) CHECK(buf[2 * i] == i + 1 && buf[2 * i + 1] == 1);
  CHECK(smatrix_getrow(db, 3, buf, 24) == 3 && smatrix_getrow(db, 3, buf, 20) == 3);
  CHECK(smatrix_getrow(db, 999, buf, 64) == 0);
  /* S3: a value-0 cell is still a cell */
  CHECK(smatrix_set(db, 4, 5, 0) == 0 && smatrix_rowlen(db, 4) == 1 && smatrix_get(db, 4, 5) == 0);
  CHECK(db->mem >= 65536u * 16u);                     /* examples/smatrix_example.c:72 reads this field */
  /* the additive batch API on the same handle (host pointers) */
  {
    enum { N = 5000 };
    static uint32_t x[N], y[N], v[N], out[N], len[1];
    uint32_t row = 77;
    for (i = 0; i < N; i++) { x[i] = 77; y[i] = 1 + (i % 1000); v[i] = 2; }
    CHECK(smatrix_incr_batch(db, N, x, y, v, out) == 0);
    CHECK(smatrix_get_batch(db, N, x, y, out) == 0);
    for (i = 0; i < N; i++) CHECK(out[i] == 10);      /* every key five times, +2 each */
    CHECK(smatrix_rowlen_batch(db, 1, &row, len) == 0 && len[0] == 1000);
    CHECK(smatrix_get(db, 77, 1000) == 10 && smatrix_incr(db, 77, 1000, 1) == 11);   /* scalar after batch */
    CHECK(smatrix_get_batch(db, 1, &x[0], &y[999], out) == 0 && out[0] == 11);         /* batch after scalar */
  }
  CHECK(smatrix_shard_of(12345u, 1) == 0);
  CHECK(smatrix_flush(db) == 0);
  smatrix_close(db);
  printf("C_ABI_OK\n");
  return 0;
}
