/*
 * smatrix.h -- drop-in C ABI of the MI355X-native libsmatrix hot path.
 *
 * The eight entry points below are exactly the reference's public API
 * (/root/reference/src/smatrix.h:87-94) -- same names, argument meaning, return
 * values and error behaviour -- so that the reference's JNI glue
 * (src/smatrix_jni.c:59-158) and Ruby glue (src/smatrix_ruby.c:33-163) link
 * against this library unchanged.  Behaviour restated per function:
 *
 *   smatrix_open   src/smatrix.c:74-111   NULL fname = memory only; else open or
 *                                         create the file (reference file format)
 *   smatrix_close  src/smatrix.c:113-133  flush barrier in file mode, frees handle
 *   smatrix_get    src/smatrix.c:174-185  value or 0; never creates
 *   smatrix_set    src/smatrix.c:225-234  returns the new value
 *   smatrix_incr   src/smatrix.c:236-245  returns the new value, wraps mod 2^32
 *   smatrix_decr   src/smatrix.c:247-256  returns the new value, wraps mod 2^32
 *   smatrix_rowlen src/smatrix.c:212-223  the row's `used` counter, 0 if absent
 *   smatrix_getrow src/smatrix.c:189-210  [key,value] pairs in table slot order;
 *                                         ret_len counts BYTES; stops once
 *                                         pairs*8 >= ret_len
 *
 * All six data calls plus rowlen/getrow may be called concurrently from any
 * threads on one handle (README.md:113,120 of the reference).
 * Errors: smatrix_open returns NULL (after a message on stderr); any other
 * failure prints "libsmatrix error: ..." on stdout and abort()s
 * (src/smatrix.c:891-894).
 *
 * The handle's leading fields mirror the reference's smatrix_t
 * (src/smatrix.h:76-85) far enough for code that peeks at `mem`
 * (examples/smatrix_example.c:72); everything else lives behind `impl`.
 * A scalar call on a cell the library has not seen yet costs one device round
 * trip; calls on cells it has seen are answered from a host-side mirror and
 * written back before anything else looks at the tables (DESIGN.md "scalar ABI").
 * Bulk throughput comes from the additive batched API in smatrix_batch.h.
 */
#ifndef SMATRIX_H
#define SMATRIX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int      fd;        /* 0 = memory mode (reference: tested as truthiness) */
  int      shutdown;
  uint64_t fpos;      /* logical end of the backing file */
  uint64_t mem;       /* bytes of HBM in use by the tables */
  void*    impl;
} smatrix_t;

smatrix_t* smatrix_open(const char* fname);
uint32_t smatrix_get(smatrix_t* self, uint32_t x, uint32_t y);
uint32_t smatrix_set(smatrix_t* self, uint32_t x, uint32_t y, uint32_t value);
uint32_t smatrix_incr(smatrix_t* self, uint32_t x, uint32_t y, uint32_t value);
uint32_t smatrix_decr(smatrix_t* self, uint32_t x, uint32_t y, uint32_t value);
uint32_t smatrix_rowlen(smatrix_t* self, uint32_t x);
uint32_t smatrix_getrow(smatrix_t* self, uint32_t x, uint32_t* ret, size_t ret_len);
void smatrix_close(smatrix_t* self);

#ifdef __cplusplus
}
#endif
#endif
