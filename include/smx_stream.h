/*
 * smx_stream.h -- deterministic synthetic (x,y) streams for the benchmark.
 *
 * Counterpart of the workload loops in the reference's benchmark driver
 * (src/smatrix_benchmark.c:29-65: fixed id blocks) widened to the streams
 * BASELINE.json's configs name: uniform and Zipf(s) ids, optionally scrambled
 * by the murmur3 finaliser.  Specification: SURVEY.md Appendix B.
 *
 *   draw j (0-based) of a stream with seed S:  splitmix64 output number j,
 *       i.e. mix(S + (j+1)*0x9e3779b97f4a7c15)          -- random access
 *   op i uses draw 2i for x and draw 2i+1 for y
 *   uniform id : 1 + r % N
 *   zipf rank  : u = (r >> 11) * 2^-53 ; rank = 1 + min{k : cdf[k] >= u}
 *   scramble   : id = fmix32(rank)   (bijection, 0 -> 0)
 *
 * The CDF is float64, built on the HOST only (increasing-j summation of
 * pow(j,-s)) and handed to the device generator as a table, so host and
 * device streams agree bit for bit.
 */
#ifndef SMX_STREAM_H
#define SMX_STREAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SMX_DIST_CF: the collaborative-filtering shape of BASELINE config 3: op i belongs to row
 * 1 + i / per_row and names a uniform column, i.e. x = id(1 + i / per_row), y = id(1 + draw_i % n_ids)
 * with ONE draw per op (draw i); `zipf_s` carries per_row (ops per row, e.g. 115.0). */
enum { SMX_DIST_UNIFORM = 0, SMX_DIST_ZIPF = 1, SMX_DIST_CF = 2 };

typedef struct smx_stream smx_stream_t;

/* n_ids = N per axis; zipf_s ignored for uniform; scramble: 0 dense ids, 1 fmix32 */
smx_stream_t* smx_stream_new(int dist, uint64_t seed, uint32_t n_ids, double zipf_s, int scramble);
void          smx_stream_free(smx_stream_t* s);

/* ops [first, first+n) of the stream into host arrays */
void smx_stream_fill(const smx_stream_t* s, uint64_t first, size_t n, uint32_t* x, uint32_t* y);

/* same ops into DEVICE arrays (hipMalloc'd), generated on the GPU on `hip_stream`
 * (a hipStream_t passed as void*; NULL = default stream).  Returns 0 on success. */
int smx_stream_fill_device(smx_stream_t* s, uint64_t first, size_t n, uint32_t* d_x,
                           uint32_t* d_y, void* hip_stream);

/* building blocks, exported for the tests */
uint64_t smx_splitmix64_at(uint64_t seed, uint64_t j);
uint32_t smx_fmix32(uint32_t h);
const double* smx_stream_cdf(const smx_stream_t* s, uint32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif
