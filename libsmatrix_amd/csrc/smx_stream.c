/*
 * smx_stream.c -- host side of the synthetic stream generator (include/smx_stream.h).
 * Pure C, no HIP: usable on a box without a GPU.  The device generator in
 * smx_kernels.hip consumes the CDF table built here.
 */
#include "smx_stream_priv.h"

#include <math.h>
#include <stdlib.h>

static uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

uint64_t smx_splitmix64_at(uint64_t seed, uint64_t j) {
  return mix64(seed + (j + 1) * 0x9e3779b97f4a7c15ULL);
}

uint32_t smx_fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bU;
  h ^= h >> 13;
  h *= 0xc2b2ae35U;
  h ^= h >> 16;
  return h;
}

smx_stream_t* smx_stream_new(int dist, uint64_t seed, uint32_t n_ids, double zipf_s, int scramble) {
  if (n_ids == 0) return NULL;
  if (dist == SMX_DIST_CF && !(zipf_s >= 1.0)) return NULL;      /* ops per row */
  smx_stream_t* s = calloc(1, sizeof *s);
  if (!s) return NULL;
  s->dist = dist;
  s->seed = seed;
  s->n_ids = n_ids;
  s->zipf_s = zipf_s;
  s->scramble = scramble;
  if (dist == SMX_DIST_ZIPF) {
    s->cdf = malloc((size_t)n_ids * sizeof(double));
    if (!s->cdf) { free(s); return NULL; }
    double total = 0.0;
    for (uint32_t j = 1; j <= n_ids; j++) total += pow((double)j, -zipf_s);
    double run = 0.0;
    for (uint32_t j = 1; j <= n_ids; j++) {
      run += pow((double)j, -zipf_s);
      s->cdf[j - 1] = run / total;
    }
  }
  return s;
}

void smx_stream_free(smx_stream_t* s) {
  if (!s) return;
  smx_stream_release_device(s);
  free(s->cdf);
  free(s);
}

const double* smx_stream_cdf(const smx_stream_t* s, uint32_t* n_out) {
  if (n_out) *n_out = s->n_ids;
  return s->cdf;
}

static uint32_t draw_id(const smx_stream_t* s, uint64_t r) {
  uint32_t id;
  if (s->dist == SMX_DIST_UNIFORM) {
    id = 1u + (uint32_t)(r % s->n_ids);
  } else {
    double u = (double)(r >> 11) * 0x1.0p-53;
    uint32_t lo = 0, hi = s->n_ids - 1;      /* smallest k with cdf[k] >= u */
    while (lo < hi) {
      uint32_t mid = lo + (hi - lo) / 2;
      if (s->cdf[mid] < u) lo = mid + 1; else hi = mid;
    }
    id = lo + 1;
  }
  return s->scramble ? smx_fmix32(id) : id;
}

void smx_stream_fill(const smx_stream_t* s, uint64_t first, size_t n, uint32_t* x, uint32_t* y) {
  if (s->dist == SMX_DIST_CF) {
    const uint64_t per_row = (uint64_t)s->zipf_s;
    for (size_t i = 0; i < n; i++) {
      uint64_t op = first + i;
      uint32_t row = 1u + (uint32_t)(op / per_row);
      uint32_t col = 1u + (uint32_t)(smx_splitmix64_at(s->seed, op) % s->n_ids);
      x[i] = s->scramble ? smx_fmix32(row) : row;
      y[i] = s->scramble ? smx_fmix32(col) : col;
    }
    return;
  }
  for (size_t i = 0; i < n; i++) {
    uint64_t op = first + i;
    x[i] = draw_id(s, smx_splitmix64_at(s->seed, 2 * op));
    y[i] = draw_id(s, smx_splitmix64_at(s->seed, 2 * op + 1));
  }
}
