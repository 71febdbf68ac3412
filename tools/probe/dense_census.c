/*
 * dense_census.c -- CPU census of the dense-id variant of the config-2 stream (ids = Zipf ranks, unscrambled).
 *
 * Replays the stream one op at a time on identity-hashed, linearly probed row tables with the reference's growth rule
 * (insert only while used <= size/2, else double and re-insert in old slot order: /root/reference/src/smatrix.c:343-416)
 * and counts, for the batches after a warm-up:
 *   - per op: cells walked from home, how many of them were AT HOME (slot == key mod size) and how many displaced;
 *   - per doubling of a row of >= 2^14 cells: displaced cells, their walks in the new table split the same way, and
 *     the 64-cell mask words a walk crosses.
 * It answers what a persistent at-home bitmap can save before any kernel is written.
 * Build: gcc -O2 -o /tmp/dense_census tools/probe/dense_census.c libsmatrix_amd/csrc/smx_stream.c -Iinclude -Ilibsmatrix_amd/csrc -lm
 * Run:   /tmp/dense_census [batches=20] [census_from=16] [log2 batch=24]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "smx_stream.h"

void smx_stream_release_device(smx_stream_t* s) { (void)s; }

typedef struct { uint32_t* k; uint32_t lg, used; } row_t;
static row_t* rows;
static int census = 0;

/* growth census (rows of >= 2^14 old cells) */
static uint64_t g_rows, g_cells, g_disp, g_walk_home, g_walk_disp, g_words, g_maxpile;
static uint64_t g_hist_disp[24];
/* op census */
static uint64_t o_far, o_far_ins, o_walk, o_walk_home, o_walk_disp, o_words, o_all_walk;
static uint64_t o_hist[24], o_hist_disp[24];
static uint64_t o_far_big;

static int lg2(uint64_t v) { int l = 0; while (v >>= 1) l++; return l; }

static void grow(row_t* r) {
  const uint32_t os = 1u << r->lg, ns = os * 2, nm = ns - 1;
  uint32_t* nk = calloc(ns, 4);
  const int big = census && r->lg >= 14;
  if (big) { g_rows++; g_cells += os; }
  for (uint32_t p = 0; p < os; p++) {
    const uint32_t key = r->k[p];
    if (!key) continue;
    uint32_t i = key & nm;
    const int was_home = (key & (os - 1)) == p;
    uint64_t wh = 0, wd = 0;
    const uint32_t i0 = i;
    while (nk[i]) {
      if ((nk[i] & nm) == i) wh++; else wd++;
      i = (i + 1) & nm;
    }
    nk[i] = key;
    if (big && !was_home) {
      g_disp++;
      g_walk_home += wh; g_walk_disp += wd;
      g_words += (((i - i0) & nm) >> 6) + 1;
      g_hist_disp[lg2(wd + 1)]++;
      if (wd > g_maxpile) g_maxpile = wd;
    }
  }
  free(r->k);
  r->k = nk;
  r->lg++;
}

static void incr(uint32_t x, uint32_t y) {
  row_t* r = &rows[x];
  if (!r->k) { r->k = calloc(16, 4); r->lg = 4; r->used = 0; }
  for (int pass = 0; pass < 2; pass++) {
    const uint32_t m = (1u << r->lg) - 1;
    uint32_t i = y & m;
    uint64_t wh = 0, wd = 0;
    while (r->k[i] && r->k[i] != y) {
      if (census && pass == 0) { if ((r->k[i] & m) == i) wh++; else wd++; }
      i = (i + 1) & m;
    }
    const int hit = r->k[i] == y;
    if (census && pass == 0) {
      const uint64_t w = wh + wd;
      o_all_walk += w;
      if (w > 48) {
        o_far++; if (!hit) o_far_ins++;
        if (r->lg >= 15) o_far_big++;
        o_walk += w; o_walk_home += wh; o_walk_disp += wd;
        o_words += (w >> 6) + 1;
        o_hist[lg2(w)]++; o_hist_disp[lg2(wd + 1)]++;
      }
    }
    if (hit) return;
    if (r->used > (1u << r->lg) / 2) { grow(r); continue; }
    r->k[i] = y; r->used++;
    return;
  }
}

int main(int argc, char** argv) {
  const int batches = argc > 1 ? atoi(argv[1]) : 20, from = argc > 2 ? atoi(argv[2]) : 16, blg = argc > 3 ? atoi(argv[3]) : 24;
  const size_t B = (size_t)1 << blg;
  smx_stream_t* s = smx_stream_new(SMX_DIST_ZIPF, 12345, 1000000, 1.1, 0);
  rows = calloc(1000001, sizeof(row_t));
  uint32_t* x = malloc(B * 4), *y = malloc(B * 4);
  for (int b = 0; b < batches; b++) {
    smx_stream_fill(s, (uint64_t)b * B, B, x, y);
    census = b >= from;
    for (size_t i = 0; i < B; i++) incr(x[i], y[i]);
    fprintf(stderr, "batch %d done\n", b);
  }
  const int nb = batches - from;
  printf("# dense-id config 2, sequential replay, census over batches %d..%d (per batch of 2^%d ops)\n", from, batches - 1, blg);
  printf("ops more than 48 cells from home: %.0f per batch (%.0f inserts, %.0f in rows >= 2^15); all ops walk %.3g cells per batch\n",
         (double)o_far / nb, (double)o_far_ins / nb, (double)o_far_big / nb, (double)o_all_walk / nb);
  printf("  cells they walk: %.3g per batch = %.3g at home (%.1f %%) + %.3g displaced (%.1f %%); 64-cell words crossed %.3g\n",
         (double)o_walk / nb, (double)o_walk_home / nb, 100.0 * o_walk_home / (o_walk + 1), (double)o_walk_disp / nb,
         100.0 * o_walk_disp / (o_walk + 1), (double)o_words / nb);
  printf("  walk length (log2 bucket: ops per batch | by DISPLACED cells walked):\n");
  for (int l = 5; l < 22; l++) printf("    2^%d: %.0f | %.0f\n", l, (double)o_hist[l] / nb, (double)o_hist_disp[l] / nb);
  printf("doublings of rows >= 2^14 cells: %.1f per batch, %.3g old cells, %.0f displaced cells moved\n", (double)g_rows / nb,
         (double)g_cells / nb, (double)g_disp / nb);
  printf("  their walks in the new table: %.3g cells at home + %.3g displaced residents per batch; words crossed %.3g; longest walk over displaced %llu\n",
         (double)g_walk_home / nb, (double)g_walk_disp / nb, (double)g_words / nb, (unsigned long long)g_maxpile);
  printf("  displaced residents walked per moved cell (log2 bucket: cells per batch):\n");
  for (int l = 0; l < 20; l++) if (g_hist_disp[l]) printf("    2^%d: %.0f\n", l, (double)g_hist_disp[l] / nb);
  /* the table as it stands: rows by size, at-home share */
  uint64_t nrows[32] = {0}, cells_home[32] = {0}, cells_disp[32] = {0};
  for (uint32_t r = 1; r <= 1000000; r++) {
    if (!rows[r].k) continue;
    const uint32_t m = (1u << rows[r].lg) - 1;
    nrows[rows[r].lg]++;
    for (uint32_t i = 0; i <= m; i++) if (rows[r].k[i]) { if ((rows[r].k[i] & m) == i) cells_home[rows[r].lg]++; else cells_disp[rows[r].lg]++; }
  }
  printf("rows by size at the end (log2 size: rows, cells at home, displaced):\n");
  for (int l = 4; l < 32; l++) if (nrows[l]) printf("    2^%d: %llu rows, %llu at home, %llu displaced\n", l, (unsigned long long)nrows[l],
                                                    (unsigned long long)cells_home[l], (unsigned long long)cells_disp[l]);
  return 0;
}
