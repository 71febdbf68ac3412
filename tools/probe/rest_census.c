/*
 * rest_census.c -- CPU census of what k_grow_rest_lds has to place (round 6): the displaced cells of the rows of >= 2^14
 * cells that double in late batches of the dense-id stream, split into the CLUSTERS of the re-insertion.
 *
 * smatrix_rmap_resize re-inserts in old slot order, each cell into the first free slot from its home
 * (/root/reference/src/smatrix.c:392-404).  Cells that sat at home go first (they land at home whatever the others do);
 * number the slots they leave free ("rank space") and let f_i be the rank of the first free slot at/after cell i's home.
 * The occupied SET after any prefix of the cells does not depend on their order; a boundary between two ranks that no
 * cell crosses (#cells with f <= r  ==  #occupied ranks <= r) separates two independent problems.  This tool replays the
 * stream sequentially (like dense_census.c), and at every doubling of a big row after the warm-up reports
 *   - displaced cells, pieces between empty old slots, the longest piece (what one wave places today);
 *   - clusters by size, cells in clusters of one (placed without looking at anybody), the largest cluster.
 * Build: gcc -O2 -o /tmp/rest_census tools/probe/rest_census.c libsmatrix_amd/csrc/smx_stream.c -Iinclude -Ilibsmatrix_amd/csrc -lm
 * Run:   /tmp/rest_census [batches=20] [census_from=16]
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
static uint64_t g_rows, g_disp, g_single, g_hist[24], g_maxcl, g_maxpiece, g_clusters, g_sum_maxcl, g_sum_maxpiece;
static uint64_t g_inorder, g_hard_inorder, g_hard;
static int verbose = 0;

static int lg2(uint64_t v) { int l = 0; while (v >>= 1) l++; return l; }

static void census_row(const row_t* r) {
  const uint32_t os = 1u << r->lg, om = os - 1, ns = os * 2, nm = ns - 1;
  /* B0: at-home cells in the new table */
  uint8_t* b0 = calloc(ns, 1);
  uint32_t nd = 0;
  for (uint32_t p = 0; p < os; p++) {
    const uint32_t key = r->k[p];
    if (key && (key & om) == p) b0[key & nm] = 1;
    else if (key) nd++;
  }
  /* rank of free slots; first free at/after */
  uint32_t* rank = malloc(4 * (ns + 1));            /* free slots before s */
  rank[0] = 0;
  for (uint32_t s = 0; s < ns; s++) rank[s + 1] = rank[s] + !b0[s];
  const uint32_t nfree = rank[ns];
  uint32_t* nxt = malloc(4 * (ns + 1));              /* first free at/after s (no wrap: census only) */
  nxt[ns] = ns;
  for (int64_t s = ns - 1; s >= 0; s--) nxt[s] = b0[s] ? nxt[s + 1] : (uint32_t)s;
  uint32_t* f = malloc(4 * (nd + 1));
  uint32_t* cnt = calloc(nfree + 2, 4);
  uint32_t i = 0, piece = 0, maxpiece = 0;
  for (uint32_t p = 0; p < os; p++) {
    const uint32_t key = r->k[p];
    if (!key) { if (piece > maxpiece) maxpiece = piece; piece = 0; continue; }
    if ((key & om) == p) continue;
    piece++;
    uint32_t t = nxt[key & nm];
    if (t == ns) t = nxt[0];
    f[i] = rank[t];
    cnt[f[i]]++;
    i++;
  }
  if (piece > maxpiece) maxpiece = piece;
  /* clusters in rank space: carry scan */
  uint32_t* cl_of = malloc(4 * (nfree + 1));         /* cluster id of each rank (or ~0) */
  uint32_t carry = 0, ncl = 0, cur = 0, maxcl = 0;
  uint32_t* clsz = calloc(nd + 2, 4);
  for (uint32_t q = 0; q < nfree; q++) {
    const uint32_t have = carry + cnt[q];
    if (have == 0) { cl_of[q] = 0xFFFFFFFFu; continue; }
    if (carry == 0) { cur = ncl++; }
    cl_of[q] = cur;
    clsz[cur]++;
    carry = have - 1;
  }
  uint64_t single = 0;
  for (uint32_t c = 0; c < ncl; c++) {
    g_hist[lg2(clsz[c])]++;
    if (clsz[c] == 1) single++;
    if (clsz[c] > maxcl) maxcl = clsz[c];
  }
  /* how often are the f's of consecutive (time-order) cells of one hard cluster non-decreasing? */
  uint32_t* last_f = malloc(4 * (ncl + 1));
  memset(last_f, 0, 4 * (ncl + 1));
  uint64_t hard = 0, hard_inorder = 0;
  for (uint32_t j = 0; j < nd; j++) {
    const uint32_t c = cl_of[f[j]];
    if (clsz[c] == 1) continue;
    hard++;
    if (f[j] >= last_f[c]) hard_inorder++;
    if (f[j] > last_f[c]) last_f[c] = f[j];
  }
  g_rows++; g_disp += nd; g_single += single; g_clusters += ncl; g_sum_maxcl += maxcl; g_sum_maxpiece += maxpiece;
  g_hard += hard; g_hard_inorder += hard_inorder;
  if (maxcl > g_maxcl) g_maxcl = maxcl;
  if (maxpiece > g_maxpiece) g_maxpiece = maxpiece;
  if (verbose) printf("row lg %u: displaced %u, longest piece %u, clusters %u (single %llu), largest %u, hard %llu (f at/above the running max: %llu)\n",
                      r->lg, nd, maxpiece, ncl, (unsigned long long)single, maxcl, (unsigned long long)hard, (unsigned long long)hard_inorder);
  free(b0); free(rank); free(nxt); free(f); free(cnt); free(cl_of); free(clsz); free(last_f);
}

static void grow(row_t* r) {
  const uint32_t os = 1u << r->lg, ns = os * 2, nm = ns - 1;
  if (census && r->lg >= 14 && r->lg + 1 <= 20) census_row(r);
  uint32_t* nk = calloc(ns, 4);
  for (uint32_t p = 0; p < os; p++) {
    const uint32_t key = r->k[p];
    if (!key) continue;
    uint32_t i = key & nm;
    while (nk[i]) i = (i + 1) & nm;
    nk[i] = key;
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
    while (r->k[i] && r->k[i] != y) i = (i + 1) & m;
    if (r->k[i] == y) return;
    if (r->used > (1u << r->lg) / 2) { grow(r); continue; }
    r->k[i] = y; r->used++;
    return;
  }
}

int main(int argc, char** argv) {
  const int batches = argc > 1 ? atoi(argv[1]) : 20, from = argc > 2 ? atoi(argv[2]) : 16;
  verbose = argc > 3 ? atoi(argv[3]) : 0;
  const size_t B = (size_t)1 << 24;
  smx_stream_t* s = smx_stream_new(SMX_DIST_ZIPF, 12345, 1000000, 1.1, 0);
  rows = calloc(1000001, sizeof(row_t));
  uint32_t* x = malloc(B * 4), *y = malloc(B * 4);
  for (int b = 0; b < batches; b++) {
    smx_stream_fill(s, (uint64_t)b * B, B, x, y);
    census = b >= from;
    for (size_t i = 0; i < B; i++) incr(x[i], y[i]);
    fprintf(stderr, "batch %d done\n", b);
  }
  const double nb = batches - from;
  printf("# dense-id config 2, sequential replay, doublings of rows of 2^14..2^19 cells in batches %d..%d\n", from, batches - 1);
  printf("per batch: %.1f rows, %.0f displaced cells in %.0f clusters; %.0f cells are clusters of one (%.1f %%)\n", g_rows / nb, g_disp / nb,
         g_clusters / nb, g_single / nb, 100.0 * g_single / (g_disp + 1));
  printf("longest piece between empty old slots: %llu cells (mean over rows %.0f); largest cluster %llu cells (mean over rows of the row's largest %.0f)\n",
         (unsigned long long)g_maxpiece, (double)g_sum_maxpiece / g_rows, (unsigned long long)g_maxcl, (double)g_sum_maxcl / g_rows);
  printf("cells in clusters of >= 2: %.0f per batch; of them %.1f %% have f at/above the running maximum of their cluster (in order)\n",
         g_hard / nb, 100.0 * g_hard_inorder / (g_hard + 1));
  printf("clusters by size (log2 bucket: clusters per batch):\n");
  for (int l = 0; l < 24; l++) if (g_hist[l]) printf("    2^%d: %.1f\n", l, g_hist[l] / nb);
  return 0;
}
