/*
 * smatrix_oracle.c -- single-threaded CPU restatement of libsmatrix's
 * get/set/incr/decr/getrow/rowlen path and of its file format.
 *
 * TEST INFRASTRUCTURE ONLY (see smatrix_oracle.h).  Parity: PINNED against the
 * compiled reference (oracle/_ref) through tests/golden/ and the live
 * differential tests in tests/test_oracle_vs_ref.py.
 *
 * The reference keeps two levels of open-addressing tables with the identity
 * hash `key % size` and linear probing (src/smatrix.c:366,376 and :677,689):
 *   directory ("cmap", src/smatrix.h:51-65): x -> row, occupancy by flag bit
 *   row table ("rmap", src/smatrix.h:35-49): y -> value, 8-byte {key,value}
 *       slots, a slot is EMPTY iff key==0 && value==0 (src/smatrix.c:373)
 * The reference's locks (src/smatrix.c:843-889) have no single-threaded
 * observable effect and are not restated.
 */
#define _GNU_SOURCE
#include "smatrix_oracle.h"

#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

/* constants: src/smatrix.h:15-28 */
#define ROW_FIRST_SIZE      16u        /* SMATRIX_RMAP_INITIAL_SIZE */
#define DIR_FIRST_SIZE      65536u     /* SMATRIX_CMAP_INITIAL_SIZE */
#define FILE_HEADER_BYTES   512u       /* SMATRIX_META_SIZE         */
#define ROWBLK_HEAD_BYTES   16u        /* SMATRIX_RMAP_HEAD_SIZE    */
#define DIRBLK_HEAD_BYTES   16u        /* SMATRIX_CMAP_HEAD_SIZE    */
#define DIRBLK_ENTRY_BYTES  12u        /* SMATRIX_CMAP_SLOT_SIZE    */
#define DIRBLK_ENTRIES      4194304u   /* SMATRIX_CMAP_BLOCK_SIZE   */
/* in-memory footprints the reference charges to self->mem ([probe] SURVEY A.1) */
#define REF_SIZEOF_ROWHDR   48u        /* sizeof(smatrix_rmap_t)      */
#define REF_SIZEOF_DIRSLOT  16u        /* sizeof(smatrix_cmap_slot_t) */

typedef struct { uint32_t key, value; } cell_t;

typedef struct row {
  uint32_t x;
  uint32_t size;   /* slots */
  uint32_t used;   /* the reference's rmap->used (NOT always the non-empty count) */
  cell_t*  cells;
} row_t;

typedef struct { uint32_t occupied; uint32_t key; row_t* row; } dslot_t;

struct ora_matrix {
  uint64_t  dir_size, dir_used;
  dslot_t*  dir;
  row_t**   order;          /* rows in creation order (file: CMAP entry order) */
  uint64_t  order_len, order_cap;
  uint64_t  mem;            /* mirrors self->mem, src/smatrix.c:151-166 */
  char*     fname;          /* NULL = memory only */
};

static void die(const char* msg) {
  /* src/smatrix.c:891-894: message on stdout, then abort */
  printf("libsmatrix error: %s", msg);
  abort();
}

static void* xcalloc(size_t n, size_t sz) {
  void* p = calloc(n ? n : 1, sz);
  if (!p) die("malloc() failed");
  return p;
}

/* ---- row table ---------------------------------------------------------- */

static int cell_is_empty(const cell_t* c) { return c->key == 0 && c->value == 0; }

/* src/smatrix.c:363-380 smatrix_rmap_probe: first slot from key%size whose key
 * equals `key` or that is empty; after a full cycle, wherever it stands. */
static cell_t* row_probe(row_t* r, uint32_t key) {
  uint64_t pos = key % r->size;
  for (uint64_t n = 0; n < r->size; n++) {
    cell_t* c = &r->cells[pos];
    if (c->key == key || cell_is_empty(c)) break;
    pos = (pos + 1) % r->size;
  }
  return &r->cells[pos];
}

static cell_t* row_insert(ora_matrix_t* m, row_t* r, uint32_t key);

/* src/smatrix.c:383-416 smatrix_rmap_resize: double, re-insert every non-empty
 * slot in old slot order THROUGH row_insert (which recounts `used`, Q2). */
static void row_grow(ora_matrix_t* m, row_t* r) {
  row_t bigger;
  bigger.x = r->x;
  bigger.size = r->size * 2;
  bigger.used = 0;
  bigger.cells = xcalloc(bigger.size, sizeof(cell_t));
  m->mem += (uint64_t)bigger.size * sizeof(cell_t);
  for (uint32_t p = 0; p < r->size; p++) {
    if (cell_is_empty(&r->cells[p])) continue;
    cell_t* c = row_insert(m, &bigger, r->cells[p].key);
    c->value = r->cells[p].value;
  }
  m->mem -= (uint64_t)r->size * sizeof(cell_t);
  free(r->cells);
  r->cells = bigger.cells;
  r->size = bigger.size;
  r->used = bigger.used;
}

/* src/smatrix.c:343-360 smatrix_rmap_insert: growth test BEFORE the insert
 * (used > size/2), then claim the probed slot unless it already holds a
 * non-zero key equal to `key`. */
static cell_t* row_insert(ora_matrix_t* m, row_t* r, uint32_t key) {
  if (r->used > r->size / 2) row_grow(m, r);
  cell_t* c = row_probe(r, key);
  if (c->key == 0 || c->key != key) {
    r->used++;
    c->key = key;
    c->value = 0;
  }
  return c;
}

static row_t* row_new(ora_matrix_t* m, uint32_t x, uint32_t size) {
  /* src/smatrix.c:641-643 + :323-339 */
  row_t* r = xcalloc(1, sizeof(row_t));
  r->x = x;
  r->size = size;
  r->used = 0;
  r->cells = size ? xcalloc(size, sizeof(cell_t)) : NULL;
  m->mem += REF_SIZEOF_ROWHDR + (uint64_t)size * sizeof(cell_t);
  if (m->order_len == m->order_cap) {
    m->order_cap = m->order_cap ? m->order_cap * 2 : 1024;
    m->order = realloc(m->order, m->order_cap * sizeof(row_t*));
    if (!m->order) die("malloc() failed");
  }
  m->order[m->order_len++] = r;
  return r;
}

/* ---- directory ---------------------------------------------------------- */

/* src/smatrix.c:673-693 smatrix_cmap_probe: start key%size, then (key+1)%size,
 * (key+2)%size ... with 32-bit wrap of the running counter. */
static dslot_t* dir_probe(dslot_t* dir, uint64_t size, uint32_t key) {
  unsigned pos = key;
  dslot_t* s = dir + (key % size);
  for (;;) {
    if (!s->occupied || s->key == key) return s;
    pos++;
    s = dir + (pos % size);
  }
}

static dslot_t* dir_insert(ora_matrix_t* m, uint32_t key);

/* src/smatrix.c:715-741 smatrix_cmap_resize */
static void dir_grow(ora_matrix_t* m) {
  uint64_t old_size = m->dir_size;
  dslot_t* old = m->dir;
  m->dir_size = old_size * 2;
  m->dir_used = 0;
  m->dir = xcalloc(m->dir_size, sizeof(dslot_t));
  m->mem += m->dir_size * REF_SIZEOF_DIRSLOT;
  m->mem -= old_size * REF_SIZEOF_DIRSLOT;
  for (uint64_t p = 0; p < old_size; p++) {
    if (!old[p].occupied) continue;
    dir_insert(m, old[p].key)->row = old[p].row;
  }
  free(old);
}

/* src/smatrix.c:695-713 smatrix_cmap_insert: grow at used*4 >= size*3, tested first */
static dslot_t* dir_insert(ora_matrix_t* m, uint32_t key) {
  if (m->dir_used * 4 >= m->dir_size * 3) dir_grow(m);
  dslot_t* s = dir_probe(m->dir, m->dir_size, key);
  if (!s->occupied || s->key != key) {
    m->dir_used++;
    s->key = key;
    s->occupied = 1;
    s->row = NULL;
  }
  return s;
}

/* src/smatrix.c:621-670 smatrix_cmap_lookup (single-threaded view) */
static row_t* dir_lookup(ora_matrix_t* m, uint32_t x, int create) {
  dslot_t* s = dir_probe(m->dir, m->dir_size, x);
  if (s->occupied && s->key == x) return s->row;
  if (!create) return NULL;
  row_t* r = row_new(m, x, ROW_FIRST_SIZE);
  s = dir_insert(m, x);
  s->row = r;
  return r;
}

/* src/smatrix.c:258-304 smatrix_lookup: resolve (x,y) to a cell.  A probed slot
 * whose key field equals y counts as a hit -- for y==0 that includes the first
 * EMPTY slot (quirk Q1: no insert, `used` untouched). */
static cell_t* locate(ora_matrix_t* m, uint32_t x, uint32_t y, int write, row_t** row_out) {
  row_t* r = dir_lookup(m, x, write);
  if (row_out) *row_out = r;
  if (!r) return NULL;
  cell_t* c = row_probe(r, y);
  if (c->key == y) return c;
  return write ? row_insert(m, r, y) : NULL;
}

/* ---- file format (src/smatrix.c:30-72) ---------------------------------- */

static void put64(unsigned char* p, uint64_t v) { memcpy(p, &v, 8); }
static uint64_t get64(const unsigned char* p) { uint64_t v; memcpy(&v, p, 8); return v; }

static void pwrite_all(int fd, const void* buf, size_t n, uint64_t off) {
  if (pwrite(fd, buf, n, (off_t)off) != (ssize_t)n) die("write() failed");
}

/* Writes the whole matrix as a reference-compatible file: header, chained CMAP
 * blocks of DIRBLK_ENTRIES entries (sparse, like ftruncate at :141 leaves them),
 * entries in row-creation order (:744-757), one RMAP block per row = the raw
 * slot table (:454-482).  Physical row order is not part of the format
 * (the reference flushes LIFO, :904-905). */
static void file_store(ora_matrix_t* m) {
  int fd = open(m->fname, O_RDWR | O_CREAT | O_TRUNC, 00600);
  if (fd == -1) die("cannot open file");

  uint64_t nrows = m->order_len;
  uint64_t nblocks = nrows / DIRBLK_ENTRIES + 1;    /* always >= 1 (:573) */
  uint64_t blk_bytes = DIRBLK_HEAD_BYTES + (uint64_t)DIRBLK_ENTRIES * DIRBLK_ENTRY_BYTES;
  uint64_t rows_at = FILE_HEADER_BYTES + nblocks * blk_bytes;

  uint64_t end = rows_at;
  for (uint64_t i = 0; i < nrows; i++)
    end += ROWBLK_HEAD_BYTES + (uint64_t)m->order[i]->size * 8;
  if (ftruncate(fd, (off_t)end) == -1) die("truncate() failed");

  unsigned char hdr[FILE_HEADER_BYTES];
  memset(hdr, 0, sizeof hdr);
  memset(hdr, 0x17, 8);
  put64(hdr + 8, FILE_HEADER_BYTES);               /* first CMAP block at 512 */
  pwrite_all(fd, hdr, sizeof hdr, 0);

  uint64_t fpos = rows_at;
  for (uint64_t b = 0; b < nblocks; b++) {
    uint64_t at = FILE_HEADER_BYTES + b * blk_bytes;
    unsigned char bh[DIRBLK_HEAD_BYTES];
    put64(bh, DIRBLK_ENTRIES);
    put64(bh + 8, b + 1 < nblocks ? at + blk_bytes : 0);
    pwrite_all(fd, bh, sizeof bh, at);
    uint64_t lo = b * (uint64_t)DIRBLK_ENTRIES;
    uint64_t hi = lo + DIRBLK_ENTRIES < nrows ? lo + DIRBLK_ENTRIES : nrows;
    if (hi > lo) {
      size_t n = (size_t)(hi - lo);
      unsigned char* ents = xcalloc(n, DIRBLK_ENTRY_BYTES);
      for (size_t i = 0; i < n; i++) {
        row_t* r = m->order[lo + i];
        memcpy(ents + i * DIRBLK_ENTRY_BYTES, &r->x, 4);
        put64(ents + i * DIRBLK_ENTRY_BYTES + 4, fpos);
        fpos += ROWBLK_HEAD_BYTES + (uint64_t)r->size * 8;
      }
      pwrite_all(fd, ents, n * DIRBLK_ENTRY_BYTES, at + DIRBLK_HEAD_BYTES);
      free(ents);
    }
  }

  fpos = rows_at;
  for (uint64_t i = 0; i < nrows; i++) {
    row_t* r = m->order[i];
    unsigned char rh[ROWBLK_HEAD_BYTES];
    memset(rh, 0x23, 8);
    put64(rh + 8, r->size);
    pwrite_all(fd, rh, sizeof rh, fpos);
    pwrite_all(fd, r->cells, (size_t)r->size * 8, fpos + ROWBLK_HEAD_BYTES);
    fpos += ROWBLK_HEAD_BYTES + (uint64_t)r->size * 8;
  }
  close(fd);
}

/* src/smatrix.c:576-596 smatrix_fload + :790-830 smatrix_cmap_load +
 * :499-545 smatrix_rmap_load.  Rows are loaded eagerly (lazy loading is not
 * observable through the API).  Loading keeps a slot's key only if its value
 * is non-zero (:533-540, quirk Q4) and recounts `used` as the number of
 * non-zero values. */
static void file_load(ora_matrix_t* m, int fd) {
  unsigned char hdr[FILE_HEADER_BYTES];
  if (pread(fd, hdr, sizeof hdr, 0) != (ssize_t)sizeof hdr) die("invalid file header\n");
  if (hdr[0] != 0x17 || hdr[1] != 0x17) die("invalid file header\n");
  uint64_t at = get64(hdr + 8);
  while (at) {
    unsigned char bh[DIRBLK_HEAD_BYTES];
    if (pread(fd, bh, sizeof bh, (off_t)at) != (ssize_t)sizeof bh)
      die("pread() failed (cmap_load). corrupt file?");
    uint64_t n = get64(bh), next = get64(bh + 8);
    size_t bytes = (size_t)n * DIRBLK_ENTRY_BYTES;
    unsigned char* ents = xcalloc(bytes, 1);
    if (pread(fd, ents, bytes, (off_t)(at + DIRBLK_HEAD_BYTES)) != (ssize_t)bytes)
      die("pread() failed (cmap_load). corrupt file?");
    for (uint64_t i = 0; i < n; i++) {
      uint64_t row_at = get64(ents + i * DIRBLK_ENTRY_BYTES + 4);
      if (!row_at) break;                       /* :814-815 first zero offset ends the block */
      uint32_t x;
      memcpy(&x, ents + i * DIRBLK_ENTRY_BYTES, 4);

      unsigned char rh[ROWBLK_HEAD_BYTES];
      if (pread(fd, rh, sizeof rh, (off_t)row_at) != (ssize_t)sizeof rh)
        die("pread() failed (rmap_load). corrupt file?");
      static const unsigned char magic[8] = {0x23, 0x23, 0x23, 0x23, 0x23, 0x23, 0x23, 0x23};
      if (memcmp(rh, magic, 8)) die("file is corrupt (rmap_load)");
      uint32_t size = (uint32_t)get64(rh + 8);
      size_t rbytes = (size_t)size * 8;
      cell_t* disk = xcalloc(size, sizeof(cell_t));
      if (pread(fd, disk, rbytes, (off_t)(row_at + ROWBLK_HEAD_BYTES)) != (ssize_t)rbytes)
        die("read() failed (rmap_load)");

      row_t* r = row_new(m, x, size);
      for (uint32_t p = 0; p < size; p++) {
        r->cells[p].value = disk[p].value;
        if (disk[p].value) {
          r->cells[p].key = disk[p].key;
          r->used++;
        }
      }
      free(disk);
      /* a key that appears twice in the chain keeps the later row, like
       * smatrix_cmap_insert(...)->rmap = rmap at :823 */
      dir_insert(m, x)->row = r;
    }
    free(ents);
    at = next;
  }
}

/* ---- public entry points ------------------------------------------------ */

/* src/smatrix.c:74-111 */
ora_matrix_t* ora_open(const char* fname) {
  ora_matrix_t* m = calloc(1, sizeof *m);
  if (!m) return NULL;
  m->dir_size = DIR_FIRST_SIZE;                   /* :598-612 */
  m->dir = xcalloc(m->dir_size, sizeof(dslot_t));
  m->mem += m->dir_size * REF_SIZEOF_DIRSLOT;
  if (!fname) return m;

  int fd = open(fname, O_RDWR | O_CREAT, 00600);
  if (fd == -1) {
    perror("cannot open file");
    free(m->dir);
    free(m);
    return NULL;
  }
  m->fname = strdup(fname);
  if (lseek(fd, 0, SEEK_END) != 0) file_load(m, fd);
  close(fd);
  return m;
}

/* src/smatrix.c:113-133: close is the flush barrier in file mode */
void ora_close(ora_matrix_t* m) {
  if (m->fname) file_store(m);
  for (uint64_t i = 0; i < m->order_len; i++) {
    free(m->order[i]->cells);
    free(m->order[i]);
  }
  free(m->order);
  free(m->dir);
  free(m->fname);
  free(m);
}

/* src/smatrix.c:174-185 */
uint32_t ora_get(ora_matrix_t* m, uint32_t x, uint32_t y) {
  cell_t* c = locate(m, x, y, 0, NULL);
  return c ? c->value : 0;
}

/* src/smatrix.c:225-234 */
uint32_t ora_set(ora_matrix_t* m, uint32_t x, uint32_t y, uint32_t value) {
  return locate(m, x, y, 1, NULL)->value = value;
}

/* src/smatrix.c:236-245 */
uint32_t ora_incr(ora_matrix_t* m, uint32_t x, uint32_t y, uint32_t value) {
  return locate(m, x, y, 1, NULL)->value += value;
}

/* src/smatrix.c:247-256 */
uint32_t ora_decr(ora_matrix_t* m, uint32_t x, uint32_t y, uint32_t value) {
  return locate(m, x, y, 1, NULL)->value -= value;
}

/* src/smatrix.c:212-223 */
uint32_t ora_rowlen(ora_matrix_t* m, uint32_t x) {
  row_t* r = dir_lookup(m, x, 0);
  return r ? r->used : 0;
}

/* src/smatrix.c:189-210: pairs in slot order; stops once ++num*8 >= ret_len
 * (ret_len counts BYTES; at least one pair is written for a non-empty row) */
uint32_t ora_getrow(ora_matrix_t* m, uint32_t x, uint32_t* ret, size_t ret_len) {
  row_t* r = dir_lookup(m, x, 0);
  uint32_t num = 0;
  if (!r) return 0;
  for (uint32_t p = 0; p < r->size; p++) {
    if (cell_is_empty(&r->cells[p])) continue;
    ret[num * 2] = r->cells[p].key;
    ret[num * 2 + 1] = r->cells[p].value;
    if ((++num * 2 * sizeof(uint32_t)) >= ret_len) break;
  }
  return num;
}

/* ---- checker conveniences ----------------------------------------------- */

void ora_apply(ora_matrix_t* m, int op, size_t n, const uint32_t* x,
               const uint32_t* y, const uint32_t* v, uint32_t* out) {
  for (size_t i = 0; i < n; i++) {
    uint32_t r;
    switch (op) {
      case ORA_OP_GET:  r = ora_get(m, x[i], y[i]); break;
      case ORA_OP_SET:  r = ora_set(m, x[i], y[i], v[i]); break;
      case ORA_OP_INCR: r = ora_incr(m, x[i], y[i], v[i]); break;
      default:          r = ora_decr(m, x[i], y[i], v[i]); break;
    }
    if (out) out[i] = r;
  }
}

uint64_t ora_sum_get(ora_matrix_t* m, size_t n, const uint32_t* x, const uint32_t* y) {
  uint64_t s = 0;
  for (size_t i = 0; i < n; i++) s += ora_get(m, x[i], y[i]);
  return s;
}

/* examples/cf_recommender.c:50-64 neighbors_for_item + :67-86 cf_cosine.  (The example does not
 * compile as shipped -- missing brace at :50-51, undeclared `pset` -- so this follows its evident
 * intent: total = get(item,0); for each getrow pair (b, cc): cc / (sqrt(total)*sqrt(get(b,0))),
 * get(b,0)==0 -> 1, den==0 -> 0, cc > den -> 0.) */
#include <math.h>
uint32_t ora_cf_neighbors(ora_matrix_t* m, uint32_t item, uint32_t* ids, double* scores, uint32_t cap) {
  row_t* r = dir_lookup(m, item, 0);
  uint32_t n = 0;
  if (!r) return 0;
  uint32_t a_total = ora_get(m, item, 0);
  for (uint32_t p = 0; p < r->size && n < cap; p++) {
    if (cell_is_empty(&r->cells[p])) continue;
    uint32_t b_total = ora_get(m, r->cells[p].key, 0);
    if (b_total == 0) b_total = 1;
    double num = r->cells[p].value;
    double den = sqrt((double)a_total) * sqrt((double)b_total);
    double score = 0.0;
    if (den != 0.0 && !(num > den)) score = num / den;
    ids[n] = r->cells[p].key;
    scores[n] = score;
    n++;
  }
  return n;
}

/* examples/cf_recommender.c:36-47 import_preference_set: for every position n of the session
 *   incr(ids[n], 0, 1)  and, for every OTHER position i, incr(ids[n], ids[i], 1)
 * (the example loops `i < pset->len` with an undeclared pset; the evident intent is the session itself).  The test is on
 * POSITIONS (i != n), so an id that occurs twice in a session also counts itself. */
void ora_cf_import_preference_set(ora_matrix_t* m, const uint32_t* ids, uint32_t num_ids) {
  for (uint32_t n = 0; n < num_ids; n++) {
    ora_incr(m, ids[n], 0, 1);
    for (uint32_t i = 0; i < num_ids; i++)
      if (i != n) ora_incr(m, ids[n], ids[i], 1);
  }
}

uint64_t ora_num_rows(ora_matrix_t* m) { return m->dir_used; }
uint64_t ora_dir_size(ora_matrix_t* m) { return m->dir_size; }
uint64_t ora_mem(ora_matrix_t* m) { return m->mem; }

uint64_t ora_nnz(ora_matrix_t* m) {
  uint64_t n = 0;
  for (uint64_t i = 0; i < m->order_len; i++) {
    row_t* r = m->order[i];
    for (uint32_t p = 0; p < r->size; p++) n += !cell_is_empty(&r->cells[p]);
  }
  return n;
}

int ora_row_info(ora_matrix_t* m, uint32_t x, uint32_t* size, uint32_t* used) {
  row_t* r = dir_lookup(m, x, 0);
  if (!r) return 0;
  if (size) *size = r->size;
  if (used) *used = r->used;
  return 1;
}

uint32_t ora_row_slots(ora_matrix_t* m, uint32_t x, uint32_t* kv, uint32_t cap_slots) {
  row_t* r = dir_lookup(m, x, 0);
  if (!r) return 0;
  uint32_t n = r->size < cap_slots ? r->size : cap_slots;
  memcpy(kv, r->cells, (size_t)n * 8);
  return r->size;
}

uint64_t ora_list_rows(ora_matrix_t* m, uint32_t* xs, uint64_t cap) {
  uint64_t n = 0;
  for (uint64_t p = 0; p < m->dir_size && n < cap; p++)
    if (m->dir[p].occupied) xs[n++] = m->dir[p].key;
  return n;
}
