/*
 * ref_probe.c -- introspection shim linked INTO the compiled reference
 * (oracle/_ref/libsmatrix_ref.so).  TEST INFRASTRUCTURE ONLY.
 *
 * It is compiled against the reference's own header where it lies
 * (-I/root/reference/src, see oracle/Makefile); nothing from the reference is
 * copied here.  It lets the checker read a row's table geometry and raw slots
 * (the public struct layouts, src/smatrix.h:35-85) so that layouts can be
 * compared slot for slot.
 */
#include <string.h>
#include <smatrix.h>

/* non-static in the reference: src/smatrix.c:673, declared in smatrix_private.h:34 */
smatrix_cmap_slot_t* smatrix_cmap_probe(smatrix_cmap_t* cmap, uint32_t key);

static smatrix_rmap_t* find_row(smatrix_t* m, uint32_t x) {
  smatrix_rowlen(m, x); /* forces the lazy load in file mode (src/smatrix.c:279-289) */
  smatrix_cmap_slot_t* s = smatrix_cmap_probe(&m->cmap, x);
  if (s && (s->flags & SMATRIX_CMAP_SLOT_USED) && s->key == x) return s->rmap;
  return NULL;
}

uint64_t ref_num_rows(smatrix_t* m) { return m->cmap.used; }
uint64_t ref_dir_size(smatrix_t* m) { return m->cmap.size; }
uint64_t ref_mem(smatrix_t* m) { return m->mem; }

int ref_row_info(smatrix_t* m, uint32_t x, uint32_t* size, uint32_t* used) {
  smatrix_rmap_t* r = find_row(m, x);
  if (!r) return 0;
  if (size) *size = r->size;
  if (used) *used = r->used;
  return 1;
}

uint32_t ref_row_slots(smatrix_t* m, uint32_t x, uint32_t* kv, uint32_t cap_slots) {
  smatrix_rmap_t* r = find_row(m, x);
  if (!r) return 0;
  uint32_t n = r->size < cap_slots ? r->size : cap_slots;
  memcpy(kv, r->data, (size_t)n * 8);
  return r->size;
}

uint64_t ref_list_rows(smatrix_t* m, uint32_t* xs, uint64_t cap) {
  uint64_t n = 0, p;
  for (p = 0; p < m->cmap.size && n < cap; p++)
    if (m->cmap.data[p].flags & SMATRIX_CMAP_SLOT_USED) xs[n++] = m->cmap.data[p].key;
  return n;
}

/* batch drivers so that Python does not pay one ctypes call per op */
void ref_apply(smatrix_t* m, int op, size_t n, const uint32_t* x, const uint32_t* y,
               const uint32_t* v, uint32_t* out) {
  size_t i;
  for (i = 0; i < n; i++) {
    uint32_t r;
    switch (op) {
      case 0:  r = smatrix_get(m, x[i], y[i]); break;
      case 1:  r = smatrix_set(m, x[i], y[i], v[i]); break;
      case 2:  r = smatrix_incr(m, x[i], y[i], v[i]); break;
      default: r = smatrix_decr(m, x[i], y[i], v[i]); break;
    }
    if (out) out[i] = r;
  }
}

uint64_t ref_sum_get(smatrix_t* m, size_t n, const uint32_t* x, const uint32_t* y) {
  uint64_t s = 0;
  size_t i;
  for (i = 0; i < n; i++) s += smatrix_get(m, x[i], y[i]);
  return s;
}
