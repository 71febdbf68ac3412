/*
 * smatrix_shim.c -- the "thin C shim" of the drop-in boundary.
 *
 * The reference's bindings link the OBJECT file ../smatrix.o into their own shared
 * library (src/java/Makefile:22-23, src/ruby/Makefile:18-19) with no extra -l flags.
 * This shim is that object: it exports the eight public symbols of
 * src/smatrix.h:87-94 and forwards each to the HIP library smatrix.so, which it
 * dlopen()s on first use (path: $SMATRIX_HIP_LIB, else smatrix.so next to the
 * process's library search path).  No CPU fallback: if the HIP library cannot be
 * loaded, smatrix_open prints the reason and returns NULL (src/smatrix.c:92-96
 * convention); any other call aborts like smatrix_error (src/smatrix.c:891-894).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/smatrix.h"

static struct {
  void* lib;
  smatrix_t* (*open)(const char*);
  void (*close)(smatrix_t*);
  uint32_t (*get)(smatrix_t*, uint32_t, uint32_t);
  uint32_t (*set)(smatrix_t*, uint32_t, uint32_t, uint32_t);
  uint32_t (*incr)(smatrix_t*, uint32_t, uint32_t, uint32_t);
  uint32_t (*decr)(smatrix_t*, uint32_t, uint32_t, uint32_t);
  uint32_t (*rowlen)(smatrix_t*, uint32_t);
  uint32_t (*getrow)(smatrix_t*, uint32_t, uint32_t*, size_t);
} hip;

static int bind(void) {
  if (hip.lib) return 1;
  const char* path = getenv("SMATRIX_HIP_LIB");
  void* lib = dlopen(path && *path ? path : "smatrix.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    fprintf(stderr, "libsmatrix: cannot load the HIP library: %s\n", dlerror());
    return 0;
  }
#define B(name) *(void**)(&hip.name) = dlsym(lib, "smatrix_" #name); if (!hip.name) { fprintf(stderr, "libsmatrix: missing symbol smatrix_" #name "\n"); return 0; }
  B(open) B(close) B(get) B(set) B(incr) B(decr) B(rowlen) B(getrow)
#undef B
  hip.lib = lib;
  return 1;
}

static void need(void) {
  if (!hip.lib) {
    printf("libsmatrix error: HIP library not loaded");
    abort();
  }
}

smatrix_t* smatrix_open(const char* fname) { return bind() ? hip.open(fname) : NULL; }
void smatrix_close(smatrix_t* self) { need(); hip.close(self); }
uint32_t smatrix_get(smatrix_t* self, uint32_t x, uint32_t y) { need(); return hip.get(self, x, y); }
uint32_t smatrix_set(smatrix_t* self, uint32_t x, uint32_t y, uint32_t v) { need(); return hip.set(self, x, y, v); }
uint32_t smatrix_incr(smatrix_t* self, uint32_t x, uint32_t y, uint32_t v) { need(); return hip.incr(self, x, y, v); }
uint32_t smatrix_decr(smatrix_t* self, uint32_t x, uint32_t y, uint32_t v) { need(); return hip.decr(self, x, y, v); }
uint32_t smatrix_rowlen(smatrix_t* self, uint32_t x) { need(); return hip.rowlen(self, x); }
uint32_t smatrix_getrow(smatrix_t* self, uint32_t x, uint32_t* ret, size_t ret_len) { need(); return hip.getrow(self, x, ret, ret_len); }
