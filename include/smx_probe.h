/*
 * smx_probe.h -- micro-kernels that measure the chip's random-access rates, the ceiling the
 * op kernels are priced against ("R_rand is not a datasheet number").
 * Each touch addresses a pseudo-random, 8-byte aligned word of d_buf[0, bytes).
 */
#ifndef SMX_PROBE_H
#define SMX_PROBE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  SMX_PROBE_READ8 = 0,        /* independent 8-byte loads                                  */
  SMX_PROBE_ATOMIC_RET = 1,   /* 32-bit atomicAdd whose old value is used                  */
  SMX_PROBE_ATOMIC_NORET = 2, /* 32-bit atomicAdd, result unused                           */
  SMX_PROBE_CHAIN2 = 3        /* 16-byte load, then an 8-byte load whose address depends on it
                                 (the shape of get: directory slot -> cell)                */
};

/* enqueues ONE kernel doing `touches` touches on hip_stream (time it with events on that
 * stream); d_sink: >= 8 bytes of device memory that receives a checksum.  0 on success. */
int smx_probe_random_dev(void* d_buf, size_t bytes, size_t touches, int mode, uint64_t seed,
                         void* d_sink, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
