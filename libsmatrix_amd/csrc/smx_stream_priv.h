/* smx_stream_priv.h -- layout of smx_stream_t shared by smx_stream.c and the HIP side */
#ifndef SMX_STREAM_PRIV_H
#define SMX_STREAM_PRIV_H
#include "../../include/smx_stream.h"

#ifdef __cplusplus
extern "C" {
#endif

struct smx_stream {
  int      dist;
  int      scramble;
  uint64_t seed;
  uint32_t n_ids;
  double   zipf_s;
  double*  cdf;      /* host, n_ids entries (zipf only) */
  double*  d_cdf;    /* device copy, made on first smx_stream_fill_device */
};

/* frees d_cdf; defined on the HIP side (smx_runtime.cpp) */
void smx_stream_release_device(smx_stream_t* s);

#ifdef __cplusplus
}
#endif
#endif
