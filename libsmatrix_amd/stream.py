"""Synthetic (x,y) streams -- Python face of include/smx_stream.h.

Counterpart of the reference benchmark's workload loops
(src/smatrix_benchmark.c:29-65) widened to the uniform / Zipf streams that
BASELINE.json's configs name.  Host generation needs no GPU; `fill_device`
generates straight into HBM.
"""
import ctypes as C

import numpy as np

from . import _lib

DIST = {"uniform": 0, "zipf": 1, "cf": 2}     # "cf": zipf_s carries the ops per row (include/smx_stream.h)


class Stream:
    def __init__(self, dist="zipf", seed=12345, n_ids=1000000, zipf_s=1.1, scramble=1):
        self._lib = _lib.load()
        self._h = self._lib.smx_stream_new(DIST[dist], seed, n_ids, zipf_s, int(scramble))
        if not self._h:
            raise ValueError("bad stream parameters")
        self.n_ids = n_ids

    def fill(self, first, n):
        """ops [first, first+n) as two uint32 numpy arrays (host)."""
        x = np.empty(n, dtype=np.uint32)
        y = np.empty(n, dtype=np.uint32)
        self._lib.smx_stream_fill(self._h, first, n, x.ctypes.data_as(_lib.u32p),
                                  y.ctypes.data_as(_lib.u32p))
        return x, y

    def fill_device(self, first, n, x_ptr, y_ptr, stream=None):
        """same ops into device memory (raw pointers, e.g. torch tensor .data_ptr())."""
        rc = self._lib.smx_stream_fill_device(self._h, first, n, x_ptr, y_ptr, stream)
        if rc:
            raise RuntimeError("smx_stream_fill_device failed")

    def close(self):
        if self._h:
            self._lib.smx_stream_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
