#!/usr/bin/env python3
"""GPU probe: throughput of the host-pointer batch API (PCIe copies included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libsmatrix_amd import SparseMatrix, Stream
B = 1 << 24
gen = Stream("zipf", 12345, 1000000, 1.1, 1)
m = SparseMatrix()
ones = np.ones(B, np.uint32)
r1 = np.zeros(B, np.uint32); r2 = np.zeros(B, np.uint32)       # the caller's own result arrays, reused (what a binding passes)
for s in range(5):
    x, y = gen.fill(s * B, B)
    t0 = time.perf_counter(); m.incr_batch(x, y, ones, out=r1); t1 = time.perf_counter(); m.get_batch(x, y, out=r2); t2 = time.perf_counter()
    print("batch %d: incr %.1f ms (%.2f Gops/s)  get %.1f ms (%.2f Gops/s)" % (s, (t1 - t0) * 1e3, B / (t1 - t0) / 1e9, (t2 - t1) * 1e3, B / (t2 - t1) / 1e9))
m.close()
