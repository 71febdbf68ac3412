import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from libsmatrix_amd import SparseMatrix, Stream
B = 1 << 24
gen = Stream("zipf", 12345, 1000000, 1.1, 1)
m = SparseMatrix()
ones = np.ones(B, np.uint32); r1 = np.zeros(B, np.uint32)
for s in range(3):
    x, y = gen.fill(s * B, B)
    print("=== call", s, flush=True); sys.stderr.flush()
    t0 = time.perf_counter(); m.incr_batch(x, y, ones, out=r1); t1 = time.perf_counter()
    print("incr %.2f ms" % ((t1 - t0) * 1e3), flush=True)
m.close()
