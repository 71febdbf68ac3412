#!/usr/bin/env python3
"""GPU probe: latency of the scalar (drop-in) ABI."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from libsmatrix_amd import SparseMatrix
m = SparseMatrix()
N = 20000
t0 = time.perf_counter()
for i in range(N): m.incr(i % 500, 1 + i % 97, 1)
t1 = time.perf_counter()
for i in range(N): m.get(i % 500, 1 + i % 97)
t2 = time.perf_counter()
for i in range(N): m.set(1000 + i, 1 + i, 5)        # every op creates a row
t3 = time.perf_counter()
print("scalar incr %.1f us/op, get %.1f us/op, set(new row) %.1f us/op" % ((t1 - t0) / N * 1e6, (t2 - t1) / N * 1e6, (t3 - t2) / N * 1e6))
m.close()
import threading
m = SparseMatrix()
for T in (1, 2, 8, 32):
    N = 4000
    def work(t):
        for i in range(N):
            m.incr((t * 131 + i) % 997, 1 + i % 97, 1)
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    print("T=%2d threads: %.1f us per call per thread, %.0f k calls/s in total" % (T, dt / N * 1e6, T * N / dt / 1e3))
tot = sum(int(v) for v in m.get_batch([a for a in range(997) for b in range(1, 98)], [b for a in range(997) for b in range(1, 98)]))
print("total", tot, "expected", (1 + 2 + 8 + 32) * 4000)
m.close()
