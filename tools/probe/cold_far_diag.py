"""Diagnostic: the first batch of tests/test_gpu_round5.py::test_cold_start_of_dense_ids_takes_small_keys_first against the oracle --
which rows differ in size / used / cells, and whether a probe sequence has a hole (python tools/probe/cold_far_diag.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ.setdefault("SMATRIX_COLD_MIN", "4096")
from libsmatrix_amd import Stream
from tests.gpu_adapter import GpuMatrix
from oracle import oracle as O
gen = Stream("zipf", 4242, 1000000, 1.1, 0)
g, o = GpuMatrix(), O.Oracle()
n = 1 << 21
x, y = gen.fill(0, n)
x = (x % 48).astype(np.uint32)
v = np.ones(n, np.uint32)
a, b = g.apply(2, x, y, v), o.apply(2, x, y, v)
kk = x.astype(np.uint64) << np.uint64(32) | y
sa, sb = a[np.lexsort((a, kk))], b[np.lexsort((b, kk))]
ks = kk[np.lexsort((a, kk))]
bad = np.flatnonzero(sa != sb)
print("returns differ at", bad.size, "of", n, "ops;", np.unique(ks[bad]).size, "keys; rows", np.unique(ks[bad] >> np.uint64(32))[:20])
for r in o.list_rows().tolist():
    gi, oi = g.row_info(r), o.row_info(r)
    A = np.asarray(g.row_slots(r)); B = np.asarray(o.row_slots(r))
    ne = (A[:, 0] != 0) | (A[:, 1] != 0)
    ka = A[ne]; kb = B[(B[:, 0] != 0) | (B[:, 1] != 0)]
    dup = ka.shape[0] - np.unique(ka[:, 0]).size
    size = A.shape[0]; pos = np.flatnonzero(ne); empties = np.flatnonzero(~ne)
    home = A[pos, 0].astype(np.int64) & (size - 1)
    nxt = empties[np.searchsorted(empties, home) % empties.size]
    holes = int((((nxt - home) % size) <= ((pos - home) % size)).sum())
    sa_ = set(map(tuple, ka.tolist())); sb_ = set(map(tuple, kb.tolist()))
    if gi != oi or dup or holes or sa_ != sb_:
        print("row", r, "info", gi, oi, "cells", ka.shape[0], kb.shape[0], "keys twice", dup, "unreachable", holes, "only here", len(sa_ - sb_), "only oracle", len(sb_ - sa_), list(sa_ - sb_)[:4], list(sb_ - sa_)[:4])
print(g.stats()["cold_starts"], "cold starts")
