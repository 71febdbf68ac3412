"""diagnostic: round trace of a batch that creates 600 rows whose ids share one directory home slot; and of dense-id batches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SMATRIX_TRACE_ROUNDS"] = "1"
import numpy as np
import torch
from libsmatrix_amd import SparseMatrix, Stream
ids = np.arange(1, 1 << 26, dtype=np.uint32)
h = ids.copy()
h ^= h >> 16; h *= np.uint32(0x85ebca6b); h ^= h >> 13; h *= np.uint32(0xc2b2ae35); h ^= h >> 16
crowd = ids[(h & 0xFFFF) == 0x1234][:600]
m = SparseMatrix()
x = np.repeat(crowd, 3); y = np.tile(np.array([1, 2, 17], np.uint32), crowd.size)
m.incr_batch(x, y, np.ones_like(x))
print("rows", m.stats()["rows"], "rounds", m.stats()["rounds"], flush=True)
m.close()
if len(sys.argv) > 1:
    print("---- dense", flush=True)
    g = Stream("zipf", 12345, 1000000, 1.1, 0)
    m = SparseMatrix()
    for b in range(4):
        x, y = g.fill(b << 24, 1 << 24)
        m.incr_batch(x, y, np.ones_like(x))
    m.close()
