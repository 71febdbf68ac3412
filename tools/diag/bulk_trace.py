import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SMATRIX_TRACE_ROUNDS"] = "1"
os.environ["SMATRIX_BULK_MIN"] = "1"
import numpy as np
import torch
from libsmatrix_amd import SparseMatrix
rng = np.random.default_rng(41)
m = SparseMatrix()
x = rng.integers(0, 3000, 40000, dtype=np.uint32); y = rng.integers(1, 1 << 20, 40000, dtype=np.uint32)
m.incr_batch(x, y, np.ones_like(x))
print(m.stats())
m.close()
