import os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from libsmatrix_amd import SparseMatrix
rng = np.random.default_rng(17)
d = tempfile.mkdtemp()
p = os.path.join(d, "c.smx")
m = SparseMatrix(p)
HDR = 512 + 16 + 4194304 * 12
def live(rows):
    return sum(16 + 8 * m.row_info(int(r))[0] for r in rows)
x = rng.integers(0, 5000, 300000, dtype=np.uint32); y = rng.integers(1, 1 << 20, 300000, dtype=np.uint32)
m.incr_batch(x, y, np.ones_like(x)); m.flush()
rows = set(np.unique(x).tolist())
print("flush1 size-HDR", os.path.getsize(p) - HDR, "live", live(rows), m.stats()["file_leaked_bytes"])
x2 = np.concatenate([rng.integers(100, 200, 40000, dtype=np.uint32), rng.integers(9000, 9050, 5000, dtype=np.uint32)])
y2 = rng.integers(1, 1 << 20, x2.size, dtype=np.uint32)
m.incr_batch(x2, y2, np.ones_like(x2)); m.flush()
rows |= set(np.unique(x2).tolist())
st = m.stats()
print("flush2 size-HDR", os.path.getsize(p) - HDR, "live", live(rows), "leaked", st["file_leaked_bytes"], "bulk", st["bulk_rounds"], st["bulk_ops"])
m.compact()
print("compact size-HDR", os.path.getsize(p) - HDR, "live", live(rows), m.stats()["file_leaked_bytes"])
m.close()
