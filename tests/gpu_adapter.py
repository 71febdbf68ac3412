"""Gives the HIP library (through its C ABI, via libsmatrix_amd.SparseMatrix) the same
method names as the checker classes in oracle/oracle.py so tests/replay.py can drive both."""
import numpy as np

from libsmatrix_amd import SparseMatrix


class GpuMatrix:
    def __init__(self, fname=None, scalar=False):
        """scalar=True: apply() issues one C-ABI scalar call per op (reference call pattern)."""
        self.m = SparseMatrix(fname)
        self.scalar = scalar

    def get(self, x, y): return self.m.get(x, y)
    def set(self, x, y, v): return self.m.set(x, y, v)
    def incr(self, x, y, v): return self.m.incr(x, y, v)
    def decr(self, x, y, v): return self.m.decr(x, y, v)
    def rowlen(self, x): return self.m.getRowLength(x)

    def getrow(self, x, ret_len_bytes=None):
        if ret_len_bytes is None:
            ret_len_bytes = (self.rowlen(x) + 1) * 8
        return self.m.getrow_raw(x, ret_len_bytes)

    def row_info(self, x): return self.m.row_info(x)
    def row_slots(self, x): return self.m.row_slots(x)

    def apply(self, op, x, y, v=None):
        x = np.asarray(x, dtype=np.uint32); y = np.asarray(y, dtype=np.uint32)
        if v is None:
            v = np.zeros_like(x)
        if not self.scalar:
            return self.m.apply_batch(op, x, y, v)
        f = (lambda a, b, c: self.m.get(a, b), self.m.set, self.m.incr, self.m.decr)[op]
        return np.array([f(int(a), int(b), int(c)) for a, b, c in zip(x, y, v)], dtype=np.uint32)

    def sum_get(self, x, y):
        return int(self.m.get_batch(x, y).astype(np.uint64).sum())

    def stats(self): return self.m.stats()
    def close(self): self.m.close()
