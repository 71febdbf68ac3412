#!/usr/bin/env python3
"""GPU probe: latency of write batches of various sizes on a populated config-2 table (device pointers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import SparseMatrix, Stream, OP_INCR, OP_GET
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
gen = Stream("zipf", 12345, 1000000, 1.1, 1)
B = 1 << 24
m = SparseMatrix()
x = torch.empty(B, dtype=torch.int32, device=dev); y = torch.empty_like(x); ones = torch.ones_like(x); out = torch.empty_like(x)
for s in range(6):
    gen.fill_device(s * B, B, x.data_ptr(), y.data_ptr(), st)
    m.apply_batch_dev(OP_INCR, B, x.data_ptr(), y.data_ptr(), ones.data_ptr(), out.data_ptr(), st)
pos = 6 * B
for n in (1, 64, 1024, 10000, 100000, 1000000):
    ts = []; r0 = m.stats()["rounds"]
    for rep in range(20):
        gen.fill_device(pos, n, x.data_ptr(), y.data_ptr(), st); pos += n
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.apply_batch_dev(OP_INCR, n, x.data_ptr(), y.data_ptr(), ones.data_ptr(), out.data_ptr(), st)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("incr batch n=%8d: median %8.1f us  (%.2f rounds per batch, %.1f Mops/s)" % (n, ts[10] * 1e6, (m.stats()["rounds"] - r0) / 20, n / ts[10] / 1e6))
