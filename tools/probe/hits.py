#!/usr/bin/env python3
"""GPU probe: all-hit incr/get kernel time on a table built with the stock library, measured
with the library named by SMATRIX_LIB_AB (ablation builds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_INCR
B = 1 << 24
dev = torch.device("cuda", 0)
gen = Stream("zipf", 12345, 1000000, 1.1, 1)
xs = torch.empty((3, B), dtype=torch.int32, device=dev); ys = torch.empty_like(xs)
ones = torch.ones(B, dtype=torch.int32, device=dev); out = torch.empty(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for s in range(3):
    gen.fill_device(s * B, B, xs[s].data_ptr(), ys[s].data_ptr(), st)
torch.cuda.synchronize()
m = SparseMatrix()
for s in range(3):
    m.apply_batch_dev(OP_INCR, B, xs[s].data_ptr(), ys[s].data_ptr(), ones.data_ptr(), out.data_ptr(), st)
for rep in range(3):
    m.profile(True)
    m.apply_batch_dev(OP_INCR, B, xs[rep].data_ptr(), ys[rep].data_ptr(), ones.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize()
    print("all-hit incr batch %d: kernel %.3f ms" % (rep, m.stats()["kernel_ms_incr"]))
