#!/usr/bin/env python3
"""GPU probe: how much of the all-hit incr kernel time is the serialisation of the hottest cells?
Replaces ops whose (x,y) are both among the top-K Zipf ranks by copies of other ops of the batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libsmatrix_amd import SparseMatrix, Stream, OP_INCR, _lib
B = 1 << 24
dev = torch.device("cuda", 0)
gen = Stream("zipf", 12345, 1000000, 1.1, 1)
xs = torch.empty((4, B), dtype=torch.int32, device=dev); ys = torch.empty_like(xs)
ones = torch.ones(B, dtype=torch.int32, device=dev); out = torch.empty(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for s in range(4):
    gen.fill_device(s * B, B, xs[s].data_ptr(), ys[s].data_ptr(), st)
torch.cuda.synchronize()
m = SparseMatrix()
for s in range(4):
    m.apply_batch_dev(OP_INCR, B, xs[s].data_ptr(), ys[s].data_ptr(), ones.data_ptr(), out.data_ptr(), st)
lib = _lib.load()
def timed(x, y, label):
    m.profile(True)
    m.apply_batch_dev(OP_INCR, B, x.data_ptr(), y.data_ptr(), ones.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize()
    print("%-50s kernel %.3f ms" % (label, m.stats()["kernel_ms_incr"]))
x, y = xs[3], ys[3]
timed(x, y, "all-hit, full Zipf batch")
for K in (3, 10, 100, 1000):
    top = torch.tensor([lib.smx_fmix32(r) for r in range(1, K + 1)], dtype=torch.int64, device=dev)
    hot = torch.isin(x.long() & 0xFFFFFFFF, top) & torch.isin(y.long() & 0xFFFFFFFF, top)
    cold_idx = torch.nonzero(~hot).squeeze(1)
    repl = cold_idx[torch.randint(0, cold_idx.numel(), (int(hot.sum()),), device=dev)]
    x2, y2 = x.clone(), y.clone()
    x2[hot] = x[repl]; y2[hot] = y[repl]
    timed(x2, y2, "top-%d x top-%d cells removed (%.2f%% of ops)" % (K, K, 100.0 * hot.float().mean().item()))
