#!/usr/bin/env python3
"""GPU check: getrow over the config-2 table (1 M rows, 100 M cells, one row of 2 M slots) -- all rows in one device
call, and the largest row alone -- with HIP events.  The giant row is cut into 32768-cell segments (k_getrow_big)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from libsmatrix_amd import SparseMatrix, Stream, OP_INCR

dev = torch.device("cuda", 0)
B, steps = 1 << 24, int(sys.argv[1]) if len(sys.argv) > 1 else 24
st = torch.cuda.current_stream().cuda_stream
gen = Stream("zipf", 12345, 1000000, 1.1, 1)
m = SparseMatrix()
x = torch.empty(B, dtype=torch.int32, device=dev); y = torch.empty_like(x)
ones = torch.ones(B, dtype=torch.int32, device=dev); o = torch.empty_like(ones)
rows = torch.empty(0, dtype=torch.int32, device=dev)
for k in range(steps):
    gen.fill_device(k * B, B, x.data_ptr(), y.data_ptr(), st)
    m.apply_batch_dev(OP_INCR, B, x.data_ptr(), y.data_ptr(), ones.data_ptr(), o.data_ptr(), st)
    rows = torch.unique(torch.cat([rows, x]))
torch.cuda.synchronize()
assert rows.numel() == m.stats()["rows"]
n = rows.numel()
lens = torch.empty(n, dtype=torch.int32, device=dev)
m.rowlen_batch_dev(n, rows.data_ptr(), lens.data_ptr(), st)
torch.cuda.synchronize()
l64 = lens.to(torch.int64) & 0xFFFFFFFF
off = torch.zeros(n + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(l64 + 1, 0)
total = int(off[-1].item())
ret = torch.empty((total, 2), dtype=torch.int32, device=dev); cnt = torch.empty(n, dtype=torch.int32, device=dev)


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


t_all = timed(lambda: m.getrow_batch_dev(n, rows.data_ptr(), off.data_ptr(), ret.data_ptr(), cnt.data_ptr(), st))
assert int((cnt.to(torch.int64) & 0xFFFFFFFF).sum().item()) == int(l64.sum().item())
nnz = int(l64.sum().item())
big = int(torch.argmax(l64).item())
one = rows[big:big + 1].contiguous(); off1 = torch.tensor([0, int(l64[big].item()) + 1], dtype=torch.int64, device=dev)
ret1 = torch.empty((int(off1[1].item()), 2), dtype=torch.int32, device=dev); cnt1 = torch.empty(1, dtype=torch.int32, device=dev)
t_one = timed(lambda: m.getrow_batch_dev(1, one.data_ptr(), off1.data_ptr(), ret1.data_ptr(), cnt1.data_ptr(), st))
a = int(off[big].item())
assert (ret1[:int(l64[big].item())] == ret[a:a + int(l64[big].item())]).all()
t0 = time.perf_counter(); xr = int(one.item()) & 0xFFFFFFFF; h = m.getrow_raw(xr, m.getRowLength(xr) * 8); t_host = time.perf_counter() - t0
assert h.shape[0] == int(l64[big].item())
print("rows %d nnz %d | all rows: %.3f ms (%.1f G nnz/s) | largest row (%d pairs) alone: %.3f ms device, %.2f ms through smatrix_getrow"
      % (n, nnz, t_all, nnz / t_all / 1e6, int(l64[big].item()), t_one, t_host * 1e3))
