#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import SparseMatrix, OP_INCR, OP_GET
from tools.bench_getrow import fmix32, as_i32
rows = 2500000; npr = 115
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev); g.manual_seed(12345)
m = SparseMatrix()
rpb = (1 << 24) // npr
hist = []
last = None
for r0 in range(0, rows, rpb):
    r1 = min(rows, r0 + rpb)
    rid = torch.arange(r0 + 1, r1 + 1, device=dev, dtype=torch.int64)
    x = as_i32(fmix32(rid).repeat_interleave(npr))
    y = as_i32(fmix32(1 + torch.randint(0, 13000000, (x.numel(),), device=dev, generator=g, dtype=torch.int64)))
    ones = torch.ones_like(x); out = torch.empty_like(x)
    m.apply_batch_dev(OP_INCR, x.numel(), x.data_ptr(), y.data_ptr(), ones.data_ptr(), out.data_ptr(), st)
    last = (x, y)                       # keep the previous batch alive like bench_file does
    hist.append((rid, x, y, out))
    if len(hist) > 2: hist.pop(0)
torch.cuda.synchronize()
# verify everything at the end
g = torch.Generator(device=dev); g.manual_seed(12345)
bad_total = 0
for r0 in range(0, rows, rpb):
    r1 = min(rows, r0 + rpb)
    rid = torch.arange(r0 + 1, r1 + 1, device=dev, dtype=torch.int64)
    x = as_i32(fmix32(rid).repeat_interleave(npr))
    y = as_i32(fmix32(1 + torch.randint(0, 13000000, (x.numel(),), device=dev, generator=g, dtype=torch.int64)))
    got = torch.empty_like(x)
    m.apply_batch_dev(OP_GET, x.numel(), x.data_ptr(), y.data_ptr(), None, got.data_ptr(), st)
    xs = as_i32(fmix32(rid)); lens = torch.empty(xs.numel(), dtype=torch.int32, device=dev)
    m.rowlen_batch_dev(xs.numel(), xs.data_ptr(), lens.data_ptr(), st); torch.cuda.synchronize()
    k = (x.long() & 0xFFFFFFFF) << 32 | (y.long() & 0xFFFFFFFF)
    uk, inv, cnts = torch.unique(k, return_inverse=True, return_counts=True)
    wrong = int((got.long() != cnts[inv]).sum())
    zero = int((got == 0).sum())
    lsum = int(lens.long().sum())
    if wrong or lsum != uk.numel():
        bad_total += 1
        print("batch row0=%d: ops with wrong get %d (zero: %d), rowlen sum %d vs distinct %d" % (r0, wrong, zero, lsum, uk.numel()))
        badop = torch.nonzero(got.long() != cnts[inv]).squeeze(1)[:5].tolist()
        for j in badop:
            xx = int(x[j]) & 0xFFFFFFFF; yy = int(y[j]) & 0xFFFFFFFF
            info = m.row_info(xx); sl = m.row_slots(xx)
            import numpy as np
            present = bool((sl[:, 0] == yy).any()) if sl is not None else None
            print("    op %d (x=%d,y=%d) get=%d want=%d row(size,used)=%s key-present-in-row=%s nonempty=%s" % (
                j, xx, yy, int(got[j]), int(cnts[inv][j]), info, present, None if sl is None else int(((sl[:,0]!=0)|(sl[:,1]!=0)).sum())))
print("bad rows total", bad_total, m.stats()["rounds"])
