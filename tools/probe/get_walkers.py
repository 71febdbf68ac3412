#!/usr/bin/env python3
"""(round 6) Who walks in the get batches of the dense-id stream, and for how long.

Needs a library built with -DSMX_GET_WALK_TIMES (k_get_clu then returns, for every op that had to walk with its wave, the duration
of the walk in 10 ns ticks + 0xFFF00000 instead of the value):

    cd libsmatrix_amd/csrc && touch smx_runtime.hip && make HIPCC="/opt/rocm/bin/hipcc -DSMX_GET_WALK_TIMES"
    python tools/probe/get_walkers.py [steps=14]        # (SMATRIX_HINT_LG=22 / 24 / 26: the table's size)
    touch smx_runtime.hip && make                       # back to the product

Per step: walkers, the distribution of their walks, and for the long ones (> 50 us) what the incr batch before had returned for
the same op (1 = the key was new in this batch) and the rows they belong to.  Results: profiles/r06_get_walkers.txt."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_INCR
dev = torch.device("cuda:0"); B = 1 << 24
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 14
gen = Stream("zipf", 12345, 1000000, 1.1, 0)
x = torch.empty(B, dtype=torch.int32, device=dev); y = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
m = SparseMatrix(); m.reserve(8 << 30)
ones = torch.ones(B, dtype=torch.int32, device=dev); o1 = torch.empty(B, dtype=torch.int32, device=dev); o2 = torch.empty(B, dtype=torch.int32, device=dev)
for i in range(steps):
    gen.fill_device(i * B, B, x.data_ptr(), y.data_ptr(), s)
    m.apply_batch_dev(OP_INCR, B, x.data_ptr(), y.data_ptr(), ones.data_ptr(), o1.data_ptr(), s)
    m.apply_batch_dev(OP_GET, B, x.data_ptr(), y.data_ptr(), None, o2.data_ptr(), s)
    torch.cuda.synchronize()
    u = o2.to(torch.int64) & 0xFFFFFFFF
    w = u >= 0xFFF00000
    dd = (u & 0xFFFFF).double() / 100.0       # us (a 100 MHz counter)
    ds = torch.sort(dd[w]).values
    n = ds.numel()
    if n == 0:
        print("step %d: no walkers (a library without -DSMX_GET_WALK_TIMES?)" % i); continue
    print("step %d walkers %d | walk us: sum %.0f mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f | >20us %d >50us %d >100us %d" % (
        i, n, float(ds.sum()), float(ds.mean()), float(ds[n // 2]), float(ds[int(n * 0.9)]), float(ds[int(n * 0.99)]), float(ds[-1]),
        int((ds > 20).sum()), int((ds > 50).sum()), int((ds > 100).sum())), flush=True)
    lw = w & (dd > 50)
    if int(lw.sum()) == 0: continue
    o1l = (o1.to(torch.int64) & 0xFFFFFFFF)[lw]
    ux, cx = torch.unique(x.to(torch.int64)[lw], return_counts=True)
    top = torch.topk(cx, min(8, cx.numel()))
    print("   long (>50us) %d: incr returned 1: %d, <=4: %d, <=16: %d | rows %d, top x:count %s" % (
        int(lw.sum()), int((o1l == 1).sum()), int((o1l <= 4).sum()), int((o1l <= 16).sum()), ux.numel(),
        " ".join("%d:%d" % (int(ux[j]), int(c)) for c, j in zip(top.values, top.indices))), flush=True)
m.close()
