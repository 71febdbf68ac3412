#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import SparseMatrix, OP_INCR, OP_GET
from tools.bench_getrow import fmix32, as_i32
rows = 2500000; npr = 115
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
for mode in sys.argv[1:]:
    path = None if mode == "mem" else "/tmp/bc.smx"
    if path and os.path.exists(path): os.remove(path)
    g = torch.Generator(device=dev); g.manual_seed(12345)
    m = SparseMatrix(path)
    rpb = (1 << 24) // npr
    tot_distinct = 0
    for r0 in range(0, rows, rpb):
        r1 = min(rows, r0 + rpb)
        rid = torch.arange(r0 + 1, r1 + 1, device=dev, dtype=torch.int64)
        x = as_i32(fmix32(rid).repeat_interleave(npr))
        y = as_i32(fmix32(1 + torch.randint(0, 13000000, (x.numel(),), device=dev, generator=g, dtype=torch.int64)))
        ones = torch.ones_like(x); out = torch.empty_like(x)
        m.apply_batch_dev(OP_INCR, x.numel(), x.data_ptr(), y.data_ptr(), ones.data_ptr(), out.data_ptr(), st)
        if mode == "clone":
            keep = (x.clone(), y.clone())
        torch.cuda.synchronize()
        k = (x.long() & 0xFFFFFFFF) << 32 | (y.long() & 0xFFFFFFFF)
        tot_distinct += torch.unique(k).numel()
        xs = as_i32(fmix32(rid)); lens = torch.empty(xs.numel(), dtype=torch.int32, device=dev)
        m.rowlen_batch_dev(xs.numel(), xs.data_ptr(), lens.data_ptr(), st); torch.cuda.synchronize()
        got = int(lens.long().sum())
        want = torch.unique(k).numel()
        if got != want:
            print(mode, "batch at row", r0, "rowlen sum", got, "distinct", want, "rounds", m.stats()["rounds"])
    print(mode, "total distinct", tot_distinct, m.stats())
    os._exit(0) if False else None
