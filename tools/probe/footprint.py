#!/usr/bin/env python3
"""GPU probe: random-touch rates as a function of the footprint (TLB / L2 / MALL reach)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
T = 1 << 26
sink = torch.zeros(2, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
names = {0: "read8", 1: "atomic_ret", 2: "atomic_noret", 3: "chain2"}
for mb in (16, 64, 256, 1024, 4096, 16384):
    buf = torch.zeros(mb * (1 << 20) // 8, dtype=torch.int64, device=dev)
    line = "%6d MB:" % mb
    for mode in (0, 3, 1, 2):
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.smx_probe_random_dev(buf.data_ptr(), buf.numel() * 8, T, mode, 99 + rep, sink.data_ptr(), st)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        line += "  %s %6.1f G/s" % (names[mode], T / best / 1e6)
    print(line)
    del buf
