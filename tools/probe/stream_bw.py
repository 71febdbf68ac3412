#!/usr/bin/env python3
"""GPU probe: streaming bandwidth of this box (the ceiling getrow and the file paths are priced against)."""
import torch
dev = torch.device("cuda", 0)
for gb in (1, 4, 8):
    n = gb * (1 << 30) // 8
    a = torch.empty(n, dtype=torch.int64, device=dev).fill_(3); b = torch.empty_like(a)
    for name, fn, moved in (("copy (read+write)", lambda: b.copy_(a), 2 * n * 8), ("read-only sum", lambda: a.sum(), n * 8),
                            ("fill (write only)", lambda: b.fill_(1), n * 8)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        print("%2d GiB  %-20s %7.2f TB/s" % (gb, name, moved * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e12))
    del a, b
