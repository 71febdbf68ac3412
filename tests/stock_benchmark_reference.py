#!/usr/bin/env python3
"""The REFERENCE's row of the stock benchmark table (src/smatrix_benchmark.c:134-176), measured with the
compiled reference (oracle/_ref): T real threads, one C call per thread, thread create/join inside the
timed region like :109-122.  Checker-side companion of tools/smatrix_benchmark.py (which prints the HIP rows).
usage: stock_benchmark_reference.py [times]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from oracle import oracle as O
from smatrix_benchmark import pattern, measure

times = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Ts = [1, 2, 4, 8, 16, 32]
for op in ("incr", "get"):
    m = O.Reference()
    cells = []
    for T in Ts:
        user1 = max(times // T, 1)
        def fn(t):
            x, y = pattern(t, user1)
            m.apply(O.OP_INCR if op == "incr" else O.OP_GET, x, y, np.ones_like(x))
        cells.append("%.1fms" % measure(fn, T))
    print("%-6s reference  " % op + "".join("%-11s" % c for c in cells))
    m.close()
