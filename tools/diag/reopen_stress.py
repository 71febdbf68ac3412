#!/usr/bin/env python3
"""GPU check: the 13 M-row file (27 GB) reopened repeatedly in one process (bulk loader + chunk pool), every time scanned"""
import os, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from libsmatrix_amd import SparseMatrix
dev = torch.device("cuda", 0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 13000000
d = tempfile.mkdtemp(prefix="smxre", dir=os.environ.get("SMX_SCRATCH", "/tmp"))
path = os.path.join(d, "cf.smx")
try:
    m = SparseMatrix(path); bench.build_cf(torch, dev, m, rows); ref = bench.scan_cf(torch, dev, m, rows, reps=1); m.close()
    for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
        m = SparseMatrix(path)
        r = bench.scan_cf(torch, dev, m, rows, reps=1)
        assert (r["nnz"], r["key_checksum"], r["verified_sum_of_values_eq_ops"]) == (ref["nnz"], ref["key_checksum"], True), k
        m.close()
        print("reopen", k, "ok", r["nnz"], flush=True)
    print("REOPEN_OK")
finally:
    shutil.rmtree(d, ignore_errors=True)
