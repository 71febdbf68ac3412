"""config 3's full-row scan with the k_getrow variants (SMATRIX_GETROW_VARIANT): one process per variant, same box.
   python tools/probe/getrow_variants.py [rows]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = sys.argv[1] if len(sys.argv) > 1 else "13000000"
CODE = r'''
import sys, time
sys.path.insert(0, %r)
import torch, bench
from libsmatrix_amd import SparseMatrix
dev = torch.device("cuda", 0)
rows = %s
m = SparseMatrix()
bench.build_cf(torch, dev, m, rows)
r = bench.scan_cf(torch, dev, m, rows, 5)
print("getrow %%.3f ms  %%.1f Gnnz/s  ok=%%s" %% (r["getrow_ms"], r["Gnnz_per_s"], r.get("verified_sum_of_values_eq_ops")))
m.close()
'''
names = {0: "default: 2 steps ahead, XCD-renumbered", 1: "1 step ahead, plain numbering (round 2)", 2: "1 step ahead, XCD", 3: "4 steps ahead, XCD",
         4: "2 steps ahead, plain", 5: "2 ahead, XCD, NO pair stores", 6: "2 ahead, XCD, NO cell loads"}
for v in (1, 2, 0, 4, 3, 5, 6):
    out = subprocess.run([sys.executable, "-c", CODE % (ROOT, rows)], env=dict(os.environ, SMATRIX_GETROW_VARIANT=str(v)),
                         capture_output=True, text=True)
    print("variant %d (%s): %s" % (v, names[v], (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)
