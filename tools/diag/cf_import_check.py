import sys, numpy as np
sys.path.insert(0, '/root/repo')
from tests.gpu_adapter import GpuMatrix
from oracle import oracle as O
from libsmatrix_amd import Stream
S, L = int(sys.argv[1]), 12
gen = Stream("zipf", 12352, 1000000, 1.1, 1)
ids, _ = gen.fill(0, S * L)
sessions = ids.reshape(S, L)
g, o = GpuMatrix(), O.Oracle()
g.m.cf_import_sessions([r for r in sessions])
for r in sessions:
    O.cf_import_preference_set(o, r)
items = np.unique(ids)
z = np.zeros_like(items)
a = g.apply(0, items, z); b = o.apply(0, items, z)
print("rows", g.stats()["rows"], o.num_rows(), "totals equal", bool((a == b).all()), int(a.astype(np.uint64).sum()), int(b.astype(np.uint64).sum()), S * L)
bad = np.nonzero(a != b)[0][:5]
for k in bad.tolist():
    print("item", items[k], "gpu", a[k], "oracle", b[k], "rowinfo", g.row_info(int(items[k])), o.row_info(int(items[k])))
