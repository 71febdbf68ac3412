"""config 3's full-row scan (smatrix_getrow over all 13 M rows) with the request in shuffled order (the bench's) and in creation
order (rows were created in this order, so their blocks lie in ascending arena order: VERDICT r3 asked whether an arena-order
scan lets the reads stream), next to the box's plain fill / copy rates for the same byte counts.
   python tools/probe/getrow_order.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from libsmatrix_amd import SparseMatrix
dev = torch.device("cuda", 0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 13000000
def ev(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
a = torch.empty(12 << 30, dtype=torch.uint8, device=dev); b = torch.empty(12 << 30, dtype=torch.uint8, device=dev)
t = ev(lambda: a.zero_()); print("fill 12 GiB: %.2f ms = %.2f TB/s written" % (t, (12 << 30) / t / 1e9))
t = ev(lambda: b.copy_(a)); print("copy 12 GiB: %.2f ms = %.2f TB/s moved (read + write)" % (t, 2 * (12 << 30) / t / 1e9))
t = ev(lambda: a.sum(dtype=torch.int64) if False else torch.sum(a.view(torch.int64))); print("read 12 GiB (sum): %.2f ms = %.2f TB/s read" % (t, (12 << 30) / t / 1e9))
del a, b; torch.cuda.empty_cache()
m = SparseMatrix(); m.reserve(int(rows * 256 * 8 * 1.15) + (1 << 30))
bench.build_cf(torch, dev, m, rows)
stream = torch.cuda.current_stream().cuda_stream
for shuffle in (True, False):
    xs = bench.cf_row_ids(torch, dev, rows, shuffle)
    lens = torch.empty(rows, dtype=torch.int32, device=dev)
    m.rowlen_batch_dev(rows, xs.data_ptr(), lens.data_ptr(), stream)
    off = torch.zeros(rows + 1, dtype=torch.int64, device=dev); torch.cumsum(lens.long() + 1, 0, out=off[1:])
    ret = torch.zeros((int(off[-1].item()), 2), dtype=torch.int32, device=dev); cnt = torch.empty(rows, dtype=torch.int32, device=dev)
    t = ev(lambda: m.getrow_batch_dev(rows, xs.data_ptr(), off.data_ptr(), ret.data_ptr(), cnt.data_ptr(), stream))
    nnz = int(cnt.sum(dtype=torch.int64).item())
    print("getrow, request in %s order: %.3f ms  %.1f G nnz/s  (%d pairs)" % ("SHUFFLED" if shuffle else "CREATION (arena)", t, nnz / t / 1e6, nnz), flush=True)
    del ret, off, cnt, lens, xs
m.close()
