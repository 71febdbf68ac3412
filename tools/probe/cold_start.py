"""First batches of the config-2 stream, step by step (wall clock per step, rounds, cold-start statistics)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_INCR
dev = torch.device("cuda:0")
B = 1 << 24
gen = Stream("zipf", 12345, 1000000, 1.1, 1)
xs = torch.empty((6, B), dtype=torch.int32, device=dev); ys = torch.empty((6, B), dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for i in range(6): gen.fill_device(i * B, B, xs[i].data_ptr(), ys[i].data_ptr(), s)
torch.cuda.synchronize()
for rep in range(2):
    m = SparseMatrix(); m.reserve(8 << 30)
    ones = torch.ones(B, dtype=torch.int32, device=dev); o1 = torch.empty(B, dtype=torch.int32, device=dev); o2 = torch.empty(B, dtype=torch.int32, device=dev)
    chk = 0; prev = m.stats()
    for i in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.apply_batch_dev(OP_INCR, B, xs[i].data_ptr(), ys[i].data_ptr(), ones.data_ptr(), o1.data_ptr(), s)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        m.apply_batch_dev(OP_GET, B, xs[i].data_ptr(), ys[i].data_ptr(), None, o2.data_ptr(), s)
        torch.cuda.synchronize()
        chk = (chk * 31 + int(o2.to(torch.int64).sum().item()) + 7 * int(o1.to(torch.int64).sum().item())) % (1 << 61)
        st = m.stats()
        print("rep %d batch %d: incr %.2f ms  rounds +%d  cold_starts +%d keys +%d  rows %d" % (rep, i, (t1 - t0) * 1e3, st["rounds"] - prev["rounds"],
              st["cold_starts"] - prev["cold_starts"], st["cold_keys"] - prev["cold_keys"], st["rows"]), flush=True)
        prev = st
    print("   checksum", chk, flush=True)
    m.close()
