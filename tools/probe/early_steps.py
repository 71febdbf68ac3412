"""per-step wall time of the first batches of config 2 (young table): python tools/probe/early_steps.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_INCR
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = 1 << 24
dev = torch.device("cuda", 0)
gen = Stream("zipf", 12345, 1000000, 1.1, 1)
xs = torch.empty((nb, B), dtype=torch.int32, device=dev); ys = torch.empty_like(xs)
stream = torch.cuda.current_stream().cuda_stream
for s in range(nb):
    gen.fill_device(s * B, B, xs[s].data_ptr(), ys[s].data_ptr(), stream)
ones = torch.ones(B, dtype=torch.int32, device=dev); o1 = torch.empty(B, dtype=torch.int32, device=dev); o2 = torch.empty_like(o1)
m = SparseMatrix(); m.reserve(8 << 30)
torch.cuda.synchronize()
for s in range(nb):
    st0 = m.stats()
    t0 = time.perf_counter()
    m.apply_batch_dev(OP_INCR, B, xs[s].data_ptr(), ys[s].data_ptr(), ones.data_ptr(), o1.data_ptr(), stream)
    m.apply_batch_dev(OP_GET, B, xs[s].data_ptr(), ys[s].data_ptr(), None, o2.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = m.stats()
    print("step %2d: %.3f ms  rounds %d deferred %d grown %d bulk_ops %d chains %d" % (s, dt * 1e3, st["rounds"] - st0["rounds"],
          st["deferred_ops"] - st0["deferred_ops"], st["rows_grown"] - st0["rows_grown"], st["bulk_ops"] - st0["bulk_ops"], st["spec_chains"] - st0["spec_chains"]), flush=True)
