"""where a bench step's wall time goes on the host: incr call (blocks: the round loop reads counters back), get call
(asynchronous), final sync.  python tools/probe/step_host_times.py [profile 0/1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_INCR
prof = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = 1 << 24
dev = torch.device("cuda", 0)
gen = Stream("zipf", 12345, 1000000, 1.1, int(os.environ.get("SMX_SCRAMBLE", "1")))
xs = torch.empty((24, B), dtype=torch.int32, device=dev); ys = torch.empty_like(xs)
stream = torch.cuda.current_stream().cuda_stream
for s in range(24):
    gen.fill_device(s * B, B, xs[s].data_ptr(), ys[s].data_ptr(), stream)
ones = torch.ones(B, dtype=torch.int32, device=dev); o1 = torch.empty(B, dtype=torch.int32, device=dev); o2 = torch.empty_like(o1)
m = SparseMatrix(); m.reserve(8 << 30)
m.profile(bool(prof))
torch.cuda.synchronize()
rows = []
for s in range(24):
    t0 = time.perf_counter()
    m.apply_batch_dev(OP_INCR, B, xs[s].data_ptr(), ys[s].data_ptr(), ones.data_ptr(), o1.data_ptr(), stream)
    t1 = time.perf_counter()
    m.apply_batch_dev(OP_GET, B, xs[s].data_ptr(), ys[s].data_ptr(), None, o2.data_ptr(), stream)
    t2 = time.perf_counter()
    if len(sys.argv) > 2:
        torch.cuda.synchronize()
    t3 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2))
torch.cuda.synchronize()
for s in (2, 5, 10, 15, 20, 23):
    print("step %2d: incr call %.3f ms, get call %.3f ms, sync %.3f ms" % ((s,) + tuple(v * 1e3 for v in rows[s])))
import numpy as np
a = np.array(rows[10:]) * 1e3
print("mean steps 10..23: incr %.3f get %.3f sync %.3f  total %.3f ms/step" % (a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a.sum(1).mean()))
st = m.stats()
print("kernel ms incr %.3f get %.3f" % (st["kernel_ms_incr"] / st["kernel_launches_incr"], st["kernel_ms_get"] / max(st["kernel_launches_get"], 1)))
