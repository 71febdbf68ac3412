"""The dense-id variant of the config-2 stream (bench.py: dense_ids_leg), step by step; under rocprofv3 with
tools/probe/kernel_sums_window.py it gives the kernel time by name of the steady steps."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_INCR
dev = torch.device("cuda:0")
B = 1 << 24
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gen = Stream("zipf", 12345, 1000000, 1.1, 0)
xs = torch.empty((steps, B), dtype=torch.int32, device=dev); ys = torch.empty_like(xs)
s = torch.cuda.current_stream().cuda_stream
for i in range(steps): gen.fill_device(i * B, B, xs[i].data_ptr(), ys[i].data_ptr(), s)
torch.cuda.synchronize()
m = SparseMatrix(); m.reserve(8 << 30)
ones = torch.ones(B, dtype=torch.int32, device=dev); o1 = torch.empty(B, dtype=torch.int32, device=dev); o2 = torch.empty(B, dtype=torch.int32, device=dev)
chk = 0; ts = []
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.apply_batch_dev(OP_INCR, B, xs[i].data_ptr(), ys[i].data_ptr(), ones.data_ptr(), o1.data_ptr(), s)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    m.apply_batch_dev(OP_GET, B, xs[i].data_ptr(), ys[i].data_ptr(), None, o2.data_ptr(), s)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    chk = (chk * 31 + int(o2.to(torch.int64).sum().item()) + 7 * int(o1.to(torch.int64).sum().item())) % (1 << 61)
st = m.stats()
print("incr/get ms per step:", " ".join("%.1f/%.1f" % t for t in ts))
print("last 8 steps: incr %.3f ms  get %.3f ms" % (sum(a for a, b in ts[-8:]) / 8, sum(b for a, b in ts[-8:]) / 8))
print("mean of steps 2..: %.2f ms  (%.3f G ops/s)  rounds %d long_probe_rounds %d  checksum %d" % (
    sum(a + b for a, b in ts[2:]) / (steps - 2), 2 * B / (sum(a + b for a, b in ts[2:]) / (steps - 2)) / 1e6, st["rounds"], st["long_probe_rounds"], chk), flush=True)
m.close()
