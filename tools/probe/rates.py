#!/usr/bin/env python3
"""GPU probe: op-kernel times on a pre-built config-2 table, separating hits from inserts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_INCR

B = 1 << 24
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
gen = Stream("zipf", 12345, 1000000, 1.1, 1)
ugen = Stream("uniform", 777, 1 << 22, 1.1, 1)
xs = torch.empty((NB + 1, B), dtype=torch.int32, device=dev); ys = torch.empty_like(xs)
ones = torch.ones(B, dtype=torch.int32, device=dev); out = torch.empty(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for s in range(NB + 1):
    gen.fill_device(s * B, B, xs[s].data_ptr(), ys[s].data_ptr(), st)
ux = torch.empty(B, dtype=torch.int32, device=dev); uy = torch.empty_like(ux)
ugen.fill_device(0, B, ux.data_ptr(), uy.data_ptr(), st)
torch.cuda.synchronize()
m = SparseMatrix()

def run(op, x, y, label):
    m.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.apply_batch_dev(op, B, x.data_ptr(), y.data_ptr(), ones.data_ptr() if op else None, out.data_ptr(), st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = m.stats(); k = "incr" if op else "get"
    print("%-44s wall %8.3f ms   round-0 kernel %8.3f ms   rounds=%d deferred=%d" % (
        label, dt * 1e3, s["kernel_ms_" + k], s["rounds"], s["deferred_ops"]))

for s in range(NB):
    run(OP_INCR, xs[s], ys[s], "zipf incr batch %d (growing)" % s)
run(OP_INCR, xs[NB - 1], ys[NB - 1], "zipf incr: repeat last batch (all hits)")
run(OP_INCR, xs[0], ys[0], "zipf incr: repeat batch 0 (all hits)")
run(OP_GET, xs[NB - 1], ys[NB - 1], "zipf get last batch")
run(OP_INCR, xs[NB], ys[NB], "zipf incr next new batch")
run(OP_INCR, ux, uy, "uniform 4M x 4M incr (all new rows+cells)")
run(OP_INCR, ux, uy, "uniform incr repeat (all hits, no dups)")
run(OP_GET, ux, uy, "uniform get")
print(m.stats())
