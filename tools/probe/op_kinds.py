#!/usr/bin/env python3
"""GPU probe: throughput of the four op kinds on the config-2 stream (2^24-op batches, table grown by 12 incr batches first):
get, incr, decr, set -- with and without a result array."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_SET, OP_INCR, OP_DECR
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
B = 1 << 24
gen = Stream("zipf", 12345, 1000000, 1.1, 1)
m = SparseMatrix()
x = torch.empty((16, B), dtype=torch.int32, device=dev); y = torch.empty_like(x)
for k in range(16):
    gen.fill_device(k * B, B, x[k].data_ptr(), y[k].data_ptr(), st)
ones = torch.ones(B, dtype=torch.int32, device=dev); out = torch.empty_like(ones)
vals = torch.randint(1, 1000, (B,), dtype=torch.int32, device=dev)
for k in range(12):
    m.apply_batch_dev(OP_INCR, B, x[k].data_ptr(), y[k].data_ptr(), ones.data_ptr(), out.data_ptr(), st)
torch.cuda.synchronize()


def timed(op, k, v, o):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.apply_batch_dev(op, B, x[k].data_ptr(), y[k].data_ptr(), v, o, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for name, op, v in (("get", OP_GET, None), ("incr", OP_INCR, ones.data_ptr()), ("decr", OP_DECR, ones.data_ptr()), ("set", OP_SET, vals.data_ptr())):
    # old keys (batch 3: all present) and new keys (batches 12..15: ~25 % new cells, rows grow)
    a = min(timed(op, 3, v, out.data_ptr()) for _ in range(3))
    b = timed(op, 12 + (op % 4), v, out.data_ptr())
    c = min(timed(op, 3, v, None) for _ in range(3)) if op != OP_GET else float("nan")
    print("%-5s present keys %.3f ms (%.1f G ops/s) | batch with new keys %.3f ms | present keys, no result array %.3f ms" % (name, a, B / a / 1e6, b, c), flush=True)
