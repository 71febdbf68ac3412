"""one incr batch of n ops (default 2^31 + 2^27) into an empty matrix, rounds traced: python tools/diag/big_batch.py [n]"""
import os, sys, time
os.environ.setdefault("SMATRIX_TRACE_ROUNDS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import SparseMatrix, OP_GET, OP_INCR
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 31) + (1 << 27)
dev = torch.device("cuda", 0)
R = n // 1024
x = torch.empty(n, dtype=torch.int32, device=dev); y = torch.empty_like(x)
step = 1 << 28
for a in range(0, n, step):
    i = torch.arange(a, min(n, a + step), dtype=torch.int64, device=dev)
    x[a:a + i.numel()] = (i % R).to(torch.int32)
    y[a:a + i.numel()] = (i // R + 1).to(torch.int32)
    del i
ones = torch.ones(n, dtype=torch.int32, device=dev); out = torch.zeros(n, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
m = SparseMatrix()
torch.cuda.synchronize(); t0 = time.perf_counter()
m.apply_batch_dev(OP_INCR, n, x.data_ptr(), y.data_ptr(), ones.data_ptr(), out.data_ptr(), stream)
torch.cuda.synchronize()
print("incr %.2f s, out min %d max %d, stats %s" % (time.perf_counter() - t0, int(out.min()), int(out.max()), {k: v for k, v in m.stats().items() if k in ("rows", "rounds", "rows_grown", "dir_grown", "bulk_ops")}), flush=True)
