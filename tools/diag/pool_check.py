#!/usr/bin/env python3
"""GPU check: three config-3 builds in one process (the second and third reuse the first one's physical chunks), then the
pool is given back."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from libsmatrix_amd import SparseMatrix, _lib
dev = torch.device("cuda", 0)
for rep in range(3):
    free0 = torch.cuda.mem_get_info()[0]
    m = SparseMatrix()
    t = bench.build_cf(torch, dev, m, 13000000)
    used = free0 - torch.cuda.mem_get_info()[0]
    m.close()
    print("build %d: %.3f s, device memory taken during the build %.1f GB, free after close %.1f GB" % (rep, t, used / 1e9, torch.cuda.mem_get_info()[0] / 1e9), flush=True)
_lib.load().smatrix_release_cached_memory()
print("after release: free %.1f GB" % (torch.cuda.mem_get_info()[0] / 1e9))
