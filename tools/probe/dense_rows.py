"""What do the big rows of the dense-id stream look like?  After N steps: for the largest rows, how many cells sit at
home, how far the others are displaced, and what a doubling would do (new homes, the run of at-home cells from slot 1,
the pile behind it, walk lengths of the remaining displaced cells)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libsmatrix_amd import SparseMatrix, Stream, OP_INCR
dev = torch.device("cuda:0")
B = 1 << 24
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
gen = Stream("zipf", 12345, 1000000, 1.1, 0)
m = SparseMatrix(); m.reserve(8 << 30)
s = torch.cuda.current_stream().cuda_stream
x = torch.empty(B, dtype=torch.int32, device=dev); y = torch.empty_like(x); ones = torch.ones_like(x); o = torch.empty_like(x)
for i in range(steps):
    gen.fill_device(i * B, B, x.data_ptr(), y.data_ptr(), s)
    m.apply_batch_dev(OP_INCR, B, x.data_ptr(), y.data_ptr(), ones.data_ptr(), o.data_ptr(), s)
torch.cuda.synchronize()
for row in (1, 3, 10, 30, 60, 100, 200, 400, 800, 1500, 3000):
    size, used = m.row_info(row)[:2]
    if size < 16384: 
        print("row %d: size %d used %d (not chunked)" % (row, size, used)); continue
    sl = np.asarray(m.row_slots(row)).reshape(-1, 2)
    key = sl[:, 0].astype(np.int64); ne = (sl[:, 0] != 0) | (sl[:, 1] != 0)
    p = np.arange(size)
    home = key & (size - 1)
    disp = (p - home) % size
    at_home = ne & (disp == 0)
    dd = disp[ne & ~at_home]
    print("row %d: size %d used %d at-home %d displaced %d  displacement median %d p90 %d max %d" % (
        row, size, used, at_home.sum(), dd.size, np.median(dd) if dd.size else 0, np.percentile(dd, 90) if dd.size else 0, dd.max() if dd.size else 0))
    # the doubling
    nh = key & (2 * size - 1)
    bits = np.zeros(2 * size, bool); bits[nh[at_home]] = True
    e = 1
    while e < 2 * size and bits[e]: e += 1
    dis = ne & ~at_home
    pile = dis & (nh >= 1) & (nh < e)
    print("     doubled: run of at-home cells from slot 1 ends at %d; pile members %d of %d displaced" % (e, pile.sum(), dis.sum()))
    # serial rehash (the reference's order) to get the true walk lengths of displaced cells
    T = np.zeros(2 * size, bool); steps_walk = []
    T[nh[at_home]] = True      # (at-home cells never move: see DESIGN)
    order = np.flatnonzero(dis)
    # runs of occupied slots make the plain walk long; count steps over NOT-at-home slots only (what k_grow_move_rest examines)
    occ_nothome = np.zeros(2 * size, bool)
    for q in order[:200000]:
        i = nh[q]; st = 0
        while T[i]:
            if occ_nothome[i]: st += 1
            i = (i + 1) & (2 * size - 1)
        T[i] = True; occ_nothome[i] = True; steps_walk.append(st)
    sw = np.array(steps_walk if steps_walk else [0])
    print("     walks over displaced residents (serial order): total %d  median %d p90 %d max %d" % (sw.sum(), np.median(sw), np.percentile(sw, 90), sw.max()))
m.close()
