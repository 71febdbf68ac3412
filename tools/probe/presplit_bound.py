#!/usr/bin/env python3
"""GPU probe: what would a key-hash multi-split pre-pass buy the incr kernel?  (round 2)

The config-2 stream is replayed with every batch PRE-PERMUTED by the top `bits` bits of a hash of (x, y) --
the permutation is computed with torch outside the timed region, so the numbers below are the UPPER BOUND a
free pre-pass would reach: a tile of the LDS-folding kernel then sees one 2^-bits slice of the key space and
folds all of a key's duplicates.  Reported per variant: average k_apply_agg<INCR> and k_apply<GET> launch time
(HIP events inside the library), whole step time, and the table's shape (must be the same for all variants).

    python tools/probe/presplit_bound.py [steps]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_INCR

SEED, N_IDS, ZIPF_S, B = 12345, 1000000, 1.1, 1 << 24
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
gen = Stream("zipf", SEED, N_IDS, ZIPF_S, 1)
xs = torch.empty((steps, B), dtype=torch.int32, device=dev); ys = torch.empty_like(xs)
for k in range(steps):
    gen.fill_device(k * B, B, xs[k].data_ptr(), ys[k].data_ptr(), stream)
torch.cuda.synchronize()
ones = torch.ones(B, dtype=torch.int32, device=dev)
o1 = torch.empty(B, dtype=torch.int32, device=dev); o2 = torch.empty_like(o1)


def permuted(kind, bits):
    if kind == "none":
        return xs, ys
    px = torch.empty_like(xs); py = torch.empty_like(ys)
    for k in range(steps):
        x = xs[k].to(torch.int64) & 0xFFFFFFFF; y = ys[k].to(torch.int64) & 0xFFFFFFFF
        if kind == "key":
            h = ((x * 0x9E3779B1) ^ (y * 0x85EBCA77)) & 0xFFFFFFFF
            h = (h ^ (h >> 15)) * 0x2C1B3C6D & 0xFFFFFFFF
            h = h ^ (h >> 12)
        else:   # "row": all ops of a row land in one slice (directory slot and table lines shared)
            h = (x * 0x9E3779B1) & 0xFFFFFFFF
            h = (h ^ (h >> 15)) * 0x2C1B3C6D & 0xFFFFFFFF
            h = h ^ (h >> 12)
        b = (h & 0xFFFFFFFF) >> (32 - bits)
        order = torch.sort(b, stable=True).indices
        px[k] = xs[k][order]; py[k] = ys[k][order]
    return px, py


def run(kind, bits):
    px, py = permuted(kind, bits)
    m = SparseMatrix()
    warm = 2
    m.profile(True)
    for k in range(steps):
        if k == warm:
            m.profile(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
        m.apply_batch_dev(OP_INCR, B, px[k].data_ptr(), py[k].data_ptr(), ones.data_ptr(), o1.data_ptr(), stream)
        m.apply_batch_dev(OP_GET, B, px[k].data_ptr(), py[k].data_ptr(), None, o2.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = m.stats()
    n = steps - warm
    print("%-5s bits=%2d  step %.3f ms  incr kernel %.3f ms  get kernel %.3f ms  rounds %d deferred %d  rows %d nnz %d" % (
        kind, bits, dt / n * 1e3, st["kernel_ms_incr"] / max(st["kernel_launches_incr"], 1),
        st["kernel_ms_get"] / max(st["kernel_launches_get"], 1), st["rounds"], st["deferred_ops"], st["rows"],
        st.get("nnz", -1)), flush=True)
    m.close()
    del px, py


run("none", 0)
for bits in (6, 10, 13, 16):
    run("key", bits)
for bits in (10, 13):
    run("row", bits)
