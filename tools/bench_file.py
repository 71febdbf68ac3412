#!/usr/bin/env python3
"""Config 5 shape on one GPU: build a CF-shaped matrix (rows x ~115 nnz), close (= persist in the
reference's file format), reopen (= bulk load into HBM), verify, re-run the get and getrow benches.
Prints one JSON line.  Default 1M rows (~2 GB file); --rows 13000000 is the 27 GB full scale."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libsmatrix_amd import SparseMatrix, OP_INCR, OP_GET
from tools.bench_getrow import fmix32, as_i32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1000000)
    ap.add_argument("--nnz-per-row", type=int, default=115)
    ap.add_argument("--path", default="/tmp/smx_bench_file.smx")
    a = ap.parse_args()
    dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
    if os.path.exists(a.path):
        os.remove(a.path)
    g = torch.Generator(device=dev); g.manual_seed(12345)
    m = SparseMatrix(a.path)
    rows_per_batch = (1 << 24) // a.nnz_per_row
    keep = None
    for r0 in range(0, a.rows, rows_per_batch):
        r1 = min(a.rows, r0 + rows_per_batch)
        rid = torch.arange(r0 + 1, r1 + 1, device=dev, dtype=torch.int64)
        x = as_i32(fmix32(rid).repeat_interleave(a.nnz_per_row))
        y = as_i32(fmix32(1 + torch.randint(0, 13000000, (x.numel(),), device=dev, generator=g, dtype=torch.int64)))
        ones = torch.ones_like(x); out = torch.empty_like(x)
        m.apply_batch_dev(OP_INCR, x.numel(), x.data_ptr(), y.data_ptr(), ones.data_ptr(), out.data_ptr(), st)
        if keep is None:
            keep = (x.clone(), y.clone())
        last = (x, y)
    torch.cuda.synchronize()
    print("built", m.stats(), file=sys.stderr, flush=True)
    want = torch.empty_like(keep[0])
    m.apply_batch_dev(OP_GET, want.numel(), keep[0].data_ptr(), keep[1].data_ptr(), None, want.data_ptr(), st)
    torch.cuda.synchronize()
    want_last = torch.empty_like(last[0])
    m.apply_batch_dev(OP_GET, want_last.numel(), last[0].data_ptr(), last[1].data_ptr(), None, want_last.data_ptr(), st)
    xs0 = as_i32(fmix32(torch.arange(1, a.rows + 1, device=dev, dtype=torch.int64)))
    lens0 = torch.empty(a.rows, dtype=torch.int32, device=dev)
    m.rowlen_batch_dev(a.rows, xs0.data_ptr(), lens0.data_ptr(), st)
    torch.cuda.synchronize()
    rows_before = m.stats()["rows"]
    print("closing", file=sys.stderr, flush=True)
    t0 = time.perf_counter(); m.close(); t_close = time.perf_counter() - t0
    print("closed", file=sys.stderr, flush=True)
    fbytes = os.path.getsize(a.path)
    t0 = time.perf_counter(); m = SparseMatrix(a.path); t_open = time.perf_counter() - t0
    print("reopened", m.stats(), file=sys.stderr, flush=True)
    got = torch.empty_like(want)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    m.apply_batch_dev(OP_GET, got.numel(), keep[0].data_ptr(), keep[1].data_ptr(), None, got.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    got_last = torch.empty_like(want_last)
    m.apply_batch_dev(OP_GET, got_last.numel(), last[0].data_ptr(), last[1].data_ptr(), None, got_last.data_ptr(), st)
    torch.cuda.synchronize()
    ok = bool(torch.equal(got, want)) and bool(torch.equal(got_last, want_last)) and m.stats()["rows"] == rows_before
    print("gets after reopen done", file=sys.stderr, flush=True)
    xs = as_i32(fmix32(torch.arange(1, a.rows + 1, device=dev, dtype=torch.int64)))
    lens = torch.empty(a.rows, dtype=torch.int32, device=dev)
    m.rowlen_batch_dev(a.rows, xs.data_ptr(), lens.data_ptr(), st)
    off = torch.zeros(a.rows + 1, dtype=torch.int64, device=dev); torch.cumsum(lens.long() + 1, 0, out=off[1:])
    ret = torch.empty((int(off[-1].item()), 2), dtype=torch.int32, device=dev); cnt = torch.empty(a.rows, dtype=torch.int32, device=dev)
    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e2.record(); m.getrow_batch_dev(a.rows, xs.data_ptr(), off.data_ptr(), ret.data_ptr(), cnt.data_ptr(), st); e3.record()
    torch.cuda.synchronize()
    nnz = int(cnt.long().sum().item())
    ok = ok and bool(torch.equal(lens, lens0)) and nnz == int(lens0.long().sum().item())
    print("sum rowlen before %d after %d getrow %d" % (int(lens0.long().sum()), int(lens.long().sum()), nnz), file=sys.stderr)
    print(json.dumps({"metric": "file-backed round trip", "rows": a.rows, "nnz": nnz, "file_bytes": fbytes,
                      "close_s": t_close, "write_GBps": fbytes / t_close / 1e9, "open_s": t_open,
                      "load_GBps": fbytes / t_open / 1e9, "verified": ok,
                      "get_after_reopen_Gops": got.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9,
                      "getrow_after_reopen_Gnnz": nnz / (e2.elapsed_time(e3) * 1e-3) / 1e9}))
    # do not rewrite the file again on exit: drop the handle without the flush (bench only)
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    main()
