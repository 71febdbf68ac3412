"""Randomized differential run of the cold start of dense-id batches (insert_pending_keys: keys below their row's size first, free
home cells, walkers by the far join) against the oracle -- by hand on a GPU box: python tests/cold_soak.py [configurations] [seed].
Per configuration: a handful of rows (4..300), two or three batches of 2^16..2^21 incr / decr ops of Zipf ranks -- plain, shifted by a
random base (small keys that are not small), or mixed with hashed ids --, the later batches on top of the tables the first ones
left.  Per-key return multisets, gets, row sizes / used counters, the cells as a set and the probe invariant (no empty cell
inside a probe sequence) must be the oracle's.  Prints COLD_SOAK_OK and the number of cold starts that ran."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("SMATRIX_COLD_MIN", "4096")


def check_rows(g, o, tag):
    rows = o.list_rows()
    assert (g.m.rowlen_batch(rows) == np.array([o.rowlen(int(r)) for r in rows], dtype=np.uint32)).all(), (tag, "rowlen")
    for r in rows.tolist():
        assert g.row_info(r) == o.row_info(r), (tag, r, g.row_info(r), o.row_info(r))
        a = np.asarray(g.row_slots(r)); b = np.asarray(o.row_slots(r))
        ne = (a[:, 0] != 0) | (a[:, 1] != 0)
        ka = a[ne]; kb = b[(b[:, 0] != 0) | (b[:, 1] != 0)]
        ka = ka[np.lexsort((ka[:, 1], ka[:, 0]))]; kb = kb[np.lexsort((kb[:, 1], kb[:, 0]))]
        assert ka.shape == kb.shape and (ka == kb).all(), (tag, r, "cells")
        size = a.shape[0]; pos = np.flatnonzero(ne); empties = np.flatnonzero(~ne)
        home = a[pos, 0].astype(np.int64) & (size - 1)
        nxt = empties[np.searchsorted(empties, home) % empties.size]
        assert (((nxt - home) % size) > ((pos - home) % size)).all(), (tag, r, "an empty cell inside a probe sequence")


def run(configs=8, seed=1):
    from libsmatrix_amd import Stream
    from tests.gpu_adapter import GpuMatrix
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    cold = 0
    for c in range(configs):
        nrows = int(rng.choice([4, 12, 48, 300]))
        kind = int(rng.integers(0, 3))
        base = int(rng.integers(1, 1 << 20)) if kind == 1 else 0
        gen = Stream("zipf", int(rng.integers(1, 1 << 30)), int(rng.choice([100000, 1000000])), 1.1, 0)
        if os.environ.get("COLD_SOAK_TRACE") == str(c): os.environ["SMATRIX_TRACE_ROUNDS"] = "1"     # (the library reads it at open)
        else: os.environ.pop("SMATRIX_TRACE_ROUNDS", None)
        g, o = GpuMatrix(), O.Oracle()
        at = 0
        for k in range(int(rng.integers(2, 4))):
            n = 1 << int(rng.integers(16, 22))
            x, y = gen.fill(at, n); at += n
            x = (x % nrows).astype(np.uint32)
            y = (y.astype(np.uint64) + base).astype(np.uint32)
            if kind == 2:                      # a third of the ops name hashed ids
                h = rng.integers(0, 3, n) == 0
                y = np.where(h, (y * np.uint32(2654435761)) | np.uint32(1), y).astype(np.uint32)
            op = 3 if (k == 1 and rng.integers(0, 2)) else 2
            if os.environ.get("COLD_SOAK_TRACE") == str(c): print("config %d batch %d: %d rows, kind %d, base %d, n %d, op %d" % (c, k, nrows, kind, base, n, op), file=sys.stderr, flush=True)
            v = np.full(n, int(rng.integers(1, 4)), np.uint32)      # (one amount per batch: per-key return multisets are then order-free)
            a, b = g.apply(op, x, y, v), o.apply(op, x, y, v)
            kk = x.astype(np.uint64) << np.uint64(32) | y
            assert (a[np.lexsort((a, kk))] == b[np.lexsort((b, kk))]).all(), (c, k, "returns")
            assert (g.apply(0, x, y) == o.apply(0, x, y)).all(), (c, k, "gets")
            check_rows(g, o, (c, k))
        cold += g.stats()["cold_starts"]
        print("config %d: %d rows, kind %d, base %d: ok (%d cold starts so far)" % (c, nrows, kind, base, cold), flush=True)
        g.close(); o.close(); gen.close()
    print("COLD_SOAK_OK configurations=%d cold_starts=%d" % (configs, cold))


if __name__ == "__main__":
    t0 = time.time()
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("%.0f s" % (time.time() - t0))
