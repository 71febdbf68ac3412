#!/usr/bin/env python3
"""Counterpart of the reference's benchmark driver (src/smatrix_benchmark.c).

    smatrix_benchmark.py {full,incr,get} [times] [threads] [file]      (argv as :182-208)

Restates the stock workload exactly: thread t uses o = 42+t and runs, `user1 = times/T` times,
for n<23, i<22: incr(n+o, i+o, 1); incr(i+o, n+o, 1)   (src/smatrix_benchmark.c:29-46; get: :48-65),
printing the reference's table (:134-138) -- one row of T=1..32 per backend:

  hip-batch  the same op multiset as ONE batched call per thread count
  hip-scalar the drop-in scalar ABI, T host threads on one handle (one device round trip per op)

The batched line is what the GPU path is for; the scalar line documents the cost of the
one-op-per-call ABI.  The reference's own row of the table is printed by
tests/stock_benchmark_reference.py (the compiled reference is checker code and stays under tests/)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def pattern(threadn, user1):
    o = 42 + threadn
    n, i = np.meshgrid(np.arange(23, dtype=np.uint32), np.arange(22, dtype=np.uint32), indexing="ij")
    a, b = (n + o).ravel(), (i + o).ravel()
    x = np.stack([a, b], 1).ravel(); y = np.stack([b, a], 1).ravel()       # incr(n+o,i+o); incr(i+o,n+o)
    return np.tile(x, user1), np.tile(y, user1)


def measure(fn, nthreads):
    t0 = time.perf_counter()
    th = [threading.Thread(target=fn, args=(t,)) for t in range(nthreads)]
    [t.start() for t in th]; [t.join() for t in th]
    return (time.perf_counter() - t0) * 1e3


def main():
    test = sys.argv[1] if len(sys.argv) > 1 else "full"
    times = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    only_t = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    fname = sys.argv[4] if len(sys.argv) > 4 else None
    scalar_cap = int(os.environ.get("SMX_BENCH_SCALAR_OPS", "0")) or (1 << 62)     # 0 = the full op count
    Ts = [only_t] if only_t else [1, 2, 4, 8, 16, 32]
    import libsmatrix_amd
    from libsmatrix_amd import SparseMatrix
    if not libsmatrix_amd.device_available():
        sys.exit("no HIP device (the product has no CPU path)")
    backends = [("hip-batch", lambda: SparseMatrix(fname)), ("hip-scalar", lambda: SparseMatrix(fname))]
    for op in (["incr", "get"] if test == "full" else [test]):
        print("TEST: %s" % ("1 million x mixed " + op))
        print("-" * 63)
        print("%-11s" % "" + "".join("T=%-9d" % t for t in Ts))
        for name, mk in backends:
            m = mk()
            cells = []
            for T in Ts:
                user1 = max(times // T, 1)
                if name == "hip-batch":
                    xs, ys = zip(*(pattern(t, user1) for t in range(T)))
                    x, y = np.concatenate(xs), np.concatenate(ys)
                    t0 = time.perf_counter()
                    (m.incr_batch(x, y, np.ones_like(x)) if op == "incr" else m.get_batch(x, y))
                    cells.append("%.1fms" % ((time.perf_counter() - t0) * 1e3))
                else:
                    per = min(max(scalar_cap // T, 1), user1 * 1012)   # optionally a bounded sample, scaled to the full op count
                    def fn(t):
                        x, y = pattern(t, 1)
                        f = (lambda a, b: m.incr(a, b, 1)) if op == "incr" else m.get
                        for k in range(per):
                            f(int(x[k % x.size]), int(y[k % y.size]))
                    ms = measure(fn, T)
                    cells.append(("%.1fms" if per == user1 * 1012 else "~%.0fms") % (ms * (user1 * 1012.0 / per)))
            print("%-11s" % name + "".join("%-11s" % c for c in cells))
            m.close()
        print()


if __name__ == "__main__":
    main()
