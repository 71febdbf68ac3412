"""GPU (-m gpu): BASELINE.json's configs under the driver's eyes, pinned to numbers the REFERENCE holds.

* config 2 -- the Zipf(1.1)^2 stream at 10^7 ops: SURVEY.md A.4 (measured on the compiled reference):
  sum over the stream of get(x_i,y_i) = 52 480 898 544, 561 596 rows, 4 463 637 nnz, hottest row 159 472;
  tests/test_oracle_golden.py holds the oracle to the same figures on the CPU.
* config 3 -- the CF-recommender shape (13 M rows x ~115 nnz): oracle-compared at 1 M rows
  (SURVEY.md 6 scale), properties + per-row oracle samples at the full 13 M rows (1.5 G nnz, 27 GB of row tables).
* config 5 (one GPU's worth) -- that matrix closed, reopened and re-verified through the reference's file format.
* parity holes of round 1: per-op return values of BATCHED incr/decr under heavy duplication (the LDS-folding
  kernel), the reference's unchanged benchmark binary through the shim object, `self->mem`.
All through the C ABI (ctypes); bit-exact."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from libsmatrix_amd.stream import Stream

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CF_COLS, CF_PER_ROW, SEED = 13000000, 115, 12345
CF13M_NNZ = 1494993467   # distinct (row, column) cells of the 13 M-row CF stream (np.unique over the host generator, 500k rows at a time)


@pytest.fixture(scope="module")
def G():
    from tests.gpu_adapter import GpuMatrix
    import libsmatrix_amd
    assert libsmatrix_amd.device_available(), "no HIP device: the product has no CPU fallback"
    return GpuMatrix


def per_key_sorted(x, y, ret):
    k = x.astype(np.uint64) << 32 | y.astype(np.uint64)
    o = np.lexsort((ret, k))
    return k[o], ret[o]


def expected_size(n):
    """src/smatrix.c:346: a row that received n distinct keys (y >= 1) has the smallest 16*2^k slots with n <= 8*2^k + 1"""
    size = 16
    while n > size // 2 + 1:
        size *= 2
    return size


# ---------------------------------------------------------------------------------------------------
def test_config2_reference_checksums_1e7(G):
    """SURVEY.md A.4, numbers produced by the unmodified reference: 10^6 ops -> 576 561 nnz / 137 116 rows
    (max row 28 558); 10^7 ops -> 4 463 637 nnz / 561 596 rows (max row 159 472), sum get = 52 480 898 544"""
    gen = Stream("zipf", SEED, 1000000, 1.1, 1)
    x, y = gen.fill(0, 10000000)
    m = G()
    ones = np.ones(1000000, np.uint32)
    m.apply(2, x[:1000000], y[:1000000], ones)
    rows = np.unique(x[:1000000])
    lens = m.m.rowlen_batch(rows)
    assert m.stats()["rows"] == rows.size == 137116
    assert int(lens.astype(np.uint64).sum()) == 576561 and int(lens.max()) == 28558
    rest = 9000000
    m.apply(2, x[1000000:], y[1000000:], np.ones(rest, np.uint32))          # one 9M-op batch: the LDS-folding kernel
    assert m.sum_get(x, y) == 52480898544
    rows = np.unique(x)
    lens = m.m.rowlen_batch(rows)
    assert m.stats()["rows"] == rows.size == 561596
    assert int(lens.astype(np.uint64).sum()) == 4463637 and int(lens.max()) == 159472
    # every row's size is the reference's function of its rowlen (src/smatrix.c:346), checked on the getrow side too
    off, pairs, cnt = m.m.getrow_batch(rows[:20000])
    assert (cnt == lens[:20000]).all()
    for r in (0, 1, 17, 4242, 19999):
        assert m.row_info(int(rows[r])) == (expected_size(int(lens[r])), int(lens[r]))
    m.close()
    gen.close()


def test_config2_reference_checksums_4e8():
    """BASELINE config 2 at FULL size: exactly 4e8 ops of the Zipf(1.1)^2 stream (generated on the device, 24 batches of
    2^24) on a fresh matrix must leave 1 000 000 rows, 100 401 767 nnz and a hottest row of 935 410 columns -- the figures
    the unmodified reference produced in 263 s on one CPU thread -- plus the 10^7-op figures on the way.
    (bench.py runs the same replay after its timed region.)"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    res = bench.verify_config2(torch, dev, None, None, 1 << 24, 0)
    assert res["matches_reference"]
    assert res["at_4e8_ops"] == {"rows": 1000000, "nnz": 100401767, "max_rowlen": 935410}
    assert res["at_1e7_ops"]["sum_get"] == 52480898544
    # the handful of probe sequences beyond the folding kernel's budget that any 10^5-cell table at load 1/2 has must not
    # be taken for a clustered (dense-id) table: that mode runs a wave per deferred op (round 3: it was, 2.5 -> 3.7 ms per step)
    assert res["clustered_mode"] == 0
    g = res["getrow_all_rows_at_4e8"]                  # one getrow call over all rows, the 2 M-slot row in 64 segments
    assert g["pairs"] == 100401767 and g["sum_of_values"] == 400000000 and g["max_row_pairs"] == 935410
    assert g["counts_equal_rowlen"]


def test_batched_returns_under_duplication(G, oracle_mod):
    """k_apply_agg<INCR/DECR> (batches >= 1024 ops fold duplicate keys in LDS): the per-op RETURN values.
    With one increment value per key the multiset of a key's returns is the same in every serialisation
    (old + v, old + 2v, ...), so it must equal the oracle's; wrap-around below zero included."""
    rng = np.random.default_rng(12)
    g, o = G(), oracle_mod.Oracle()
    for rnd, (op, n, nx, ny) in enumerate(((2, 50000, 50, 40), (3, 50000, 50, 40), (3, 70000, 50, 40),
                                           (2, 1 << 20, 3000, 300), (3, 1 << 20, 3000, 300), (2, 4096, 2, 3))):
        x = rng.integers(0, nx, n, dtype=np.uint32)
        y = rng.integers(1, ny, n, dtype=np.uint32)
        v = ((x * 7 + y * 13) % 5).astype(np.uint32)                          # one value per key, 0 included
        v[(x + y) % 11 == 0] = 0xFFFFFFF0                                       # ... and near-wrap values
        a, b = g.apply(op, x, y, v), o.apply(op, x, y, v)
        ka, ra = per_key_sorted(x, y, a)
        kb, rb = per_key_sorted(x, y, b)
        assert (ra == rb).all(), (rnd, op)
        assert (g.apply(0, x, y) == o.apply(0, x, y)).all(), rnd
    rows = o.list_rows().tolist()
    for r in rows[:300]:
        assert g.row_info(r) == o.row_info(r)
    g.close(); o.close()


def test_mixed_value_returns_contain_final_value(G, oracle_mod):
    """mixed increments per key: the return multiset depends on the serialisation, but in EVERY serialisation
    the last op of a key returns the key's final value -- so get(key) must be among the key's returns"""
    rng = np.random.default_rng(3)
    g = G()
    n = 200000
    for op in (2, 3, 2):
        x = rng.integers(0, 100, n, dtype=np.uint32); y = rng.integers(1, 80, n, dtype=np.uint32)
        v = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        ret = g.apply(op, x, y, v)
        fin = g.apply(0, x, y)
        k = x.astype(np.uint64) << 32 | y
        uk, inv = np.unique(k, return_inverse=True)
        hit = np.zeros(uk.size, dtype=bool)
        np.logical_or.at(hit, inv, ret == fin)
        assert hit.all()
    g.close()


def test_mem_field_tracks_the_tables(G):
    """examples/smatrix_example.c:72 reads db->mem (src/smatrix.c:151-166 keeps it): non-zero from open on,
    never shrinking while cells are only added; SURVEY A.1: the directory alone is 65536 x 16 B"""
    m = G()
    seen = [m.m.mem]
    assert seen[0] >= 65536 * 16
    rng = np.random.default_rng(1)
    for n in (1, 10, 1000, 100000, 1000000):
        x = rng.integers(0, 1 << 20, n, dtype=np.uint32); y = rng.integers(1, 1 << 20, n, dtype=np.uint32)
        m.apply(2, x, y, np.ones(n, np.uint32))
        seen.append(m.m.mem)
    assert all(b >= a for a, b in zip(seen, seen[1:])) and seen[-1] > seen[0], seen
    nnz = int(m.m.rowlen_batch(np.arange(1 << 20, dtype=np.uint32)).astype(np.uint64).sum())
    assert seen[-1] >= 8 * nnz                                               # at least the cells themselves
    m.incr(5, 5, 1)
    assert m.m.mem >= seen[-1]
    m.close()


def test_c_program_known_answers(tmp_path, oracle_mod):
    """a plain C caller (tests/c/abi_known_answers.c) compiled against include/*.h and linked with lib/smatrix.so: the
    known answers of SURVEY.md A.1 through the eight drop-in calls, the batch API interleaved with scalar calls, flush;
    in file mode the oracle then reads what the C program left"""
    inc, lib = os.path.join(ROOT, "include"), os.path.join(ROOT, "libsmatrix_amd", "lib")
    exe = str(tmp_path / "abi_known_answers")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I" + inc, os.path.join(ROOT, "tests", "c", "abi_known_answers.c"),
                    os.path.join(lib, "smatrix.so"), "-Wl,-rpath," + lib, "-o", exe], check=True)
    for args in ([], [str(tmp_path / "c_caller.smx")]):
        p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "C_ABI_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    r = oracle_mod.Oracle(str(tmp_path / "c_caller.smx"))
    assert r.get(1, 2) == 17 and r.rowlen(3) == 12 and r.get(77, 1000) == 11 and r.rowlen(77) == 1000
    assert r.get(2, 0) == 1 and r.rowlen(2) == 3          # the (0,v) cell is counted after a reload (quirk Q2)
    r.close()


def test_scalar_abi_cell_mirror(G, oracle_mod, tmp_path):
    """The scalar ABI answers calls on cells it has already seen from a host-side mirror and writes the values back
    in one batched set before anything else can look (DESIGN.md "scalar ABI"): every return value, every batch read
    interleaved with it, getrow, a y = 0 write that cuts a probe chain (quirk Q3), and the file written at close must
    be exactly the oracle's -- and the calls must not each cost a device round trip."""
    import time
    rng = np.random.default_rng(6)
    path, path_o = str(tmp_path / "mirror.smx"), str(tmp_path / "mirror_oracle.smx")
    g, o = G(path), oracle_mod.Oracle(path_o)
    cells = [(int(a), int(b)) for a, b in zip(rng.integers(0, 30, 400), rng.integers(1, 500, 400))]
    t0 = time.perf_counter()
    for rep in range(60):                                # 24 000 scalar calls on 400 cells
        for (a, b) in cells:
            op = ("incr", "decr", "set", "get")[(a + b + rep) % 4]
            args = (a, b) if op == "get" else (a, b, (a * b + rep) % 7)
            assert getattr(g, op)(*args) == getattr(o, op)(*args)
    dt = time.perf_counter() - t0
    st = g.stats()
    assert st["scalar_cache_hits"] > 20000 and dt < 20.0, (st["scalar_cache_hits"], dt)
    xs = np.array([c[0] for c in cells], np.uint32); ys = np.array([c[1] for c in cells], np.uint32)
    assert (g.apply(0, xs, ys) == o.apply(0, xs, ys)).all()                    # batch read sees the mirrored values
    assert g.stats()["scalar_cache_flushes"] >= 1
    for r in range(30):
        assert (g.getrow(r) == o.getrow(r)).all() and g.row_info(r) == o.row_info(r)
        assert (np.asarray(g.row_slots(r)) == np.asarray(o.row_slots(r))).all()   # byte-identical: only values were deferred
    v = rng.integers(1, 5, xs.size, dtype=np.uint32)
    g.apply(2, xs, ys, v); o.apply(2, xs, ys, v)                                 # a batch write: the mirror must not go stale
    for (a, b) in cells[:100]:
        assert g.get(a, b) == o.get(a, b) and g.incr(a, b, 3) == o.incr(a, b, 3)
    # quirk Q3 through the mirror: (0,v) appears in slot 0, a key piles up behind it, (0,v) goes back to empty ->
    # the key behind the cut is no longer found by the reference; a later write makes a twin
    for op, a in (("incr", (77, 0, 1)), ("incr", (77, 16, 4)), ("get", (77, 16)), ("incr", (77, 16, 1)), ("decr", (77, 0, 1)),
                  ("get", (77, 16)), ("incr", (77, 16, 5)), ("get", (77, 16)), ("incr", (77, 0, 2)), ("get", (77, 16)),
                  ("rowlen", (77,))):
        assert getattr(g, op)(*a) == getattr(o, op)(*a), (op, a)
    assert (np.asarray(g.row_slots(77)) == np.asarray(o.row_slots(77))).all()
    for (a, b) in cells[:50]:
        g.incr(a, b, 1); o.incr(a, b, 1)                                         # left dirty in the mirror on purpose
    g.close()                                                                    # close writes the mirror back, then the file
    o.close()
    # both files through the same reader (the reference's load rule drops value-0 cells, quirk Q4 -- the same on both
    # sides because the scalar stream left byte-identical tables)
    back, want = oracle_mod.Oracle(path), oracle_mod.Oracle(path_o)
    assert (back.apply(0, xs, ys) == want.apply(0, xs, ys)).all()
    assert [back.rowlen(r) for r in range(30)] + [back.rowlen(77)] == [want.rowlen(r) for r in range(30)] + [want.rowlen(77)]
    for r in list(range(30)) + [77]:
        assert (np.asarray(back.row_slots(r)) == np.asarray(want.row_slots(r))).all()
    back.close(); want.close()


def test_scalar_abi_mirror_with_threads(G):
    """T threads hammering overlapping cells through the scalar ABI (src/smatrix_benchmark.c:29-46 pattern): the sum of
    all increments must arrive, whatever mixture of mirrored and device-path calls served them"""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from smatrix_benchmark import pattern
    m = G()
    T, reps = 8, 6

    def work(t):
        x, y = pattern(t, reps)
        for a, b in zip(x.tolist(), y.tolist()):
            m.incr(a, b, 1)
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [t.start() for t in th]; [t.join() for t in th]
    want = {}
    for t in range(T):
        x, y = pattern(t, reps)
        for a, b in zip(x.tolist(), y.tolist()):
            want[(a, b)] = want.get((a, b), 0) + 1
    ks = list(want)
    got = m.apply(0, np.array([k[0] for k in ks], np.uint32), np.array([k[1] for k in ks], np.uint32))
    assert got.tolist() == [want[k] for k in ks]
    assert m.stats()["scalar_cache_hits"] > 0
    m.close()


@pytest.mark.parametrize("over_rccl", [False, True])
def test_native_router_one_rank(G, oracle_mod, tmp_path, monkeypatch, over_rccl):
    """include/smatrix_shard.h router section, the C library's own multi-GPU path, with all a 1-GPU box allows: one
    rank.  The batch still goes through the partition kernels, the (self-)exchange, the packed in-place apply and the
    result gather; answers and tables must be the oracle's, apply_then_get must equal apply + get, a planned placement
    must not change anything, and the shard's file must be readable by the oracle.  over_rccl: the rank's own share
    travels through RCCL as well (dlopen'ed library, ncclCommInitRank, grouped ncclSend/ncclRecv to itself) instead of
    a device copy -- the call path N > 1 uses, as far as one GPU can take it."""
    import torch
    if over_rccl:
        monkeypatch.setenv("SMATRIX_SHARD_FORCE_RCCL", "1")
    from libsmatrix_amd.sharded import NativeShardedMatrix, Placement
    rng = np.random.default_rng(23)
    dev = torch.device("cuda", 0)
    path = str(tmp_path / "shard0.smx")
    sh, o = NativeShardedMatrix(path, rank=0, world=1), oracle_mod.Oracle()
    sh.set_placement(Placement(1, None, {5: 0, 77: 0, 123456: 0}))
    t = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    for rnd, (op, n) in enumerate(((2, 300000), (2, 100000), (1, 50000), (2, 1 << 21))):      # (no decr: a value-0 cell is dropped
        # by the reference's load rule, quirk Q4, and WHICH chains that cuts depends on the batch layout)
        x = rng.integers(0, 20000, n, dtype=np.uint32); y = rng.integers(1, 5000, n, dtype=np.uint32)
        v = ((x + y) % 4 + 1).astype(np.uint32)                     # one value per key: order-free returns
        dx, dy, dv = t(x), t(y), t(v)
        out = torch.empty(n, dtype=torch.int32, device=dev); outg = torch.empty_like(out)
        if rnd % 2 == 0:
            sh.apply_then_get_dev(op, dx, dy, dv, out, outg)
        else:
            sh.apply_dev(op, dx, dy, dv, out)
            sh.apply_dev(0, dx, dy, None, outg)
        torch.cuda.synchronize()
        want = o.apply(op, x, y, v)
        got = out.cpu().numpy().view(np.uint32)
        if op == 1:
            assert (got == v).all()
        else:
            ka = np.lexsort((got, x.astype(np.uint64) << 32 | y)); kb = np.lexsort((want, x.astype(np.uint64) << 32 | y))
            assert (got[ka] == want[kb]).all(), rnd
        assert (outg.cpu().numpy().view(np.uint32) == o.apply(0, x, y)).all(), rnd
    assert sh.exchanged_ops > 0
    rows = o.list_rows()
    assert (sh.local.rowlen_batch(rows) == np.array([o.rowlen(int(r)) for r in rows], np.uint32)).all()
    assert int(sh.local.stats()["rows"]) == rows.size
    sh.close()
    back = oracle_mod.Oracle(path)
    assert [back.rowlen(int(r)) for r in rows[:500]] == [o.rowlen(int(r)) for r in rows[:500]]
    back.close(); o.close()


def test_incremental_flush_writes_dirty_rows_only(G, oracle_mod, tmp_path, monkeypatch):
    """Persistence the reference's way (src/smatrix.c:418-496, :744-788): smatrix_flush writes the rows that changed
    since the last flush -- in place when the table kept its size, as a fresh block with a re-pointed CMAP entry when
    it grew, as a new CMAP entry when the row is new -- and nothing else; a process that dies without smatrix_close
    leaves the state of the last flush (copy of the file taken while the matrix is still open); every stage is read
    back by the oracle (and by the compiled reference where present) and by this library's own loader."""
    import shutil
    monkeypatch.setenv("SMATRIX_FLUSH_MS", "0")              # explicit flushes only: the counts below are exact
    monkeypatch.setenv("SMATRIX_EXPERIMENTAL", "1")          # smatrix_compact (no reference counterpart) is used at the end
    rng = np.random.default_rng(17)
    path = str(tmp_path / "inc.smx")
    g, o = G(path), oracle_mod.Oracle()
    readers = [oracle_mod.Oracle] + ([oracle_mod.Reference] if oracle_mod.have_reference() else [])

    def check(tag, qx, qy):
        snap = str(tmp_path / ("snap_%s.smx" % tag))
        shutil.copy(path, snap)                              # what a crash right now would leave behind
        size = os.path.getsize(snap)                         # (before any reader touches the copy)
        want = o.apply(0, qx, qy)
        rows = np.unique(qx)
        for R in readers:
            r = R(snap)
            assert (r.apply(0, qx, qy) == want).all(), (tag, R.__name__)
            assert [r.rowlen(int(a)) for a in rows[:300]] == [o.rowlen(int(a)) for a in rows[:300]], (tag, R.__name__)
            r.close()
        return size

    x = rng.integers(0, 5000, 300000, dtype=np.uint32); y = rng.integers(1, 1 << 20, 300000, dtype=np.uint32)
    v = rng.integers(1, 9, x.size, dtype=np.uint32)
    g.apply(2, x, y, v); o.apply(2, x, y, v)
    g.m.flush()
    st = g.stats()
    assert st["file_flushes"] == 1 and st["file_rows_written"] == 5000
    size1 = check("first", x, y)
    g.m.flush()                                              # nothing changed: nothing written
    assert g.stats()["file_rows_written"] == 5000
    # values only, in 40 rows: rewritten in place, the file keeps its length
    hot = np.nonzero(x < 40)[0]
    g.apply(2, x[hot], y[hot], v[hot]); o.apply(2, x[hot], y[hot], v[hot])
    g.m.flush()
    assert g.stats()["file_rows_written"] == 5000 + 40
    assert check("inplace", x, y) == size1
    # new keys: rows 100..199 grow (fresh blocks, entries re-pointed), rows 9000..9049 are new (new CMAP entries);
    # a scalar call on a mirrored cell rides along (the flush writes the mirror back first)
    x2 = np.concatenate([rng.integers(100, 200, 40000, dtype=np.uint32), rng.integers(9000, 9050, 5000, dtype=np.uint32)])
    y2 = rng.integers(1, 1 << 20, x2.size, dtype=np.uint32)
    g.apply(2, x2, y2, np.ones_like(x2)); o.apply(2, x2, y2, np.ones_like(x2))
    assert g.incr(7, 123456789, 5) == o.incr(7, 123456789, 5) and g.incr(7, 123456789, 2) == o.incr(7, 123456789, 2)
    g.m.flush()
    written = g.stats()["file_rows_written"] - 5040
    assert 150 <= written <= 151 + 0, written               # 100 grown + 50 new + row 7
    qx, qy = np.concatenate([x, x2, [7]]).astype(np.uint32), np.concatenate([y, y2, [123456789]]).astype(np.uint32)
    size3 = check("grown", qx, qy)
    assert size3 > size1
    # the 100 grown rows left their old blocks behind (the reference leaks them too, src/smatrix.c:430-436);
    # smatrix_compact rewrites the file without them: smaller, same contents, and incremental flushes go on
    assert g.stats()["file_leaked_bytes"] > 100 * (16 + 8 * 16)
    leaked = g.stats()["file_leaked_bytes"]
    g.m.compact()
    size4 = check("compacted", qx, qy)
    assert (size4, size3 - leaked, g.stats()["file_compactions"], g.stats()["file_leaked_bytes"]) == (size4, size4, 1, 0), \
        (size1, size3, size4, leaked)
    g.apply(2, x2[:500], y2[:500] + 3, np.ones(500, np.uint32)); o.apply(2, x2[:500], y2[:500] + 3, np.ones(500, np.uint32))
    g.m.flush()
    check("after_compaction", np.concatenate([qx, x2[:500]]), np.concatenate([qy, y2[:500] + 3]))
    # unflushed work is lost by a crash, but the file stays the last checkpoint
    g.apply(2, x[:1000], y[:1000], v[:1000])
    snap = str(tmp_path / "crash.smx"); shutil.copy(path, snap)
    r = oracle_mod.Oracle(snap); assert (r.apply(0, qx, qy) == o.apply(0, qx, qy)).all(); r.close()
    o.apply(2, x[:1000], y[:1000], v[:1000])
    g.close()                                                # the last flush
    # our own loader takes the file (leaked blocks, re-pointed entries and all), writes on, flushes incrementally again
    g = G(path)
    assert (g.apply(0, qx, qy) == o.apply(0, qx, qy)).all()
    g.apply(2, x2[:3000], y2[:3000] + 7, np.ones(3000, np.uint32)); o.apply(2, x2[:3000], y2[:3000] + 7, np.ones(3000, np.uint32))
    g.m.flush()
    assert 0 < g.stats()["file_rows_written"] <= 150
    check("reloaded", np.concatenate([qx, x2[:3000]]), np.concatenate([qy, y2[:3000] + 7]))
    g.close(); o.close()


# ---------------------------------------------------------------------------------------------------
def test_reference_benchmark_binary_through_the_shim(oracle_mod, tmp_path):
    """The reference's UNCHANGED src/smatrix_benchmark.c, compiled on the build box against include/smatrix.h and
    linked with lib/smatrix.o (oracle/Makefile -> oracle/_ref/smatrix_benchmark_hip), run HERE on the GPU: the shim
    object dlopen()s smatrix.so, T pthreads hammer one handle through the scalar ABI (src/smatrix_benchmark.c:29-46,
    :109-122), the file argument makes smatrix_close persist the result, and the oracle reads it back."""
    exe = os.path.join(ROOT, "oracle", "_ref", "smatrix_benchmark_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/smatrix_benchmark_hip not built (needs /root/reference on the build box)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from smatrix_benchmark import pattern
    env = dict(os.environ, SMATRIX_HIP_LIB=os.path.join(ROOT, "libsmatrix_amd", "lib", "smatrix.so"))
    for times, T in ((64, 4), (3, 1)):
        path = str(tmp_path / ("stock_%d.smx" % T))
        p = subprocess.run([exe, "incr", str(times), str(T), path], capture_output=True, text=True, timeout=900, env=env)
        assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
        # (the reference's format string holds a non-breaking space after the '@', src/smatrix_benchmark.c:213)
        assert "testing: %dk x incr" % times in p.stdout and "%d threads:" % T in p.stdout and "ms" in p.stdout
        p = subprocess.run([exe, "get", str(times), str(T), path], capture_output=True, text=True, timeout=900, env=env)
        assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]          # reopen + get through the shim
        want = oracle_mod.Oracle()
        for t in range(T):
            x, y = pattern(t, times // T)
            want.apply(2, x, y, np.ones_like(x))
        got = oracle_mod.Oracle(path)
        rows = want.list_rows().tolist()
        assert sorted(got.list_rows().tolist()) == sorted(rows)
        for r in rows:
            assert got.row_info(r) == want.row_info(r)
            a, b = got.getrow(r), want.getrow(r)
            assert sorted(map(tuple, a.tolist())) == sorted(map(tuple, b.tolist()))
        got.close(); want.close()


# ---------------------------------------------------------------------------------------------------
def _cf_build_device(m, rows, torch, chunk_rows=1 << 17):
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    gen = Stream("cf", SEED, CF_COLS, float(CF_PER_ROW), 1)
    n_max = chunk_rows * CF_PER_ROW
    x = torch.empty(n_max, dtype=torch.int32, device=dev); y = torch.empty_like(x)
    ones = torch.ones_like(x); out = torch.empty_like(x)
    for r0 in range(0, rows, chunk_rows):
        n = (min(rows, r0 + chunk_rows) - r0) * CF_PER_ROW
        gen.fill_device(r0 * CF_PER_ROW, n, x.data_ptr(), y.data_ptr(), st)
        # (every other chunk without a result array: d_out may be NULL, include/smatrix_batch.h)
        m.apply_batch_dev(2, n, x.data_ptr(), y.data_ptr(), ones.data_ptr(), out.data_ptr() if (r0 // chunk_rows) % 2 == 0 else None, st)
    torch.cuda.synchronize()
    return gen


def _cf_row_checks(m, gen, oracle_mod, row_numbers):
    """each CF row's ops are contiguous in the stream: rebuild single rows in the oracle and compare everything"""
    for rn in row_numbers:
        x, y = gen.fill(rn * CF_PER_ROW, CF_PER_ROW)
        o = oracle_mod.Oracle()
        o.apply(2, x, y, np.ones(CF_PER_ROW, np.uint32))
        X = int(x[0])
        assert (x == X).all()
        assert m.row_info(X) == o.row_info(X), rn
        a, b = np.asarray(m.row_slots(X)), np.asarray(o.row_slots(X))
        sa = a[np.lexsort((a[:, 1], a[:, 0]))]; sb = b[np.lexsort((b[:, 1], b[:, 0]))]
        assert (sa == sb).all(), rn
        pairs = m.getrow_raw(X, (m.getRowLength(X) + 1) * 8)
        ne = a[(a[:, 0] != 0) | (a[:, 1] != 0)]
        assert (pairs == ne).all(), rn                                          # slot order
        o.close()


def test_config3_cf_shape_1m_rows_vs_oracle(G, oracle_mod):
    """SURVEY.md 6 scale of config 3 (1 M rows x 115 ops, uniform columns over 13 M, scrambled ids): the whole
    matrix against the oracle -- rowlen of EVERY row, getrow content of sampled rows, sum get over the stream"""
    rows = 1000000
    n = rows * CF_PER_ROW
    gen = Stream("cf", SEED, CF_COLS, float(CF_PER_ROW), 1)
    x, y = gen.fill(0, n)
    g, o = G(), oracle_mod.Oracle()
    chunk = 1 << 24
    for a in range(0, n, chunk):
        ones = np.ones(min(chunk, n - a), np.uint32)
        g.apply(2, x[a:a + chunk], y[a:a + chunk], ones)
        o.apply(2, x[a:a + chunk], y[a:a + chunk], ones)
    xs = x[::CF_PER_ROW].copy()
    assert g.stats()["rows"] == o.num_rows() == rows
    lens_g = g.m.rowlen_batch(xs)
    lens_o = np.array([o.rowlen(int(r)) for r in xs[::97]], dtype=np.uint32)
    assert (lens_g[::97] == lens_o).all()
    nnz = int(lens_g.astype(np.uint64).sum())
    assert nnz == np.unique(x.astype(np.uint64) << 32 | y).size
    assert g.sum_get(x[: 1 << 24], y[: 1 << 24]) == o.sum_get(x[: 1 << 24], y[: 1 << 24])
    sel = np.random.default_rng(4).integers(0, rows, 400)
    off, pairs, cnt = g.m.getrow_batch(xs[sel])
    for i, r in enumerate(sel.tolist()):
        mine = pairs[off[i]: off[i] + cnt[i]]
        theirs = o.getrow(int(xs[r]))
        assert cnt[i] == theirs.shape[0] == lens_g[r]
        assert (mine[np.lexsort((mine[:, 1], mine[:, 0]))] == theirs[np.lexsort((theirs[:, 1], theirs[:, 0]))]).all()
        assert g.row_info(int(xs[r])) == o.row_info(int(xs[r]))
    g.close(); o.close(); gen.close()


def _fmix32_np(ids):
    h = ids.astype(np.uint32).copy()
    h ^= h >> 16; h *= np.uint32(0x85ebca6b); h ^= h >> 13; h *= np.uint32(0xc2b2ae35); h ^= h >> 16
    return h


def _cf_scan(m, rows, torch):
    """rowlen + getrow over ALL rows on the device -> (nnz, key checksum, value sum, max rowlen, min rowlen)"""
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    xs = torch.from_numpy(_fmix32_np(np.arange(1, rows + 1, dtype=np.uint32)).view(np.int32)).to(dev)
    lens = torch.empty(rows, dtype=torch.int32, device=dev)
    m.rowlen_batch_dev(rows, xs.data_ptr(), lens.data_ptr(), st)
    off = torch.zeros(rows + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens.long() + 1, 0, out=off[1:])                 # caller idiom: rowlen, then a buffer (+1: quirk Q5)
    total = int(off[-1].item())
    ret = torch.zeros((total, 2), dtype=torch.int32, device=dev)
    cnt = torch.empty(rows, dtype=torch.int32, device=dev)
    m.getrow_batch_dev(rows, xs.data_ptr(), off.data_ptr(), ret.data_ptr(), cnt.data_ptr(), st)
    torch.cuda.synchronize()
    assert bool((cnt == lens).all())
    out = (int(lens.sum(dtype=torch.int64).item()), int(ret[:, 0].sum(dtype=torch.int64).item()),
           int(ret[:, 1].sum(dtype=torch.int64).item()), int(lens.max().item()), int(lens.min().item()))
    del ret
    return out


def _cf_full(rows, path, oracle_mod, torch):
    from libsmatrix_amd import SparseMatrix
    m = SparseMatrix(path)
    gen = _cf_build_device(m, rows, torch)
    assert m.stats()["rows"] == rows
    nnz, ksum, vsum, lmax, lmin = _cf_scan(m, rows, torch)
    # every op is +1, so the values sum to the op count; a repeated (row, column) draw merges two ops into one cell
    assert vsum == rows * CF_PER_ROW and rows * CF_PER_ROW * 0.9995 <= nnz <= rows * CF_PER_ROW
    assert lmax <= CF_PER_ROW and lmin >= CF_PER_ROW - 5
    sample = np.random.default_rng(rows).integers(0, rows, 60).tolist() + [0, rows - 1]
    _cf_row_checks(m, gen, oracle_mod, sample)
    if path:
        m.close()                                                             # persist (src/smatrix.c:113-133)
        assert os.path.getsize(path) > 8 * nnz
        m = SparseMatrix(path)                                                # bulk load (src/smatrix.c:576-596)
        assert m.stats()["rows"] == rows
        assert _cf_scan(m, rows, torch)[:3] == (nnz, ksum, vsum)
        _cf_row_checks(m, gen, oracle_mod, sample[:20])
        for rn in sample[:5]:                                                 # still writable after the reload
            X = int(gen.fill(rn * CF_PER_ROW, 1)[0][0])
            before = m.getRowLength(X)
            m.incr(X, 0xFFFFFFF0, 1)
            assert m.getRowLength(X) == before + 1
        m.close()
        os.remove(path)
    else:
        m.close()
    gen.close()
    return nnz


def test_cf_read_path_all_items_of_a_2m_row_matrix():
    """examples/cf_recommender.c:50-86 at scale: totals in column 0 (y = 0, quirk Q1: not counted by rowlen), then every item's
    neighbours scored by the fused kernel in ONE call; bench.cf_read_path re-derives 20 000 items' scores from get() results
    with the example's arithmetic in float64 and demands equality, and neighbours == nnz + one (0,total) entry per row."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from libsmatrix_amd import SparseMatrix
    dev = torch.device("cuda", 0)
    rows = 2000000
    m = SparseMatrix()
    bench.build_cf(torch, dev, m, rows)
    r = bench.cf_read_path(torch, dev, m, rows, reps=1)
    assert r["verified"] and r["items"] == rows
    assert rows * (bench.CF_PER_ROW + 1) * 0.9995 <= r["neighbours"] <= rows * (bench.CF_PER_ROW + 1)
    m.close()


def test_config3_and_5_full_scale_13m_rows(oracle_mod, tmp_path):
    """BASELINE config 3 at full size on one GPU -- 13 M rows / 1.495 G nnz / 27 GB of row tables, built on the
    device -- checked through size-independent properties (sum rowlen == pairs getrow returns, values sum to the
    op count, every sampled row identical to the oracle's rebuild of that row), and config 5's persist / close /
    reopen / verify of the same matrix in the reference's file format (where the scratch disk has room for the
    27 GB file; else the file leg runs at 2 M rows)."""
    import torch
    full_file = shutil.disk_usage(str(tmp_path)).free > 40e9
    nnz = _cf_full(13000000, str(tmp_path / "cf13m.smx") if full_file else None, oracle_mod, torch)
    assert nnz == CF13M_NNZ, nnz                      # fixed by the generator (include/smx_stream.h, SMX_DIST_CF)
    if not full_file:
        _cf_full(2000000, str(tmp_path / "cf2m.smx"), oracle_mod, torch)
