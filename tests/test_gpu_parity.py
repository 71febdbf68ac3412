"""GPU (-m gpu): the HIP path, called through the C ABI, against the golden vectors of the
real reference and against the oracle on the same seeded inputs.  Bit-exact throughout
(uint32 arithmetic); layouts are compared slot for slot wherever the call pattern is
sequential, and as (size, used, sorted cells) for batches (see include/smatrix_batch.h)."""
import hashlib
import threading

import numpy as np
import ctypes as _C
_u32p = _C.POINTER(_C.c_uint32)
import pytest

from libsmatrix_amd.stream import Stream
from tests import replay

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    from tests.gpu_adapter import GpuMatrix
    import libsmatrix_amd
    assert libsmatrix_amd.device_available(), "no HIP device: the product has no CPU fallback"
    return GpuMatrix


def sha(a):
    return hashlib.sha256(np.asarray(a).astype("<u4").tobytes()).hexdigest()


def per_key_sorted(x, y, ret):
    k = x.astype(np.uint64) << 32 | y.astype(np.uint64)
    o = np.lexsort((ret, k))
    return k[o], ret[o]


def state_equal(g, o, rows, exact_layout):
    for x in rows:
        gi, oi = g.row_info(x), o.row_info(x)
        assert gi == oi, ("row_info", x, gi, oi)
        a, b = np.asarray(g.row_slots(x)), np.asarray(o.row_slots(x))
        if exact_layout:
            assert (a == b).all(), ("layout", x)
        else:
            ka = a[(a[:, 0] != 0) | (a[:, 1] != 0)]; kb = b[(b[:, 0] != 0) | (b[:, 1] != 0)]
            ka = ka[np.lexsort((ka[:, 1], ka[:, 0]))]; kb = kb[np.lexsort((kb[:, 1], kb[:, 0]))]
            assert ka.shape == kb.shape and (ka == kb).all(), ("content", x)
            # and the table must be a valid `key % size` linear-probe layout (src/smatrix.c:363-380)
            size = a.shape[0]
            for p in np.nonzero(a[:, 0])[0]:
                q = int(a[p, 0]) % size
                while q != p:
                    assert a[q, 0] != 0 or a[q, 1] != 0, ("probe chain broken", x, int(a[p, 0]))
                    q = (q + 1) % size


# ---------------------------------------------------------------------------
def test_quirks_transcript_scalar_abi(G, golden):
    """every op a scalar C-ABI call: table bytes identical to the reference's"""
    m = G()
    bad = replay.replay_quirks(m, golden("quirks")["transcript"])
    m.close()
    assert not bad, "\n".join(bad)


def test_java_suite(G, golden):
    """src/java/test/TestSparseMatrix.java:22-131 on one shared handle; the 10^6-op loops go
    through the batched API, the rest through the scalar ABI"""
    g = golden("java_suite")
    m = G()
    m.set(42, 23, 17); assert m.get(42, 23) == 17
    m.set(4231, 2634, 0); m.incr(4231, 2634, 1); assert m.get(4231, 2634) == 1
    m.set(1231, 2634, 0); m.incr(1231, 2634, 1); m.incr(1231, 2634, 5); assert m.get(1231, 2634) == 6
    n, i = np.meshgrid(np.arange(1000, dtype=np.uint32), np.arange(1000, dtype=np.uint32), indexing="ij")
    xs, ys = i.ravel(), n.ravel()
    out = m.apply(1, xs, ys, np.full(xs.size, 34, np.uint32))
    assert (out == 34).all()
    assert (m.apply(0, xs, ys) == 34).all()
    r = np.arange(1000, dtype=np.uint32)
    for col, key in ((42, "case5_rowlen_42"), (85, "case6_rowlen_85"), (83, None)):
        for x in r.tolist()[:50]:
            m.incr(x, col, 1)                       # scalar calls
        m.apply(2, r[50:], np.full(950, col, np.uint32), np.ones(950, np.uint32))
        if key:
            assert m.rowlen(col) == g[key]
    assert m.getrow(85, m.rowlen(85) * 8).shape[0] == g["case6_getrow_85_pairs"] == 1000
    full = m.m.getRow(83)
    assert len(full) == g["case7_getrow_83_pairs"] and len(m.m.getRow(83, 230)) == 230
    assert m.getrow(85, 16 * 8).tolist() == g["row_85_first_pairs"]
    rows = sorted(set(range(1000)) | {4231, 1231})
    assert len(rows) == g["summary"]["rows"]
    assert replay.content_digest(m, rows) == g["summary"]["content_sha256"]
    # keys 0..999 in 2048-slot rows sit at their home slots whatever the order: layout is unique
    assert replay.layout_digest(m, rows) == g["summary"]["layout_sha256"]
    m.close()


@pytest.mark.parametrize("scalar", [True, False])
def test_golden_streams_small(G, golden, scalar):
    g = golden("streams")
    for case in g["cases"]:
        sm = case["small"]
        gen = Stream(case["dist"], g["seed"], case["n_ids"], g["zipf_s"], case["scramble"])
        x, y = gen.fill(0, sm["n"])
        m = G(scalar=scalar)
        ret = m.apply(2, x, y, np.ones(sm["n"], np.uint32))
        want = np.array(sm["incr_returns"], dtype=np.uint32)
        if scalar:
            assert (ret == want).all(), case["name"]
        else:
            a, b = per_key_sorted(x, y, ret), per_key_sorted(x, y, want)
            assert (a[1] == b[1]).all(), case["name"]
        assert m.sum_get(x, y) == sm["sum_get"]
        rows = sorted(set(x.tolist()))
        assert len(rows) == sm["summary"]["rows"]
        assert replay.content_digest(m, rows) == sm["summary"]["content_sha256"], case["name"]
        if scalar:
            assert replay.layout_digest(m, rows) == sm["summary"]["layout_sha256"], case["name"]
            d = np.arange(0, sm["n"], 3)
            assert sha(m.apply(3, x[d], y[d], np.ones(d.size, np.uint32))) == sm["decr_returns_sha256"]
            s = np.arange(0, sm["n"], 7)
            assert sha(m.apply(1, x[s], y[s], (s % 5).astype(np.uint32))) == sm["set_returns_sha256"]
            assert replay.layout_digest(m, rows) == sm["after_mixed"]["layout_sha256"]
            assert m.sum_get(x, y) == sm["after_mixed_sum_get"]
        m.close()
        gen.close()


def test_golden_streams_big_batched(G, golden):
    g = golden("streams")
    for case in g["cases"]:
        big = case["big"]
        gen = Stream(case["dist"], g["seed"], case["n_ids"], g["zipf_s"], case["scramble"])
        x, y = gen.fill(0, big["n"])
        m = G()
        half = big["n"] // 2
        m.apply(2, x[:half], y[:half], np.ones(half, np.uint32))          # two batches
        m.apply(2, x[half:], y[half:], np.ones(big["n"] - half, np.uint32))
        assert m.sum_get(x, y) == big["sum_get"], case["name"]
        rows = np.unique(x).tolist()
        assert len(rows) == big["summary"]["rows"]
        assert replay.content_digest(m, rows) == big["summary"]["content_sha256"], case["name"]
        m.close()
        gen.close()


@pytest.mark.parametrize("seed,nx,ny,n", [(1, 50, 40, 20000), (2, 3, 2000, 20000), (3, 5000, 5000, 60000),
                                          (4, 1, 1 << 30, 30000), (5, 200000, 64, 200000)])
def test_random_mixed_batches_vs_oracle(G, oracle_mod, seed, nx, ny, n):
    """runs of one op kind (the batch API's unit) with y=0 / value 0 / wrap-around included;
    duplicate keys inside a batch are frequent"""
    rng = np.random.default_rng(seed)
    g, o = G(), oracle_mod.Oracle()
    for rnd in range(6):
        x = rng.integers(0, nx, n, dtype=np.uint32)
        y = rng.integers(1, ny, n, dtype=np.uint32)       # y >= 1: the order-independent domain
        v = rng.integers(0, 4, n, dtype=np.uint32)
        v[rng.random(n) < 0.01] = 0xFFFFFFFF
        op = (2, 3, 1, 2, 0, 3)[rnd]
        a, b = g.apply(op, x, y, v), o.apply(op, x, y, v)
        if op == 0:
            assert (a == b).all()
        elif op == 1:
            assert (a == v).all() and (b == v).all()
        else:
            # returns are "value after the op in some serialisation": the per-key multiset is NOT order
            # independent for mixed increments, but in every serialisation the LAST op of a key returns the
            # key's final value -- the oracle's get must be among the key's returns (and is the oracle's own last)
            fin = o.apply(0, x, y)
            k = x.astype(np.uint64) << 32 | y
            uk, inv = np.unique(k, return_inverse=True)
            for ret in (a, b):
                hit = np.zeros(uk.size, dtype=bool)
                np.logical_or.at(hit, inv, ret == fin)
                assert hit.all(), (seed, rnd)
        assert (g.apply(0, x, y) == o.apply(0, x, y)).all(), (seed, rnd)
    rows = o.list_rows().tolist()
    state_equal(g, o, rows[:400], exact_layout=False)
    lens_g = g.m.rowlen_batch(np.array(rows, dtype=np.uint32))
    lens_o = np.array([o.rowlen(r) for r in rows], dtype=np.uint32)
    assert (lens_g == lens_o).all()
    g.close(); o.close()


def test_new_rows_colliding_in_one_directory_slot(G, oracle_mod):
    """row ids crafted so that fmix32(x) & 0xFFFF is ONE value (fmix32 is a public bijection) all want
    the same first-empty slot of the 65536-slot directory; creating 600 of them in one batch must converge (it
    needed one round per id and died at 200) and give the oracle's rows"""
    ids = np.arange(1, 1 << 26, dtype=np.uint32)
    h = ids.copy()
    h ^= h >> 16; h *= np.uint32(0x85ebca6b); h ^= h >> 13; h *= np.uint32(0xc2b2ae35); h ^= h >> 16
    crowd = ids[(h & 0xFFFF) == 0x1234][:600]
    assert crowd.size == 600
    g, o = G(), oracle_mod.Oracle()
    x = np.repeat(crowd, 3); y = np.tile(np.array([1, 2, 17], np.uint32), crowd.size)
    a, b = g.apply(2, x, y, np.ones_like(x)), o.apply(2, x, y, np.ones_like(x))
    assert (a == b).all()
    assert g.stats()["rows"] == 600 and g.stats()["rounds"] <= 4      # create, insert (+ slack), not one round per id
    assert (g.apply(0, x, y) == 1).all()
    for r in crowd[::50].tolist():
        assert g.row_info(r) == o.row_info(r)
    # one op per call on a second crowd: the scalar path creates rows behind occupied slots too
    for r in ids[(h & 0xFFFF) == 0x1235][:40].tolist():
        assert g.incr(r, 5, 2) == o.incr(r, 5, 2)
    assert g.stats()["rows"] == 640
    g.close(); o.close()


def test_y_zero_quirks_sequential(G, oracle_mod):
    """y=0 (Q1-Q3) through the scalar ABI in a fixed order: bit-identical incl. `used`"""
    rng = np.random.default_rng(11)
    g, o = G(), oracle_mod.Oracle()
    for _ in range(3000):
        x = int(rng.integers(0, 6)); y = int(rng.choice([0, 0, 16, 32, 48, 5, 21, int(rng.integers(0, 200))]))
        v = int(rng.integers(0, 3)); op = ("set", "incr", "decr", "get")[int(rng.integers(0, 4))]
        args = (x, y) if op == "get" else (x, y, v)
        assert getattr(g, op)(*args) == getattr(o, op)(*args)
    state_equal(g, o, o.list_rows().tolist(), exact_layout=True)
    g.close(); o.close()


def test_getrow_and_rowlen_batch(G, oracle_mod):
    rng = np.random.default_rng(5)
    x = rng.integers(0, 400, 150000, dtype=np.uint32)
    y = rng.integers(1, 1 << 20, 150000, dtype=np.uint32)
    g, o = G(), oracle_mod.Oracle()
    g.apply(2, x, y, np.ones(x.size, np.uint32)); o.apply(2, x, y, np.ones(x.size, np.uint32))
    xs = np.arange(0, 420, dtype=np.uint32)                # includes absent rows
    lens = g.m.rowlen_batch(xs)
    assert lens.tolist() == [o.rowlen(int(r)) for r in xs]
    off, pairs, cnt = g.m.getrow_batch(xs)
    for r in range(xs.size):
        mine = pairs[off[r]: off[r] + cnt[r]]
        theirs = o.getrow(int(xs[r]))
        assert cnt[r] == theirs.shape[0]
        a = mine[np.lexsort((mine[:, 1], mine[:, 0]))]; b = theirs[np.lexsort((theirs[:, 1], theirs[:, 0]))]
        assert (a == b).all()
        slots = g.row_slots(int(xs[r]))
        if slots is not None:                               # slot order
            ne = slots[(slots[:, 0] != 0) | (slots[:, 1] != 0)]
            assert (mine == ne).all()
    # truncation: capacity 3 pairs per row; scalar ABI: ret_len in BYTES, rounds up (S4)
    off, pairs, cnt = g.m.getrow_batch(xs, caps=np.full(xs.size, 3, np.uint64))
    assert cnt.tolist() == [min(3, o.rowlen(int(r))) for r in xs]
    for rl, want in ((24, 3), (20, 3), (7, 1), (0, 1)):
        assert g.getrow(5, rl).shape[0] == want == o.getrow(5, rl).shape[0]
    assert g.getrow(4000, 64).shape[0] == 0
    g.close(); o.close()


def test_set_batch_duplicates_highest_index_wins(G):
    m = G()
    x = np.array([1, 1, 1, 2, 1, 2], dtype=np.uint32); y = np.array([5, 5, 6, 5, 5, 5], dtype=np.uint32)
    v = np.array([10, 11, 12, 13, 14, 15], dtype=np.uint32)
    assert (m.apply(1, x, y, v) == v).all()
    assert (m.get(1, 5), m.get(1, 6), m.get(2, 5)) == (14, 12, 15)
    n = 100000
    rng = np.random.default_rng(3)
    x = rng.integers(0, 50, n, dtype=np.uint32); y = rng.integers(1, 50, n, dtype=np.uint32)
    v = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    m.apply(1, x, y, v)
    k = x.astype(np.uint64) << 32 | y
    last = {}
    for i in range(n):
        last[int(k[i])] = int(v[i])
    ks = np.array(list(last.keys()), dtype=np.uint64)
    got = m.apply(0, (ks >> 32).astype(np.uint32), (ks & 0xFFFFFFFF).astype(np.uint32))
    assert got.tolist() == list(last.values())
    m.close()
    # at scale: 2^22 Zipf ops into an EMPTY matrix (rows created and grown inside the batch, the bulk path, winners that
    # wait for a round), the all-ones key (no LDS marker may collide with it), then a second batch on the grown table
    gen = Stream("zipf", 99, 200000, 1.1, 1)
    m = G()
    for rep in range(2):
        x, y = gen.fill(rep << 22, 1 << 22)
        x = x.copy(); y = y.copy()
        x[::100003] = 0xFFFFFFFF; y[::100003] = 0xFFFFFFFF
        v = rng.integers(0, 1 << 32, x.size, dtype=np.uint64).astype(np.uint32)
        assert (m.apply(1, x, y, v) == v).all()
        k = x.astype(np.uint64) << 32 | y
        uk, first_rev = np.unique(k[::-1], return_index=True)             # first in the reversed batch = last in the batch
        want = v[::-1][first_rev]
        got = m.apply(0, (uk >> 32).astype(np.uint32), (uk & 0xFFFFFFFF).astype(np.uint32))
        assert (got == want).all(), rep
        assert int(m.m.rowlen_batch(np.unique(x)).astype(np.uint64).sum()) >= uk.size
    m.close(); gen.close()


def test_full_size_batch_properties(G):
    """BASELINE config-2 batch size (2^24 ops of the Zipf(1.1) scrambled stream): size-independent
    properties instead of the oracle -- every get equals the key's multiplicity so far, the
    largest incr return per key equals it too, sum(rowlen) == number of distinct cells"""
    n = 1 << 24
    gen = Stream("zipf", 12345, 1000000, 1.1, 1)
    x, y = gen.fill(0, n)
    m = G()
    ret = m.apply(2, x, y, np.ones(n, np.uint32))
    k = x.astype(np.uint64) << 32 | y
    uk, inv, cnt = np.unique(k, return_inverse=True, return_counts=True)
    got = m.apply(0, x, y)
    assert (got == cnt[inv]).all()
    mx = np.zeros(uk.size, dtype=np.uint32)
    np.maximum.at(mx, inv, ret)
    assert (mx == cnt).all()
    assert int(ret.astype(np.uint64).sum()) == int((cnt.astype(np.uint64) * (cnt + 1) // 2).sum())
    rows = np.unique(x)
    assert int(m.m.rowlen_batch(rows).astype(np.uint64).sum()) == uk.size
    st = m.stats()
    assert st["rows"] == rows.size
    # decr everything back: all cells 0, rowlen unchanged (S3)
    m.apply(3, x, y, np.ones(n, np.uint32))
    assert not m.apply(0, x, y).any()
    assert int(m.m.rowlen_batch(rows).astype(np.uint64).sum()) == uk.size
    m.close()


def test_dense_ids_long_probe_chains(G, oracle_mod):
    """dense Zipf ids: identity hash + linear probing clusters; same answers"""
    gen = Stream("zipf", 99, 200000, 1.1, 0)
    x, y = gen.fill(0, 400000)
    g, o = G(), oracle_mod.Oracle()
    g.apply(2, x, y, np.ones(x.size, np.uint32)); o.apply(2, x, y, np.ones(x.size, np.uint32))
    assert (g.apply(0, x, y) == o.apply(0, x, y)).all()
    rows = o.list_rows()
    assert (g.m.rowlen_batch(rows) == np.array([o.rowlen(int(r)) for r in rows], dtype=np.uint32)).all()
    g.close(); o.close()


def test_dense_ids_cooperative_probe_all_ops(G, oracle_mod):
    """Clustered tables (dense Zipf ids: the reference's identity hash piles ids that wrap onto the dense low run
    behind it, SURVEY.md A.4: displacement up to 12 674) drive every kernel into the wave-cooperative window probe
    (coop_probe): get, incr/decr through the folding kernel (ops set aside -> lane-per-op retry), set with
    duplicate resolution (k_set_locate), prep's absent-key test at the growth threshold.  Same answers, same row
    sizes/rowlens as the oracle; batches and single ops mixed."""
    gen = Stream("zipf", 4242, 300000, 1.1, 0)
    g, o = G(), oracle_mod.Oracle()
    n = 600000
    x, y = gen.fill(0, n)
    x = (x % 40).astype(np.uint32)                      # few rows -> large clustered tables (2^13 .. 2^17 cells)
    rng = np.random.default_rng(9)
    for rnd, op in enumerate((2, 3, 1, 2, 0)):
        a0, a1 = rnd * 100000, rnd * 100000 + 200000
        xs, ys = x[a0:a1], y[a0:a1]
        v = ((xs + ys) % 3 + 1).astype(np.uint32)
        if op == 1:                                     # set: one value per key so that the result is order-free
            v = (ys % 7).astype(np.uint32)
        a, b = g.apply(op, xs, ys, v), o.apply(op, xs, ys, v)
        if op == 0:
            assert (a == b).all()
        assert (g.apply(0, xs, ys) == o.apply(0, xs, ys)).all(), rnd
    assert g.stats()["long_probe_rounds"] > 0           # the folding kernel did hand ops over
    assert g.stats()["clustered_mode"] == 1             # ... so many that the table counts as clustered (wave per op, two-pass move)
    # ids far outside the table wrap onto the dense run: absent keys (get must walk the whole run), then inserts
    far = (np.arange(1, 5001, dtype=np.uint32) * 131072 + rng.integers(1, 200, 5000).astype(np.uint32))
    xr = rng.integers(0, 40, 5000, dtype=np.uint32)
    assert (g.apply(0, xr, far) == o.apply(0, xr, far)).all()
    for k in range(40):                                 # scalar ABI on clustered rows
        assert g.incr(int(xr[k]), int(far[k]), 2) == o.incr(int(xr[k]), int(far[k]), 2)
    a, b = g.apply(2, xr, far, np.ones(5000, np.uint32)), o.apply(2, xr, far, np.ones(5000, np.uint32))
    assert (g.apply(0, xr, far) == o.apply(0, xr, far)).all()
    rows = o.list_rows()
    assert (g.m.rowlen_batch(rows) == np.array([o.rowlen(int(r)) for r in rows], dtype=np.uint32)).all()
    state_equal(g, o, rows.tolist()[:12], exact_layout=False)
    g.close(); o.close(); gen.close()


def test_bulk_path_groups_deferred_ops_by_row(G, oracle_mod, monkeypatch):
    """The bulk path (k_fix_*: deferred ops grouped by row, one wave per row running the reference's insert / resize
    sequence on an LDS table) forced onto small batches (SMATRIX_BULK_MIN=1): new rows, rows that already hold cells,
    duplicate keys inside a batch, decr below zero, y = 0 ops and rows that outgrow the path's 512-cell limit (both
    handed back to the round loop) -- values, per-key return multisets, row sizes and rowlens are the oracle's and the
    tables are valid probe layouts."""
    monkeypatch.setenv("SMATRIX_BULK_MIN", "1")
    monkeypatch.setenv("SMATRIX_BULK_SHARE", "1000000000")      # whatever share of the batch is pending
    rng = np.random.default_rng(41)
    g, o = G(), oracle_mod.Oracle()
    for rnd, (op, n, nx, ny) in enumerate(((2, 40000, 3000, 1 << 20), (2, 60000, 3500, 1 << 20), (3, 30000, 3500, 1 << 20),
                                           (2, 200000, 60000, 50), (2, 90000, 200, 1 << 16), (3, 5000, 100000, 7))):
        x = rng.integers(0, nx, n, dtype=np.uint32)
        y = rng.integers(0 if rnd == 1 else 1, ny, n, dtype=np.uint32)
        v = ((x * 3 + y) % 4 + 1).astype(np.uint32)               # one value per key: order-free return multisets
        if rnd == 1:
            ok = y != 0                                            # y = 0 is order dependent inside a batch: keep those ops apart
            g.apply(op, x[~ok], y[~ok], v[~ok]); o.apply(op, x[~ok], y[~ok], v[~ok])
            x, y, v = x[ok], y[ok], v[ok]
        a, b = g.apply(op, x, y, v), o.apply(op, x, y, v)
        ka, kb = per_key_sorted(x, y, a), per_key_sorted(x, y, b)
        assert (ka[1] == kb[1]).all(), rnd
        assert (g.apply(0, x, y) == o.apply(0, x, y)).all(), rnd
    # set through the same path: new rows, duplicates inside the batch resolve highest-index-wins afterwards
    x = rng.integers(200000, 203000, 50000, dtype=np.uint32); y = rng.integers(1, 400, 50000, dtype=np.uint32)
    v = rng.integers(0, 1 << 32, 50000, dtype=np.uint64).astype(np.uint32)
    assert (g.apply(1, x, y, v) == v).all() and (o.apply(1, x, y, v) == v).all()
    assert (g.apply(0, x, y) == o.apply(0, x, y)).all()
    st = g.stats()
    assert st["bulk_rounds"] >= 5 and st["bulk_ops"] > 150000, st
    rows = o.list_rows()
    assert st["rows"] == rows.size
    assert (g.m.rowlen_batch(rows) == np.array([o.rowlen(int(r)) for r in rows], dtype=np.uint32)).all()
    state_equal(g, o, rows.tolist()[:300] + rows.tolist()[-50:], exact_layout=False)
    # a row fed in ONE op per batch takes the same path and must then be byte-identical (a list of one op has one order)
    for k in range(1, 200):
        for m_ in (g, o):
            m_.apply(2, np.array([4000000], np.uint32), np.array([k * 16 + 5], np.uint32), np.array([k], np.uint32))
    assert g.row_info(4000000) == o.row_info(4000000)
    assert (np.asarray(g.row_slots(4000000)) == np.asarray(o.row_slots(4000000))).all()
    g.close(); o.close()


def test_threads_on_one_handle(G):
    """README.md:113,120 of the reference: all data calls are thread-safe on one handle"""
    m = G()
    T, N = 8, 300

    def work(t):
        for i in range(N):
            m.incr(i % 17, 1 + (i % 29), 1)
            m.get(i % 17, 1 + (i % 29))
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [t.start() for t in th]; [t.join() for t in th]
    total = sum(m.get(a, b) for a in range(17) for b in range(1, 30))
    assert total == T * N
    m.close()


def test_file_roundtrip_and_cross_open(G, oracle_mod, tmp_path, golden):
    """file-backed mode: close is the flush barrier; files are interchangeable with the oracle's
    (and with the real reference's where oracle/_ref is present)"""
    g = golden("fileformat")
    path = str(tmp_path / "g.smx")
    m = G(path)
    for op, *a in g["ops"]:
        getattr(m, op)(*a)
    m.close()
    readers = [oracle_mod.Oracle] + ([oracle_mod.Reference] if oracle_mod.have_reference() else [])
    for R in readers:
        r = R(path)
        for op, a, want in g["after_reopen"]:
            assert getattr(r, op)(*a) == want, (R.__name__, op, a)
        r.close()
    m = G(path)
    for op, a, want in g["after_reopen"]:
        assert getattr(m, op)(*a) == want, ("gpu", op, a)
    m.close()
    # bigger: written by each implementation, read by the GPU library, extended, read back
    rng = np.random.default_rng(7)
    x = rng.integers(0, 3000, 200000, dtype=np.uint32); y = rng.integers(1, 30000, 200000, dtype=np.uint32)
    v = rng.integers(1, 9, 200000, dtype=np.uint32)
    for W in readers:
        p2 = str(tmp_path / ("w_%s.smx" % W.__name__))
        w = W(p2); w.apply(2, x, y, v); want = w.apply(0, x, y)
        lens = [w.rowlen(i) for i in range(3000)]; w.close()
        m = G(p2)
        assert (m.apply(0, x, y) == want).all()
        assert m.m.rowlen_batch(np.arange(3000, dtype=np.uint32)).tolist() == lens
        m.apply(2, x, y + 40000, v); want2 = m.apply(0, x, y + 40000)
        m.close()
        w = W(p2)
        assert (w.apply(0, x, y) == want).all() and (w.apply(0, x, y + 40000) == want2).all()
        w.close()


def test_file_loader_windows_and_foreign_files(G, oracle_mod, tmp_path, monkeypatch):
    """the windowed multi-thread loader/writer on a file that (a) spans many windows (1 MB windows here),
    (b) holds a row block larger than a window, (c) lists a row id TWICE (the later entry wins,
    src/smatrix.c:823), (d) holds a row block whose size is not 16*2^k (legal for the reference's reader,
    which probes `key % size`); same answers as the oracle reading the same bytes, and as the one-thread
    code path"""
    import struct
    rng = np.random.default_rng(31)
    path = str(tmp_path / "foreign.smx")
    o = oracle_mod.Oracle(path)
    x = rng.integers(1, 3000, 150000, dtype=np.uint32); y = rng.integers(1, 1 << 20, 150000, dtype=np.uint32)
    o.apply(2, x, y, np.ones(x.size, np.uint32))
    by = rng.permutation(np.arange(1, 200000, dtype=np.uint32))[:40000]            # row 7777: 2^17 cells = 1 MB
    o.apply(2, np.full(by.size, 7777, np.uint32), by, np.full(by.size, 3, np.uint32))
    o.close()
    with open(path, "r+b") as f:
        raw = f.read()
        assert raw[:8] == b"\x17" * 8
        cmap_at = struct.unpack_from("<Q", raw, 8)[0]
        n_ent, nxt = struct.unpack_from("<QQ", raw, cmap_at)
        assert nxt == 0
        k = 0
        while struct.unpack_from("<Q", raw, cmap_at + 16 + 12 * k + 4)[0]:
            k += 1
        first_x = struct.unpack_from("<I", raw, cmap_at + 16)[0]
        end = len(raw)
        # (c) a second entry for the first row id, pointing at a fresh 16-slot block {(5,50),(21,210)}
        blk_c = bytearray(b"\x23" * 8 + struct.pack("<Q", 16) + bytes(16 * 8))
        struct.pack_into("<II", blk_c, 16 + 8 * 5, 5, 50)
        struct.pack_into("<II", blk_c, 16 + 8 * 6, 21, 210)                         # 21 % 16 = 5 -> probes to slot 6
        # (d) a new row 999999 in a 24-slot block: keys at `key % 24`, one collision chain
        blk_d = bytearray(b"\x23" * 8 + struct.pack("<Q", 24) + bytes(24 * 8))
        for key, val in ((3, 30), (27, 270), (51, 510), (23, 230)):
            q = key % 24
            while struct.unpack_from("<I", blk_d, 16 + 8 * q)[0]:
                q = (q + 1) % 24
            struct.pack_into("<II", blk_d, 16 + 8 * q, key, val)
        f.seek(end); f.write(blk_c); f.write(blk_d)
        f.seek(cmap_at + 16 + 12 * k)
        f.write(struct.pack("<IQ", first_x, end) + struct.pack("<IQ", 999999, end + len(blk_c)))
    ref = oracle_mod.Oracle(path)
    qx = np.concatenate([x, np.full(by.size, 7777, np.uint32), np.full(4, first_x, np.uint32), np.full(5, 999999, np.uint32)])
    qy = np.concatenate([y, by, np.array([5, 21, 6, 37], np.uint32), np.array([3, 27, 51, 23, 4], np.uint32)])
    want = ref.apply(0, qx, qy)
    rows = np.unique(qx)
    lens = [ref.rowlen(int(r)) for r in rows]
    ref.close()
    assert want[-9:].tolist() == [50, 210, 0, 0, 30, 270, 510, 230, 0]               # the later entry replaced the row
    got = {}
    for name, env in (("windows", {"SMATRIX_IO_WINDOW_MB": "1", "SMATRIX_IO_THREADS": "4"}), ("default", {}),
                      ("serial", {"SMATRIX_IO_THREADS": "1"})):
        for k_, v_ in (("SMATRIX_IO_WINDOW_MB", None), ("SMATRIX_IO_THREADS", None)):
            monkeypatch.delenv(k_, raising=False)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        import shutil
        p2 = str(tmp_path / ("copy_%s.smx" % name))
        shutil.copy(path, p2)
        m = G(p2)
        assert (m.apply(0, qx, qy) == want).all(), name
        assert m.m.rowlen_batch(rows).tolist() == lens, name
        m.apply(2, qx[:1000], qy[:1000] + (1 << 21), np.ones(1000, np.uint32))      # grow a little, write it back
        m.close()
        back = oracle_mod.Oracle(p2)                                                # our rewrite, read by the oracle
        assert (back.apply(0, qx, qy) == want).all(), name
        assert (back.apply(0, qx[:1000], qy[:1000] + (1 << 21)) >= 1).all(), name
        back.close()


def test_apply_packed_records(G, oracle_mod):
    """smatrix_apply_packed_dev: the op kernels read {x,y[,v]} records in place (what the sharded exchange
    delivers); same results as the three-array form, growth rounds included"""
    import torch
    rng = np.random.default_rng(5)
    n = 300000
    x = rng.integers(0, 4000, n, dtype=np.uint32); y = rng.integers(1, 9000, n, dtype=np.uint32)   # y >= 1: order independent
    v = rng.integers(0, 5, n, dtype=np.uint32)
    g, o = G(), oracle_mod.Oracle()
    dev = torch.device("cuda", 0)
    rec3 = torch.from_numpy(np.stack([x, y, v], 1).view(np.int32)).to(dev).contiguous()
    rec2 = rec3[:, :2].contiguous()
    out = torch.empty(n, dtype=torch.int32, device=dev)
    for op in (2, 3, 1, 2):
        g.m.apply_packed_dev(op, n, rec3.data_ptr(), 3, out.data_ptr(), None)
        o.apply(op, x, y, v)
        g.m.apply_packed_dev(0, n, rec2.data_ptr(), 2, out.data_ptr(), None)
        assert (out.cpu().numpy().view(np.uint32) == o.apply(0, x, y)).all(), op
    rows = np.unique(x)
    assert g.m.rowlen_batch(rows).tolist() == [o.rowlen(int(r)) for r in rows]
    with pytest.raises(ValueError):
        g.m.apply_packed_dev(2, n, rec2.data_ptr(), 2, out.data_ptr(), None)   # a write needs {x,y,v}
    g.close(); o.close()


def test_sharded_path_one_rank_nccl():
    """the HIP partitioner + RCCL all_to_all + local shard with world_size 1 (all a 1-GPU box allows):
    bench.py --force-sharded must produce the same sane line as the direct path"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-sharded", "--batch-lg", "18",
                        "--steps", "3", "--warmup", "1", "--no-cpu"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["sanity_all_gets_positive"] and res["value"] > 0 and res["table"]["rows"] > 0


def test_partition_kernels(G):
    """smatrix_partition_dev / smatrix_gather_dev: a permutation that groups ops by owner shard"""
    import ctypes as C
    import torch
    from libsmatrix_amd import _lib
    from libsmatrix_amd.sharded import HipPartitioner
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2)
    for n, world in ((1, 2), (1000, 3), (300001, 8), (0, 4)):
        x = torch.from_numpy(rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.int32)).to(dev)
        y = torch.arange(n, dtype=torch.int32, device=dev)
        v = y * 3
        part = HipPartitioner(dev)
        counts, perm, xo, yo, vo = part.partition(x, y, v, world)
        torch.cuda.synchronize()
        assert sum(counts) == n
        own = np.array([lib.smatrix_shard_of(int(a), world) for a in x.cpu().numpy().view(np.uint32)[:2000]])
        p = perm.cpu().numpy()
        assert sorted(p.tolist()) == list(range(n))                       # a permutation
        starts = np.concatenate([[0], np.cumsum(counts)])
        for i in range(min(n, 2000)):                                     # op i sits in its owner's range
            assert starts[own[i]] <= p[i] < starts[own[i] + 1]
        assert (yo.cpu().numpy()[p] == y.cpu().numpy()).all() and (vo.cpu().numpy()[p] == v.cpu().numpy()).all()
        assert (xo.cpu().numpy()[p] == x.cpu().numpy()).all()
        out = torch.empty_like(y)
        part.gather(yo, perm, out)
        torch.cuda.synchronize()
        assert (out == y).all()


def test_big_rows_subcounters(G, oracle_mod, tmp_path):
    """rows of >= 2^15 cells count inserts in per-row sub-counters with quotas:
    rowlen/size must stay exactly the reference's across growth, reload and quota re-partitioning"""
    rng = np.random.default_rng(21)
    g, o = G(), oracle_mod.Oracle()
    total = 0
    for rnd in range(10):
        n = 20000 + 7000 * rnd
        x = rng.integers(0, 3, n, dtype=np.uint32)                 # three giant rows
        y = rng.integers(1, 1 << 31, n, dtype=np.uint32)
        v = np.ones(n, np.uint32)
        g.apply(2, x, y, v); o.apply(2, x, y, v)
        total += n
        for r in range(3):
            assert g.row_info(r) == o.row_info(r), (rnd, r)
        assert (g.apply(0, x, y) == o.apply(0, x, y)).all()
    assert g.stats()["rows_rebalanced"] >= 0
    # getrow of giant rows goes through the workgroup-per-row kernel: slot order, truncation
    for r in range(3):
        slots = np.asarray(g.row_slots(r))
        ne = slots[(slots[:, 0] != 0) | (slots[:, 1] != 0)]
        full = g.getrow(r)
        assert full.shape[0] == ne.shape[0] == o.getrow(r).shape[0] and (full == ne).all()
        assert (g.getrow(r, 8 * 5000) == ne[:5000]).all() and g.getrow(r, 20).shape[0] == 3
        # these rows are cut into 32768-cell segments, one workgroup each: a buffer that ends inside the 4th segment
        assert slots.shape[0] >= 4 * 32768
        head = slots[:3 * 32768]
        cut = int(((head[:, 0] != 0) | (head[:, 1] != 0)).sum()) + 17
        assert (g.getrow(r, 8 * cut) == ne[:cut]).all()
    off, pairs, cnt = g.m.getrow_batch(np.array([0, 1, 2, 77], dtype=np.uint32))
    assert cnt.tolist() == [o.rowlen(0), o.rowlen(1), o.rowlen(2), 0]
    pairs = np.asarray(pairs).reshape(-1, 2)
    for i in range(3):
        slots = np.asarray(g.row_slots(i))
        ne = slots[(slots[:, 0] != 0) | (slots[:, 1] != 0)]
        assert (pairs[int(off[i]):int(off[i]) + int(cnt[i])] == ne).all()
    # land exactly on thresholds: one op at a time around the 2^15 -> 2^16 doubling of a fresh row
    keys = np.arange(1, (1 << 15) + 40, dtype=np.uint32) * 7
    g.apply(2, np.full(keys.size - 80, 9, np.uint32), keys[:-80], np.ones(keys.size - 80, np.uint32))
    o.apply(2, np.full(keys.size - 80, 9, np.uint32), keys[:-80], np.ones(keys.size - 80, np.uint32))
    for k in keys[-80:].tolist():
        assert g.incr(9, k, 1) == o.incr(9, k, 1)
        assert g.row_info(9) == o.row_info(9)
    # reload a file holding big rows, keep inserting
    path = str(tmp_path / "big.smx")
    w = oracle_mod.Oracle(path)
    yb = rng.integers(1, 1 << 31, 50000, dtype=np.uint32)
    w.apply(2, np.zeros(50000, np.uint32), yb, np.ones(50000, np.uint32)); w.close()
    m, o2 = G(path), oracle_mod.Oracle(path)
    assert m.row_info(0) == o2.row_info(0)
    y2 = rng.integers(1, 1 << 31, 60000, dtype=np.uint32)
    m.apply(2, np.zeros(60000, np.uint32), y2, np.ones(60000, np.uint32))
    o2.apply(2, np.zeros(60000, np.uint32), y2, np.ones(60000, np.uint32))
    assert m.row_info(0) == o2.row_info(0) and (m.apply(0, np.zeros(60000, np.uint32), y2) == o2.apply(0, np.zeros(60000, np.uint32), y2)).all()
    m.close(); o2.close(); g.close(); o.close()


def test_growth_every_size_one_op_at_a_time(G, oracle_mod):
    """One op per call (= the scalar ABI's semantics), so every table must be BYTE-identical to the
    reference's at every size: two rows grow from 16 to 32768 cells through the in-LDS rehash by a wave
    (<= 256 old cells), by a 256-lane and a 1024-lane workgroup (<= 8192) and the chunked passes beyond;
    the second row is salted with y = 0 writes and keys congruent to 0 so that the uncounted (0,v) cell
    appears, vanishes and leaves DUPLICATE keys behind (quirks Q1/Q3), which the resize must merge the
    reference's way (the sequential redo in k_grow_commit)."""
    rng = np.random.default_rng(77)
    g, o = G(), oracle_mod.Oracle()
    one = lambda a: np.array([a], dtype=np.uint32)

    def both(op, x, y, v):
        a = g.apply(op, one(x), one(y), one(v))[0]
        b = o.apply(op, one(x), one(y), one(v))[0]
        assert a == b, (op, x, y, v, a, b)

    def same_tables(tag):
        for x in (5, 6):
            assert g.row_info(x) == o.row_info(x), (tag, x, g.row_info(x), o.row_info(x))
            assert (np.asarray(g.row_slots(x)) == np.asarray(o.row_slots(x))).all(), (tag, x)

    for step in range(9200):
        both(2, 5, int(rng.integers(1, 1 << 30)), 1)                      # row 5: plain inserts
        r = rng.random()
        size = (o.row_info(6) or (16, 0))[0]
        if r < 0.55:
            both(2, 6, int(rng.integers(1, 1 << 30)), 3)
        elif r < 0.70:
            both(2, 6, int(rng.integers(1, 64)) * size, 1)                 # home slot 0: piles up behind (0,v)
        elif r < 0.80:
            both(2, 6, 0, 7)                                               # y = 0: the uncounted (0,v) cell ...
        elif r < 0.90:
            both(3, 6, 0, 7)                                               # ... and back to (0,0) = empty
        else:
            both(1, 6, int(rng.integers(1, 64)) * size, int(rng.integers(0, 3)))
        if step % 700 == 0:
            same_tables(step)
    same_tables("end")
    assert o.row_info(5)[0] >= 16384 and o.row_info(6)[0] >= 8192
    g.close(); o.close()


def test_config1_stock_benchmark_pattern(G, oracle_mod):
    """BASELINE config 1: the reference benchmark's fixed 23x22 id block (src/smatrix_benchmark.c:29-65),
    T threads' ops as one batch each for incr then get -- heavy duplication inside a batch"""
    sys_path = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(
        __import__("os").path.abspath(__file__))), "tools")
    __import__("sys").path.insert(0, sys_path)
    from smatrix_benchmark import pattern
    g, o = G(), oracle_mod.Oracle()
    for T in (1, 2, 8, 32):
        xs, ys = zip(*(pattern(t, 1024 // T) for t in range(T)))
        x, y = np.concatenate(xs), np.concatenate(ys)
        assert x.size == 1036288 // 1  # 1024 * 23*22*2 ops per cell of the reference's table
        ri, ro = g.apply(2, x, y, np.ones_like(x)), o.apply(2, x, y, np.ones_like(x))
        a, b = per_key_sorted(x, y, ri), per_key_sorted(x, y, ro)
        assert (a[1] == b[1]).all()
        assert (g.apply(0, x, y) == o.apply(0, x, y)).all()
    rows = o.list_rows().tolist()
    state_equal(g, o, rows, exact_layout=False)
    g.close(); o.close()


def test_error_convention(G, tmp_path):
    """src/smatrix.c:92-96: open failure -> NULL (ValueError in the binding);
    :582-590: a bad header is fatal -- message on stdout, then abort()"""
    import subprocess, sys, os
    from libsmatrix_amd import SparseMatrix
    with pytest.raises(ValueError):
        SparseMatrix(str(tmp_path / "no_such_dir" / "x.smx"))
    bad = tmp_path / "bad.smx"
    bad.write_bytes(b"\x00" * 600)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c",
                        "import sys; sys.path.insert(0, %r); from libsmatrix_amd import SparseMatrix; SparseMatrix(%r)"
                        % (root, str(bad))], capture_output=True, text=True)
    assert p.returncode == -6 and "libsmatrix error: invalid file header" in p.stdout      # SIGABRT
    # empty batches and a fresh file are fine
    m = G(str(tmp_path / "fresh.smx"))
    assert m.apply(2, np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint32)).size == 0
    assert m.rowlen(1) == 0 and m.getrow(1, 64).shape[0] == 0
    m.close()
    assert os.path.getsize(str(tmp_path / "fresh.smx")) == 512 + 16 + 4194304 * 12
    m = G(str(tmp_path / "fresh.smx")); assert m.get(1, 1) == 0; m.close()


def test_cf_recommender_read_path(G, oracle_mod):
    """examples/cf_recommender.c: import preference sets (incr(id,0,1) + co-occurrence incrs), then the
    fused neighbours+cosine kernel against the oracle's restatement.  double arithmetic: sqrt and
    division are correctly rounded on both sides -> tolerance 1e-15 relative (expected: bit-equal)"""
    rng = np.random.default_rng(8)
    xs, ys = [], []
    for _ in range(4000):                                        # preference sets of 2..8 items out of 600
        ids = rng.choice(600, size=int(rng.integers(2, 9)), replace=False) + 1
        for a in ids:
            xs.append(a); ys.append(0)                           # :38  total per item in column 0 (quirk Q1)
            for b in ids:
                if a != b:
                    xs.append(a); ys.append(b)                   # :40-44
    x, y = np.array(xs, np.uint32), np.array(ys, np.uint32)
    g, o = G(), oracle_mod.Oracle()
    g.apply(2, x, y, np.ones_like(x)); o.apply(2, x, y, np.ones_like(x))
    items = np.arange(0, 640, dtype=np.uint32)                   # includes absent items
    off, ids, sc, cnt = g.m.cf_neighbors_batch(items)
    worst = 0.0
    for i, it in enumerate(items.tolist()):
        wi, ws = oracle_mod.cf_neighbors(o, it, 100000)
        assert cnt[i] == wi.size
        mine_i, mine_s = ids[off[i]: off[i] + cnt[i]], sc[off[i]: off[i] + cnt[i]]
        a, b = np.argsort(mine_i, kind="stable"), np.argsort(wi, kind="stable")   # batch layout may differ
        assert (mine_i[a] == wi[b]).all()
        if wi.size:
            worst = max(worst, float(np.max(np.abs(mine_s[a] - ws[b]) / np.maximum(ws[b], 1e-300))))
    assert worst <= 1e-15, worst
    assert sc.max() <= 1.0 and sc.min() >= 0.0 and (sc > 0).any()
    # truncation like the example's 8192-byte buffer (1024 pairs)
    off, ids, sc, cnt = g.m.cf_neighbors_batch(items, caps=np.full(items.size, 5, np.uint64))
    assert cnt.max() == 5
    g.close(); o.close()


def test_cf_topk(G, oracle_mod):
    """the k best neighbours per item (smatrix_cf_topk_batch) against the oracle's full neighbour list ordered by
    (score descending, table position ascending).  First a matrix built ONE OP PER CALL, whose tables are byte-identical
    to the oracle's, so that ties must come out in the same order; then a batch-built one (layouts may differ: scores
    exact, every returned id carries the oracle's score for it, nothing better was left out)."""
    rng = np.random.default_rng(12)

    def expected(o, it, k):
        wi, ws = oracle_mod.cf_neighbors(o, it, 1 << 20)
        order = np.lexsort((np.arange(wi.size), -ws))[:k]
        return wi, ws, order

    # (a) identical layouts: scalar calls
    g, o = G(), oracle_mod.Oracle()
    for _ in range(120):
        ids = rng.choice(40, size=int(rng.integers(2, 7)), replace=False) + 1
        for a in ids.tolist():
            g.incr(a, 0, 1); o.incr(a, 0, 1)
            for b in ids.tolist():
                if a != b:
                    g.incr(a, b, 1); o.incr(a, b, 1)
    items = np.arange(0, 45, dtype=np.uint32)
    for k in (1, 3, 10, 64):
        ids_k, sc_k, cnt = g.m.cf_topk_batch(items, k)
        for i, it in enumerate(items.tolist()):
            wi, ws, order = expected(o, it, k)
            assert cnt[i] == order.size, (k, it)
            assert ids_k[i, :cnt[i]].tolist() == wi[order].tolist(), (k, it)
            assert sc_k[i, :cnt[i]].tolist() == ws[order].tolist(), (k, it)
    with pytest.raises(ValueError):
        g.m.cf_topk_batch(items, 65)
    g.close(); o.close()
    # (b) batch-built, rows of up to ~600 entries (several 64-cell steps, merges and skipped steps)
    xs, ys = [], []
    for _ in range(6000):
        ids = rng.choice(900, size=int(rng.integers(2, 12)), replace=False) + 1
        for a in ids:
            xs.append(a); ys.append(0)
            for b in ids:
                if a != b:
                    xs.append(a); ys.append(b)
    x, y = np.array(xs, np.uint32), np.array(ys, np.uint32)
    g, o = G(), oracle_mod.Oracle()
    g.apply(2, x, y, np.ones_like(x)); o.apply(2, x, y, np.ones_like(x))
    items = np.arange(0, 920, dtype=np.uint32)
    for k in (5, 64):
        ids_k, sc_k, cnt = g.m.cf_topk_batch(items, k)
        for i, it in enumerate(items.tolist()):
            wi, ws, order = expected(o, it, k)
            assert cnt[i] == order.size
            assert sc_k[i, :cnt[i]].tolist() == ws[order].tolist(), (k, it)
            score_of = dict(zip(wi.tolist(), ws.tolist()))
            assert all(score_of[a] == b for a, b in zip(ids_k[i, :cnt[i]].tolist(), sc_k[i, :cnt[i]].tolist())), (k, it)
            assert len(set(ids_k[i, :cnt[i]].tolist())) == cnt[i]
    g.close(); o.close()


def test_column_zero_totals_under_contention(G):
    """every item's total lives in column 0 in the CF example (examples/cf_recommender.c:38): millions of incr(x, 0, 1) on a few
    hot items, mixed with inserts into the same rows (the (0,v) cell is found by probing for 'key field 0', which an empty
    slot another key is claiming also satisfies).  No update may be lost: the totals are exact."""
    rng = np.random.default_rng(77)
    n = 1 << 22
    x = rng.choice(np.array([5, 6, 7, 1000003], np.uint32), n, p=[0.7, 0.2, 0.09, 0.01]).astype(np.uint32)
    y = np.where(rng.random(n) < 0.8, 0, rng.integers(1, 50000, n)).astype(np.uint32)
    k = (x.astype(np.uint64) << 32 | y)[y != 0]
    uk, cnt = np.unique(k, return_counts=True)
    for with_results in (True, False):                   # False: out == NULL -> the 64-bit-add form of the column-0 update
        g = G()
        for a in range(0, n, 1 << 20):                   # four batches: the rows grow in between
            xs_, ys_ = x[a:a + (1 << 20)], y[a:a + (1 << 20)]
            if with_results:
                g.apply(2, xs_, ys_, np.ones(1 << 20, np.uint32))
            else:
                assert g.m._lib.smatrix_apply_batch(g.m._h, 2, xs_.size, xs_.ctypes.data_as(_u32p), ys_.ctypes.data_as(_u32p),
                                                    np.ones(1 << 20, np.uint32).ctypes.data_as(_u32p), None) == 0
        for item in (5, 6, 7, 1000003):
            assert g.get(item, 0) == int(((x == item) & (y == 0)).sum()), (item, with_results)
        got = g.apply(0, (uk >> 32).astype(np.uint32), (uk & 0xFFFFFFFF).astype(np.uint32))
        assert (got == cnt).all()
        g.close()


def test_cf_recommender_write_path_on_device(G, oracle_mod):
    """examples/cf_recommender.c:36-47 import_preference_set, expanded on the GPU (smatrix_cf_import_sessions: L*L incr ops
    per session of L ids) against the oracle's restatement called session by session: every row holds the same cells
    (the (0,total) cell included), every get agrees, and the read path scores agree.  Sessions: empty, one id, an id
    twice (the example tests POSITIONS i != n, so the id then counts itself), up to 40 ids.  rowlen itself is not
    compared: with y = 0 in a row it depends on WHEN the row last grew (quirk Q2), and a batch is one legal order."""
    rng = np.random.default_rng(31)
    sessions = [[], [7], [9, 9], [3, 5, 3]]
    for _ in range(3000):
        L = int(rng.integers(0, 13))
        sessions.append((rng.integers(1, 400, L)).tolist())
    sessions.append((rng.choice(5000, 40, replace=False) + 1000).tolist())
    g, o = G(), oracle_mod.Oracle()
    g.m.cf_import_sessions(sessions)
    for s_ in sessions:
        oracle_mod.cf_import_preference_set(o, s_)
    rows = o.list_rows()
    assert g.stats()["rows"] == rows.size
    for r in rows.tolist():
        a, b = np.asarray(g.row_slots(r)), np.asarray(o.row_slots(r))
        ka = a[(a[:, 0] != 0) | (a[:, 1] != 0)]; kb = b[(b[:, 0] != 0) | (b[:, 1] != 0)]
        ka = ka[np.lexsort((ka[:, 1], ka[:, 0]))]; kb = kb[np.lexsort((kb[:, 1], kb[:, 0]))]
        assert ka.shape == kb.shape and (ka == kb).all(), r
    for a_, b_ in ((9, 9), (9, 0), (3, 3), (3, 5), (5, 3), (7, 0), (1039, 0)):
        assert g.get(a_, b_) == o.get(a_, b_), (a_, b_)
    assert o.get(9, 9) >= 2 and o.get(3, 3) >= 2                      # the repeated id counted itself
    items = rows[:300]
    off, ids, sc, cnt = g.m.cf_neighbors_batch(items)
    for i, it in enumerate(items.tolist()):
        wi, ws = oracle_mod.cf_neighbors(o, it, 100000)
        mine = dict(zip(ids[off[i]: off[i] + cnt[i]].tolist(), sc[off[i]: off[i] + cnt[i]].tolist()))
        assert mine == dict(zip(wi.tolist(), ws.tolist())), it
    g.close(); o.close()


def test_sharded_pipeline_matches_direct():
    """split-phase ShardedMatrix with its own communication stream (the form bench.py pipelines at N>1)
    against the direct path, one rank over RCCL: identical get results and per-key incr returns"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "sharded_gpu_check.py")],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "SHARDED_PIPELINE_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


@pytest.mark.parametrize("bulk_min", [None, "1"])
def test_hypothesis_small_sequences_vs_oracle(G, oracle_mod, monkeypatch, bulk_min):
    """property test: arbitrary short programs of batches over a tiny, collision-heavy id space (y = 0,
    value 0, wrap-around, duplicate keys, repeated growth) -- after every batch all cells, rowlens and row
    sizes equal the oracle's; batches of one op kind are applied to both in the same index order.
    Second run: every incr/decr batch is forced through the bulk path (SMATRIX_BULK_MIN=1), whose per-row kernel
    then meets the quirk rows: (0,v) cells, keys held twice after a chain cut, resizes that merge them."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    if bulk_min:
        monkeypatch.setenv("SMATRIX_BULK_MIN", bulk_min)
        monkeypatch.setenv("SMATRIX_BULK_SHARE", "1000000000")

    ids = st.sampled_from([0, 1, 2, 3, 16, 17, 32, 48, 64, 5, 21, 0xFFFFFFFF, 0x80000000])
    vals = st.sampled_from([0, 1, 2, 7, 0xFFFFFFFF])
    op = st.tuples(ids, ids, vals)
    batch = st.tuples(st.sampled_from([1, 2, 3]), st.lists(op, min_size=1, max_size=40))
    program = st.lists(batch, min_size=1, max_size=8)

    @settings(max_examples=int(__import__("os").environ.get("SMX_HYP_EXAMPLES", "60")), deadline=None,
              suppress_health_check=list(HealthCheck))
    @given(program)
    def run(prog):
        g, o = G(), oracle_mod.Oracle()
        try:
            for kind, ops in prog:
                x = np.array([a for a, _, _ in ops], np.uint32)
                y = np.array([b for _, b, _ in ops], np.uint32)
                v = np.array([c for _, _, c in ops], np.uint32)
                if kind == 1 or (y == 0).any():
                    # set, and anything touching the y = 0 quirk cell, is order dependent inside a batch:
                    # apply one op per call on both sides (the reference's own call pattern)
                    for a, b, c in zip(x.tolist(), y.tolist(), v.tolist()):
                        name = ("", "set", "incr", "decr")[kind]
                        assert getattr(g, name)(a, b, c) == getattr(o, name)(a, b, c)
                else:
                    g.apply(kind, x, y, v); o.apply(kind, x, y, v)
                rows = o.list_rows().tolist()
                for r in rows:
                    assert g.row_info(r) == o.row_info(r), (r, g.row_info(r), o.row_info(r))
                qx = np.repeat(np.array(rows, np.uint32), 13) if rows else np.zeros(0, np.uint32)
                qy = np.tile(np.array([0, 1, 2, 3, 16, 17, 32, 48, 64, 5, 21, 0xFFFFFFFF, 0x80000000], np.uint32), len(rows))
                if qx.size:
                    assert (g.apply(0, qx, qy) == o.apply(0, qx, qy)).all()
        finally:
            g.close(); o.close()

    run()


def test_sharded_two_ranks_one_gpu():
    """world_size 2 on ONE GPU (gloo, payload staged through the host): real HIP partitioner + shards +
    pipelined routing against an un-sharded HIP matrix -- the closest a 1-GPU box gets to the N>1 path"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29599", os.path.join(root, "tests", "sharded_gpu_2rank.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert p.returncode == 0 and "SHARDED_2RANK_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


@pytest.mark.parametrize("world,router", [(2, "c"), (8, "c"), (8, "python")])
def test_bench_launch_contract_n_ranks(world, router):
    """`python bench.py --gpus N ...` AS GIVEN (no torch.distributed.run in front): bench.py spawns its own N ranks as
    child processes before it touches a GPU, rank 0 prints the ONE JSON line.  On the 1-GPU rig all ranks sit on cuda:0
    (gloo for the process group; the payload staged through the host / the C router's shared-memory transport).  N > 1 is
    config 4 (8M x 8M ids, one stream per rank).  At 8 ranks -- the node size of the scaling run -- the planned placement
    must leave every shard within 10 % of the mean load; equal hash ranges put 1.9x the mean on the owner of the hottest
    row.  router = "c" (the default since round 4): the C library's own router (include/smatrix_shard.h), after a preflight
    batch under a watchdog; "python": --py-router, the torch.distributed one."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--batch-lg", "18" if world == 2 else "20", "--backend", "gloo", "--single-device"] + (["--py-router"] if router == "python" else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == world and res["steps"] == 3 and res["scaling"] == "weak" and res["sanity_all_gets_positive"]
    assert res["value"] > 0 and "roofline" in res
    assert res["config"]["workload"].startswith("config-4") and res["config"]["ids_per_axis"] == 8000000
    # the C library's router is the default N > 1 path (round 4); --py-router selects the torch.distributed one
    assert res["config"]["router"] == ("c-library/shm" if router == "c" else "torch.distributed/gloo"), res["config"]
    assert len(lines[0]) < 4096
    pl = res["config"]["placement"]
    assert pl["rows_placed_by_load"] > 0 and len(pl["ops_applied_over_mean"]) == world
    if world == 8:
        assert max(pl["ops_applied_over_mean"]) < 1.10, pl


def test_bench_launch_contract_torchrun():
    """the driver's other launch line: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N (RANK /
    LOCAL_RANK / WORLD_SIZE from the environment): no second fan-out, one JSON line"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29613", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch-lg", "18",
           "--backend", "gloo", "--single-device"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
