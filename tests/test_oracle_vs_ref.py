"""CPU, build container only: live differential of the oracle against the REAL reference
(oracle/_ref, compiled from /root/reference by oracle/Makefile).  Skipped where the
compiled reference is absent."""
import os

import numpy as np
import pytest

from tests import replay


@pytest.fixture(scope="module")
def pair(oracle_mod):
    if not oracle_mod.have_reference():
        pytest.skip("oracle/_ref/libsmatrix_ref.so not built (needs /root/reference)")
    return oracle_mod


def random_ops(rng, n, nx, ny, zero_frac=0.05):
    x = rng.integers(0, nx, n, dtype=np.uint32)
    y = rng.integers(0, ny, n, dtype=np.uint32)
    y[rng.random(n) < zero_frac] = 0                      # quirks Q1-Q3
    v = rng.integers(0, 4, n, dtype=np.uint32)            # value 0 included (S3)
    v[rng.random(n) < 0.01] = 0xFFFFFFFF                  # wrap-around
    op = rng.integers(0, 4, n)
    return op, x, y, v


@pytest.mark.parametrize("seed,nx,ny", [(1, 50, 40), (2, 3, 2000), (3, 5000, 5000), (4, 1, 1 << 30)])
def test_random_mixed_ops(pair, seed, nx, ny):
    rng = np.random.default_rng(seed)
    op, x, y, v = random_ops(rng, 30000, nx, ny)
    o, r = pair.Oracle(), pair.Reference()
    for k in range(4):                                    # runs of one op kind, like the batch API
        sel = op == k
        a = o.apply(k, x[sel], y[sel], v[sel])
        b = r.apply(k, x[sel], y[sel], v[sel])
        assert (a == b).all()
    # interleaved too
    for i in range(2000):
        name = ("get", "set", "incr", "decr")[op[i]]
        args = (int(x[i]), int(y[i])) + (() if op[i] == 0 else (int(v[i]),))
        assert getattr(o, name)(*args) == getattr(r, name)(*args)
    rows = r.list_rows().tolist()
    assert sorted(o.list_rows().tolist()) == sorted(rows)
    assert o.mem() == r.mem() and o.dir_size() == r.dir_size()
    assert replay.layout_digest(o, rows) == replay.layout_digest(r, rows)
    for xx in rows[:200]:
        assert o.rowlen(xx) == r.rowlen(xx)
        for rl in (0, 7, 8, 20, 24, 4096):
            assert o.getrow(xx, rl).tolist() == r.getrow(xx, rl).tolist()
    o.close(); r.close()


def test_file_cross_open(pair, tmp_path):
    """oracle-written file read by the reference and vice versa"""
    rng = np.random.default_rng(7)
    x = rng.integers(0, 300, 20000, dtype=np.uint32)
    y = rng.integers(1, 3000, 20000, dtype=np.uint32)
    v = rng.integers(1, 9, 20000, dtype=np.uint32)
    for writer, reader in ((pair.Oracle, pair.Reference), (pair.Reference, pair.Oracle)):
        path = str(tmp_path / ("x_%s.smx" % writer.__name__))
        w = writer(path)
        w.apply(pair.OP_INCR, x, y, v)
        want = w.apply(pair.OP_GET, x, y)
        lens = [w.rowlen(i) for i in range(300)]
        w.close()
        r = reader(path)
        assert (r.apply(pair.OP_GET, x, y) == want).all()
        assert [r.rowlen(i) for i in range(300)] == lens
        # keep writing through the reader, close, reopen with the writer's implementation
        r.apply(pair.OP_INCR, x[:5000], y[:5000] + 5000, v[:5000])
        want2 = r.apply(pair.OP_GET, x[:5000], y[:5000] + 5000)
        r.close()
        w = writer(path)
        assert (w.apply(pair.OP_GET, x[:5000], y[:5000] + 5000) == want2).all()
        assert (w.apply(pair.OP_GET, x, y) == want).all()
        w.close()
        os.remove(path)


def test_q4_reload_drops_zero_values(pair, tmp_path):
    """quirk Q4 (src/smatrix.c:533-540): both drop value-0 keys on load, chain break included"""
    for impl in (pair.Oracle, pair.Reference):
        path = str(tmp_path / ("q4_%s.smx" % impl.__name__))
        m = impl(path)
        m.set(9, 1, 5); m.set(9, 17, 0); m.set(9, 33, 6)
        assert m.get(9, 33) == 6 and m.rowlen(9) == 3
        m.close()
        m = impl(path)
        assert (m.get(9, 1), m.get(9, 17), m.get(9, 33), m.rowlen(9)) == (5, 0, 0, 2)
        m.close()
