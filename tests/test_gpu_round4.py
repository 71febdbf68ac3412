"""Round 4 GPU tests (through the C ABI): what VERDICT r3 / ADVICE r3 asked for."""
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def G():
    from tests.gpu_adapter import GpuMatrix
    import libsmatrix_amd
    assert libsmatrix_amd.device_available(), "no HIP device: the product has no CPU fallback"
    return GpuMatrix


def _mirror_mix(x, y):
    """CellCache::mix of csrc/smx_runtime.hip, vectorised (uint64 wrap-around arithmetic)"""
    with np.errstate(over="ignore"):
        z = ((x.astype(np.uint64) << np.uint64(32)) | y.astype(np.uint64)) * np.uint64(0x9e3779b97f4a7c15)
        z ^= z >> np.uint64(29)
        z *= np.uint64(0xbf58476d1ce4e5b9)
        z ^= z >> np.uint64(32)
    return z


def test_scalar_mirror_get_never_returns_another_cells_value(G, monkeypatch):
    """the lock-free get of the scalar ABI's host mirror read an entry's value word and its key with
    two loads; a reader that stalled in between while the shard was wiped and ANOTHER key with the same home slot was put
    there paired the new key with the old entry's value, and smatrix_get returned another cell's value.  Here eight cells
    whose mirror entries share ONE shard and ONE home slot (found with the mirror's own hash) are read by eight threads
    with a four-entry shard: every miss re-mirrors a cell at the contended slots, the shard is recycled all the time, and
    every get must still return ITS cell's value (each cell holds a value that names it)."""
    monkeypatch.setenv("SMATRIX_SCALAR_CACHE_CAP", "4")
    import threading
    x0 = np.uint32(77)
    ys = np.arange(1, 1 << 23, dtype=np.uint32)
    h = _mirror_mix(np.full(ys.size, x0, np.uint32), ys)
    home = (h & np.uint64(15)) << np.uint64(18) | ((h >> np.uint64(4)) & np.uint64((1 << 18) - 1))     # {shard, home slot}
    order = np.argsort(home, kind="stable")
    hs = home[order]
    # the longest run of equal {shard, slot}
    starts = np.flatnonzero(np.r_[True, hs[1:] != hs[:-1]])
    lens = np.diff(np.r_[starts, hs.size])
    best = int(np.argmax(lens))
    assert lens[best] >= 4, "no colliding keys found"
    cy = ys[order[starts[best]:starts[best] + min(int(lens[best]), 8)]]
    # a few more keys of the same shard on neighbouring slots (they extend the contended probe sequence)
    near = ys[order[starts[best] + int(lens[best]):starts[best] + int(lens[best]) + 8]]
    cy = np.concatenate([cy, near]).astype(np.uint32)
    g = G()
    for i, y in enumerate(cy):
        assert g.set(int(x0), int(y), 1000 + i) == 1000 + i
    bad = []
    stop = time.time() + 3.0

    def reader(t):
        rng = np.random.default_rng(t)
        n = 0
        while time.time() < stop and not bad:
            i = int(rng.integers(0, cy.size))
            v = g.get(int(x0), int(cy[i]))
            if v != 1000 + i:
                bad.append((i, v))
            n += 1
        return n

    th = [threading.Thread(target=reader, args=(t,)) for t in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad[:5]
    assert (g.m.get_batch(np.full(cy.size, x0, np.uint32), cy) == 1000 + np.arange(cy.size)).all()
    g.close()


@pytest.mark.parametrize("mode", ["fold", "locate"])
def test_set_batch_on_present_keys_uses_the_folds_cells(G, oracle_mod, monkeypatch, mode):
    """Round 4: a set batch that round 0 completes (no row created or doubled) ranks its entries at the cells k_set_fold
    found -- cleared there -- and spares the locate pass; highest-index-wins across tiles (src/smatrix.c:225-234 under the
    batch contract) must hold exactly as with the pass (SMATRIX_SET_LOCATE=1).  Zipf keys: hot cells are written from
    hundreds of tiles.  Then a batch that DOES insert and grow rows (the pass runs), then present keys again -- the
    table's addresses have changed in between."""
    if mode == "locate":
        monkeypatch.setenv("SMATRIX_SET_LOCATE", "1")
    rng = np.random.default_rng(99)
    n = 1 << 20
    g, o = G(), oracle_mod.Oracle()
    # (scrambled column ids: an odd multiplier is a bijection of the 32-bit ids.  Dense ids cluster under the reference's
    #  identity hash, the fold then hands its long probes to the lane-per-op retry and the batch is not "completed by round 0")
    scr = lambda a: (a.astype(np.uint64) * np.uint64(2654435761) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x = (rng.zipf(1.3, n) % 3000).astype(np.uint32); y = scr((rng.zipf(1.2, n) % (1 << 20)).astype(np.uint32) + 1)
    ones = np.ones(n, np.uint32)
    g.apply(2, x, y, ones); o.apply(2, x, y, ones)                       # build
    before = g.stats()["set_located_by_fold"]
    for rep in range(3):
        p = rng.permutation(n)
        v = rng.integers(0, 1 << 32, n, dtype=np.uint32)
        a = g.apply(1, x[p], y[p], v); o.apply(1, x[p], y[p], v)
        assert (a == v).all()
        assert (g.apply(0, x, y) == o.apply(0, x, y)).all(), rep
    assert g.stats()["set_located_by_fold"] - before == (3 if mode == "fold" else 0)
    # new keys + growth in a set batch, y == 0 ops in it as well (quirk path, applied in place)
    x2 = (rng.zipf(1.3, n) % 4000).astype(np.uint32); y2 = scr((rng.zipf(1.2, n) % (1 << 21)).astype(np.uint32))
    v2 = rng.integers(1, 1 << 32, n, dtype=np.uint32)
    g.apply(1, x2, y2, v2); o.apply(1, x2, y2, v2)
    nz = y2 != 0
    assert (g.apply(0, x2[nz], y2[nz]) == o.apply(0, x2[nz], y2[nz])).all()
    rows = np.unique(np.concatenate([x, x2]))
    assert (g.m.rowlen_batch(rows) == np.array([o.rowlen(int(r)) for r in rows], np.uint32)).all()
    v3 = rng.integers(0, 1 << 32, n, dtype=np.uint32)
    g.apply(1, x, y, v3); o.apply(1, x, y, v3)
    assert (g.apply(0, x, y) == o.apply(0, x, y)).all()
    g.close(); o.close()
