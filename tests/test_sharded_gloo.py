"""CPU, world_size 2, gloo: the multi-GPU routing path (partition -> all_to_all -> local apply ->
all_to_all back -> un-permute) of libsmatrix_amd/sharded.py against a single-matrix oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("packed,split,place,world", [("0", "0", "1", 2), ("1", "0", "1", 2), ("1", "1", "1", 2),
                                                      ("1", "1", "0", 2), ("1", "0", "1", 4), ("1", "1", "1", 8)])
def test_sharded_ranks_gloo(packed, split, place, world):
    """place=1: skew-aware placement planned from the first batch (hot rows one by one + unequal hash
    ranges); place=0: equal hash ranges.  world 4 and 8 (the node size the bench is scaled to) exercise several cut points."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SMX_TEST_PACKED=packed, SMX_TEST_SPLIT=split, SMX_TEST_PLACE=place)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29533 + int(packed) + 2 * int(split) + 4 * int(place) + 8 * world),
           os.path.join(ROOT, "tests", "sharded_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "SHARDED_OK world=%d" % world in p.stdout
    if place == "1":
        assert "placement:" in p.stdout


def test_plan_placement_balances_zipf():
    """the planner alone on the analytic Zipf(1.1) row marginal of config 2/4, 8 shards: equal hash ranges
    leave the hottest row's owner at ~1.9x the mean; the plan is flat"""
    import numpy as np
    sys.path.insert(0, ROOT)
    from libsmatrix_amd.sharded import Placement, plan_placement
    n_ids, total, world = 1000000, 8 << 24, 8
    p = np.arange(1, n_ids + 1) ** -1.1
    p /= p.sum()
    counts = {i + 1: int(p[i] * total) for i in range(2000)}
    pl = plan_placement(counts, total, world, 256)
    assert len(pl.place) == 256 and pl.cuts == sorted(pl.cuts) and len(pl.cuts) == world - 1
    hot = dict(sorted(counts.items(), key=lambda kv: -kv[1])[:256])
    load = np.zeros(world)
    for x, c in hot.items():
        load[pl.place[x]] += c
    edges = np.array([0] + pl.cuts + [1 << 32], dtype=np.float64)
    load += (total - sum(hot.values())) * np.diff(edges) / 2.0 ** 32
    assert load.max() / (total / world) < 1.01, load / (total / world)
    assert float(p[0]) * world > 0.99                      # why: the hottest row alone is one shard's fair share
    back = Placement.from_json(pl.to_json())
    assert back.cuts == pl.cuts and back.place == pl.place and back.world == world


def test_c_planner_equals_python_planner():
    """The C library's router plans its own placement (csrc/smx_shard.inc plan_placement) and stores it as the JSON the
    Python router reads: on the same samples both planners must produce the SAME plan -- hot rows, owners, cut points --
    for 2 to 8 shards, on the analytic Zipf marginal, on ties and on degenerate inputs; the C parser must round-trip it."""
    import ctypes as C
    import json
    import numpy as np
    sys.path.insert(0, ROOT)
    from libsmatrix_amd import _lib
    from libsmatrix_amd.sharded import Placement, plan_placement
    lib = _lib.load()

    def c_plan(counts, total, world, reparse):
        xs = np.array(sorted(counts), dtype=np.uint32)
        cs = np.array([counts[int(x)] for x in xs], dtype=np.uint64)
        buf = C.create_string_buffer(1 << 16)
        n = lib.smatrix_shard_plan_json(xs.ctypes.data_as(C.c_void_p), cs.ctypes.data_as(C.c_void_p), xs.size, total, world, reparse, buf, len(buf))
        assert n > 0
        return Placement.from_json(buf.value.decode())

    rng = np.random.default_rng(8)
    p = np.arange(1, 1000001) ** -1.1
    p /= p.sum()
    cases = []
    for world in (2, 3, 4, 8):
        total = world << 20
        cases.append(({int(rng.integers(1, 1 << 32)) if k else 0xFFFFFFFF: int(p[k] * total) + 1 for k in range(700)}, total, world))
        cases.append(({int(x): 7 for x in rng.integers(1, 1 << 31, 300)}, 5000, world))          # ties: broken by row id
        cases.append(({5: 100}, 100, world))                                                     # one row is the whole stream
        cases.append(({}, 0, world))                                                             # nothing sampled
    for counts, total, world in cases:
        want = plan_placement(dict(counts), total, world, 256)
        for reparse in (0, 1):
            got = c_plan(counts, total, world, reparse)
            assert got.world == want.world == world
            assert got.place == want.place, (world, len(counts))
            assert got.cuts == want.cuts, (world, got.cuts, want.cuts)
