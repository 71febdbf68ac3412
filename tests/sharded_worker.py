"""Worker for tests/test_sharded_gloo.py: run under torch.distributed.run with the gloo backend.
The routing logic under test is libsmatrix_amd/sharded.py; the two device-side pieces are
replaced by CPU stand-ins (a numpy partitioner and an ORACLE-backed shard -- checker code,
which is why they live under tests/)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from libsmatrix_amd import _lib, Stream  # noqa: E402
from libsmatrix_amd.sharded import ShardedMatrix  # noqa: E402
from oracle import oracle as O  # noqa: E402


def shard_mix_np(x):
    """numpy restatement of shard_mix() in smx_kernels.hpp (cross-checked against the library)"""
    h = x.astype(np.uint32) ^ np.uint32(0x9E3779B9)
    h ^= h >> np.uint32(16); h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
    h ^= h >> np.uint32(13); h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def shard_of_np(x, world):
    return ((shard_mix_np(x).astype(np.uint64) * np.uint64(world)) >> np.uint64(32)).astype(np.int64)


def owner_np(x, world, pl):
    """owner of every row id under a Placement (None = equal hash ranges)"""
    if pl is None or pl.cuts is None:
        own = shard_of_np(x, world)
    else:
        own = np.searchsorted(np.array(pl.cuts, dtype=np.uint64), shard_mix_np(x).astype(np.uint64), side="right").astype(np.int64)
    if pl is not None and pl.place:
        keys = np.array(sorted(pl.place), dtype=np.uint32)
        vals = np.array([pl.place[int(k)] for k in keys], dtype=np.int64)
        pos = np.searchsorted(keys, x)
        hit = (pos < keys.size) & (keys[np.minimum(pos, keys.size - 1)] == x)
        own = np.where(hit, vals[np.minimum(pos, keys.size - 1)], own)
    return own


class NumpyPartitioner:
    pl = None

    def set_placement(self, pl):
        self.pl = pl

    def partition(self, x, y, v, world):
        xn = x.numpy().view(np.uint32)
        own = owner_np(xn, world, self.pl)
        order = np.argsort(own, kind="stable")
        perm = np.empty_like(order); perm[order] = np.arange(order.size)
        counts = np.bincount(own, minlength=world).tolist()
        o = torch.from_numpy(order)
        return counts, torch.from_numpy(perm.astype(np.int32)), x[o], y[o], (v[o] if v is not None else None)

    def gather(self, src, perm, out):
        out.copy_(src[perm.long()])


class NumpyPackedPartitioner(NumpyPartitioner):
    """adds the one-collective record form ([n, w] rows), like HipPartitioner"""

    def partition_packed(self, x, y, v, world):
        counts, perm, xo, yo, vo = self.partition(x, y, v, world)
        cols = [xo, yo] + ([vo] if vo is not None else [])
        return counts, perm, torch.stack(cols, 1).contiguous()

    def unpack(self, packed):
        return (packed[:, 0].contiguous(), packed[:, 1].contiguous(),
                packed[:, 2].contiguous() if packed.shape[1] == 3 else None)


class OracleShard:
    def __init__(self, fname=None):
        self.m = O.Oracle(fname)

    def row_count(self):
        return self.m.num_rows()

    def displaced_rows(self, rank, world):
        rows = self.m.list_rows()
        return rows[shard_of_np(rows, world) != rank].tolist()

    def apply(self, op, x, y, v, out):
        r = self.m.apply(op, x.numpy().view(np.uint32), y.numpy().view(np.uint32),
                         v.numpy().view(np.uint32) if v is not None else None)
        out.copy_(torch.from_numpy(r.view(np.int32)))

    def rowlen(self, x, out):
        xs = x.numpy().view(np.uint32)
        out.copy_(torch.tensor([self.m.rowlen(int(v)) for v in xs], dtype=torch.int64).to(torch.int32))

    def getrow(self, x, offsets, pairs, counts):
        xs = x.numpy().view(np.uint32)
        for i, v in enumerate(xs.tolist()):
            row = self.m.getrow(v, int(offsets[i + 1] - offsets[i]) * 8)
            pairs[int(offsets[i]): int(offsets[i]) + row.shape[0]] = torch.from_numpy(row.view(np.int32))
            counts[i] = row.shape[0]

    def close(self):
        self.m.close()


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = _lib.load()
    probe = np.array([0, 1, 2, 12345, 0xFFFFFFFF, 0x80000000, 777777], dtype=np.uint32)
    assert [lib.smatrix_shard_of(int(p), world) for p in probe] == shard_of_np(probe, world).tolist()
    assert [lib.smatrix_shard_mix(int(p)) for p in probe] == shard_mix_np(probe).tolist()

    packed = os.environ.get("SMX_TEST_PACKED") == "1"
    sm = ShardedMatrix(shard=OracleShard(), partitioner=NumpyPackedPartitioner() if packed else NumpyPartitioner(),
                       auto_place=os.environ.get("SMX_TEST_PLACE", "1") == "1")
    n = 30000 + 1000 * rank                          # ragged batch sizes
    gen = Stream("zipf", 12345 + rank, 50000, 1.1, 1)
    x, y = gen.fill(0, n)
    xt, yt = torch.from_numpy(x.view(np.int32)), torch.from_numpy(y.view(np.int32))
    ones = torch.ones(n, dtype=torch.int32)
    out_i = torch.empty(n, dtype=torch.int32); out_g = torch.empty(n, dtype=torch.int32)
    if os.environ.get("SMX_TEST_SPLIT") == "1":
        # split-phase form, issued in bench.py's pipelined order: route(get) before apply(incr)
        h_i = sm.route(2, xt, yt, ones)
        h_g = sm.route(0, xt, yt)
        sm.apply_routed(h_i); sm.finish(h_i, out_i)
        sm.apply_routed(h_g); sm.finish(h_g, out_g)
        sm.wait(h_i); sm.wait(h_g)
        # the fused form (one partition, one record exchange for write + get): decr what was added, read back
        out_d = torch.empty(n, dtype=torch.int32); out_g2 = torch.empty(n, dtype=torch.int32)
        h = sm.route(3, xt, yt, ones)
        sm.apply_routed(h); sm.apply_routed_get(h); sm.finish(h, out_d, out_g2); sm.wait(h)
        assert not out_g2.any(), "fused decr+get: every cell must be back at 0"
        h = sm.route(2, xt, yt, ones)
        sm.apply_routed(h); sm.apply_routed_get(h); sm.finish(h, out_d, out_g2); sm.wait(h)
        assert (out_g2 == out_g).all(), "fused incr+get != separate incr, get"
    else:
        sm.apply_dev(2, xt, yt, ones, out_i)
        sm.apply_dev(0, xt, yt, None, out_g)
    # an empty batch on one rank must not hang the collective
    e = torch.empty(0, dtype=torch.int32)
    sm.apply_dev(0, e if rank == 0 else xt[:10], e if rank == 0 else yt[:10], None,
                 torch.empty(0 if rank == 0 else 10, dtype=torch.int32))

    # expected: one oracle over the ops of ALL ranks
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n]))
    mx = max(int(s) for s in sizes)
    pad = torch.zeros((2, mx), dtype=torch.int32); pad[0, :n] = xt; pad[1, :n] = yt
    allp = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(allp, pad)
    ref = O.Oracle()
    for r in range(world):
        k = int(sizes[r])
        ref.apply(O.OP_INCR, allp[r][0, :k].numpy().view(np.uint32), allp[r][1, :k].numpy().view(np.uint32),
                  np.ones(k, np.uint32))
    want = ref.apply(O.OP_GET, x, y)
    assert (out_g.numpy().view(np.uint32) == want).all(), "sharded get != single-matrix oracle"
    # incr returns: a legal serialisation => per key, the largest return over all ranks == final value
    key = x.astype(np.uint64) << 32 | y
    uk, inv = np.unique(key, return_inverse=True)
    mine = np.zeros(uk.size, np.uint32); np.maximum.at(mine, inv, out_i.numpy().view(np.uint32))
    assert (mine <= ref.apply(O.OP_GET, (uk >> 32).astype(np.uint32), (uk & 0xFFFFFFFF).astype(np.uint32))).all()
    # rowlen of arbitrary rows, routed to the owners
    q = torch.from_numpy(np.unique(x)[:3000].view(np.int32).copy())
    lens = torch.empty_like(q)
    sm.rowlen_dev(q, lens)
    assert lens.tolist() == [ref.rowlen(int(v)) for v in q.numpy().view(np.uint32)]
    # getrow of arbitrary rows: read on the owners in slot order, same pairs as the single matrix holds
    qg = q[:400]
    off, prs, cnt = sm.getrow_dev(qg)
    assert cnt.tolist() == lens[:400].tolist()
    for i, xv in enumerate(qg.numpy().view(np.uint32).tolist()):
        mine = prs[int(off[i]): int(off[i]) + int(cnt[i])].numpy().view(np.uint32)
        want_row = ref.getrow(xv, ref.rowlen(xv) * 8)
        assert sorted(map(tuple, mine.tolist())) == sorted(map(tuple, want_row.tolist())), xv
    # every row lives on exactly its owner, with the same length as in the single matrix
    rows = sm.shard.m.list_rows()
    assert (owner_np(rows, world, sm.placement) == rank).all(), "a row landed on the wrong shard"
    assert all(sm.owner(int(r)) == rank for r in rows[:300])
    if sm.auto_place:
        # the plan moved hot rows and narrowed the loaded shards' ranges: op counts per shard within 15 %
        # of the mean (equal ranges on this stream: the owner of the hottest row carries ~1.6x)
        own = owner_np(np.concatenate([allp[r][0, :int(sizes[r])].numpy().view(np.uint32) for r in range(world)]),
                       world, sm.placement)
        share = np.bincount(own, minlength=world) / own.size * world
        assert sm.placement.place and share.max() < 1.15, share
        if rank == 0:
            print("placement: %d rows placed, load/mean per shard %s" % (len(sm.placement.place), np.round(share, 3).tolist()))
    assert all(sm.shard.m.rowlen(int(r)) == ref.rowlen(int(r)) for r in rows[:500])
    tot = torch.tensor([rows.size]); dist.all_reduce(tot)
    assert int(tot) == ref.num_rows()
    sm.close(); ref.close()
    # file-backed shards: the plan is stored next to them and taken over at reopen (reads first, then writes)
    import tempfile
    d = [tempfile.mkdtemp(prefix="smxgloo_") if rank == 0 else None]
    dist.broadcast_object_list(d, src=0)
    fname = os.path.join(d[0], "shard%d.smx" % rank)
    mk = lambda: ShardedMatrix(shard=OracleShard(fname), partitioner=NumpyPackedPartitioner(), placement_file=fname + ".placement",
                               auto_place=os.environ.get("SMX_TEST_PLACE", "1") == "1")
    fm = mk()
    fm.apply_dev(2, xt, yt, ones, out_i)
    fm.apply_dev(0, xt, yt, None, out_g)
    want_g, plan = out_g.clone(), fm.placement.to_json()
    fm.close()
    fm = mk()
    got = torch.empty_like(out_g)
    fm.apply_dev(0, xt, yt, None, got)                       # a READ is the first call after the reopen
    assert fm.placement.to_json() == plan and torch.equal(got, want_g)
    fm.apply_dev(2, xt, yt, ones, out_i)
    fm.apply_dev(0, xt, yt, None, got)
    assert torch.equal(got, want_g * 2)
    fm.close()
    dist.barrier()
    if rank == 0:
        import shutil
        shutil.rmtree(d[0], ignore_errors=True)
        print("SHARDED_OK world=%d rows=%d" % (world, int(tot)))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
