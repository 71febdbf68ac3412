"""ShardedMatrix -- one matrix row-hash-sharded over the GPUs of a node.

No reference counterpart (the reference is a single process).  The path
shards naturally: rows are independent, so rank r owns the rows with
``smatrix_shard_of(x, world) == r`` (include/smatrix_shard.h) and keeps them in its own
HBM-resident SparseMatrix.  A batch submitted on any rank is

  1. partitioned by owner on the GPU (k_part_count / k_part_scatter),
  2. routed with ``all_to_all_single`` (RCCL over xGMI; gloo in the CPU tests),
  3. applied by each owner's local op kernels,
  4. its results routed back with a second ``all_to_all_single`` and un-permuted.

One process per GPU, ``torch.distributed`` for the exchange; torch is plumbing here
(device buffers + the collective), the compute is the HIP library.

The two device-side pieces are injectable so that the routing logic can be exercised on
CPU with gloo (tests/test_sharded_gloo.py supplies a CPU partitioner and an oracle-backed
shard); the defaults are the HIP implementations and fail loudly without a GPU.
"""
import bisect
import ctypes as C
import json
import os

import torch
import torch.distributed as dist

from . import _lib
from .matrix import OP_GET, SparseMatrix


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Placement:
    """Which shard owns a row: `place` {x: owner} for the few hot rows placed one by one, else the hash
    range that smatrix_shard_mix(x) falls in (`cuts`: world-1 ascending cut points, None = equal ranges)."""

    def __init__(self, world, cuts=None, place=None):
        self.world, self.cuts, self.place = world, cuts, dict(place or {})

    def owner(self, x, lib=None):
        x = int(x) & 0xFFFFFFFF
        if x in self.place:
            return self.place[x]
        lib = lib or _lib.load()
        if self.cuts is None:
            return lib.smatrix_shard_of(x, self.world)
        return bisect.bisect_right(self.cuts, lib.smatrix_shard_mix(x))

    def to_json(self):
        return json.dumps({"world": self.world, "cuts": self.cuts,
                           "place": sorted([int(k), int(v)] for k, v in self.place.items())})

    @staticmethod
    def from_json(text):
        d = json.loads(text)
        return Placement(d["world"], d["cuts"], {int(k): int(v) for k, v in d["place"]})


def plan_placement(counts, total, world, top=256):
    """Skew-aware placement (SURVEY.md 8e "skew handling").  counts: {x: ops seen} for the hottest rows of a
    sample of `total` ops.  Under Zipf(1.1) the hottest row is 12.4 % of all ops: with equal hash ranges its
    owner carries 23 % of the load of 8 shards (1.9x the mean).  Here
      1. the `top` hottest rows are assigned one by one, longest processing time first, to the least
         loaded shard;
      2. the remaining rows -- a flat tail, so a shard's share of it is the width of its hash range -- are
         given RANGES OF UNEQUAL WIDTH that fill every shard up to the same level (a shard already full
         of hot rows gets an empty range).
    Deterministic: every rank computes the same plan from the same gathered counts.  -> Placement"""
    hot = sorted(counts.items(), key=lambda kv: (-kv[1], kv[0]))[:top]
    load = [0] * world
    owner = {}
    for x, c in hot:
        r = min(range(world), key=lambda i: (load[i], i))
        load[r] += c
        owner[x] = r
    tail = max(total - sum(c for _, c in hot), 0)
    # water level L with sum(max(L - load, 0)) == tail
    order = sorted(load)
    level, acc = float(order[-1]) + tail, 0.0          # fallback, overwritten below
    for k in range(1, world + 1):                      # the k least loaded shards share the tail
        acc = sum(order[:k])
        level = (tail + acc) / k
        if k == world or level <= order[k]:
            break
    width = [max(level - l, 0.0) for l in load]
    tot_w = sum(width) or 1.0
    cuts, run = [], 0.0
    for r in range(world - 1):
        run += width[r] / tot_w
        cuts.append(min(int(run * 4294967296.0), 0xFFFFFFFF))
    return Placement(world, cuts, owner)


class HipShard:
    """The local shard: a SparseMatrix driven with device pointers on torch's current stream."""

    def __init__(self, filename=None):
        self.m = SparseMatrix(filename)

    def row_count(self):
        return int(self.m.stats()["rows"])

    def displaced_rows(self, rank, world):
        """rows held here whose hash owner is another shard"""
        lib = _lib.load()
        n = lib.smatrix_displaced_rows(self.m._h, rank, world, None, 0)
        buf = (C.c_uint32 * max(n, 1))()
        n = min(lib.smatrix_displaced_rows(self.m._h, rank, world, C.cast(buf, C.c_void_p), n), n)
        return [int(buf[i]) for i in range(n)]

    def apply(self, op, x, y, v, out):
        n = x.numel()
        if n:
            self.m.apply_batch_dev(op, n, x.data_ptr(), y.data_ptr(), v.data_ptr() if v is not None else None,
                                   out.data_ptr(), _stream())

    def apply_packed(self, op, rec, out):
        """rec: [n, 2] ({x,y}, get) or [n, 3] ({x,y,v}) int32 records, as the exchange delivers them"""
        if rec.shape[0]:
            self.m.apply_packed_dev(op, rec.shape[0], rec.data_ptr(), rec.shape[1], out.data_ptr(), _stream())

    def rowlen(self, x, out):
        if x.numel():
            self.m.rowlen_batch_dev(x.numel(), x.data_ptr(), out.data_ptr(), _stream())

    def getrow(self, x, offsets, pairs, counts):
        """offsets: int64 [n+1] (row r may receive offsets[r+1]-offsets[r] pairs), pairs: int32 [total, 2]"""
        if x.numel():
            self.m.getrow_batch_dev(x.numel(), x.data_ptr(), offsets.data_ptr(), pairs.data_ptr(), counts.data_ptr(), _stream())

    def close(self):
        self.m.close()


class HipPartitioner:
    """partition-by-owner and result gather on the GPU (include/smatrix_shard.h)."""

    def __init__(self, device):
        self.lib = _lib.load()
        self.device = device
        self.work = torch.zeros(128, dtype=torch.int64, device=device)
        self.place = None           # device table [slots, 2] int32 {x, owner + 1}, see include/smatrix_shard.h
        self.place_slots = 0
        self.cuts = None            # device array of world-1 cut points

    def set_placement(self, pl):
        """pl: Placement (or None for equal hash ranges)"""
        self.place, self.place_slots, self.cuts = None, 0, None
        if pl is None:
            return
        i32 = lambda u: u - (1 << 32) if u >= (1 << 31) else u
        if pl.cuts is not None and pl.world > 1:
            self.cuts = torch.tensor([i32(c) for c in pl.cuts], dtype=torch.int32, device=self.device)
        if pl.place:
            slots = 16
            while slots < 2 * len(pl.place):
                slots *= 2
            if slots > 1024:
                raise ValueError("placement table too large (%d rows)" % len(pl.place))
            tab = [[0, 0] for _ in range(slots)]
            for x, owner in pl.place.items():
                i = self.lib.smatrix_place_slot(x, slots)
                while tab[i][1]:
                    i = (i + 1) & (slots - 1)
                tab[i] = [i32(x), owner + 1]
            self.place = torch.tensor(tab, dtype=torch.int32, device=self.device)
            self.place_slots = slots

    def _place_args(self):
        return (self.place.data_ptr() if self.place is not None else None, self.place_slots,
                self.cuts.data_ptr() if self.cuts is not None else None)

    def partition(self, x, y, v, world):
        n = x.numel()
        counts = (C.c_uint64 * world)()
        perm = torch.empty(n, dtype=torch.int32, device=x.device)
        xo, yo = torch.empty_like(x), torch.empty_like(y)
        vo = torch.empty_like(v) if v is not None else None
        rc = self.lib.smatrix_partition_dev(
            n, x.data_ptr(), y.data_ptr(), v.data_ptr() if v is not None else None, world,
            C.cast(counts, _lib.u64p), self.work.data_ptr(), perm.data_ptr(), xo.data_ptr(), yo.data_ptr(),
            vo.data_ptr() if vo is not None else None, *self._place_args(), _stream())
        if rc:
            raise RuntimeError("smatrix_partition_dev failed")
        return [int(c) for c in counts], perm, xo, yo, vo

    def partition_packed(self, x, y, v, world):
        """-> (counts, perm, packed[n, w]) with w = 3 ({x,y,v}) or 2 ({x,y})"""
        n = x.numel()
        counts = (C.c_uint64 * world)()
        perm = torch.empty(n, dtype=torch.int32, device=x.device)
        packed = torch.empty((n, 3 if v is not None else 2), dtype=torch.int32, device=x.device)
        rc = self.lib.smatrix_partition_packed_dev(
            n, x.data_ptr(), y.data_ptr(), v.data_ptr() if v is not None else None, world,
            C.cast(counts, _lib.u64p), self.work.data_ptr(), perm.data_ptr(), packed.data_ptr(),
            *self._place_args(), _stream())
        if rc:
            raise RuntimeError("smatrix_partition_packed_dev failed")
        return [int(c) for c in counts], perm, packed

    def unpack(self, packed):
        n, w = packed.shape
        x = torch.empty(n, dtype=torch.int32, device=packed.device)
        y = torch.empty_like(x)
        v = torch.empty_like(x) if w == 3 else None
        self.lib.smatrix_unpack_dev(n, w, packed.data_ptr(), x.data_ptr(), y.data_ptr(),
                                    v.data_ptr() if v is not None else None, _stream())
        return x, y, v

    def gather(self, src, perm, out):
        self.lib.smatrix_gather_dev(out.numel(), src.data_ptr(), perm.data_ptr(), out.data_ptr(), _stream())


class ShardedMatrix:
    """auto_place: the first write batch of an EMPTY matrix is also the sample from which the placement is
    planned (plan_placement: hot rows by load, the rest in hash ranges of unequal width); with shard files
    it is stored next to them (<file>.placement) and read back on reopen.  Rows never move afterwards."""

    def __init__(self, group=None, shard=None, partitioner=None, device=None, auto_place=True, hot_rows=256,
                 placement_file=None):
        if not dist.is_initialized():
            raise RuntimeError("ShardedMatrix needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if shard is None or partitioner is None:
            device = device or torch.device("cuda", torch.cuda.current_device())
        self.shard = shard if shard is not None else HipShard()
        self.part = partitioner if partitioner is not None else HipPartitioner(device)
        self.exchanged_ops = 0
        self.packed = True                                                # one collective per op batch (three collectives: slower, its switch is gone)
        self.host_staged = os.environ.get("SMATRIX_SHARD_HOST_STAGED", "0") == "1"
        self.auto_place = auto_place and os.environ.get("SMATRIX_SHARD_PLACE", "1") != "0"
        self.hot_rows = hot_rows
        self.placement = Placement(self.world)   # equal hash ranges until planned
        self._placed = False
        self._known_empty = False
        self.placement_file = placement_file
        if placement_file is None and getattr(getattr(self.shard, "m", None), "filename", None):
            self.placement_file = self.shard.m.filename + ".placement"

    # ---- placement ----------------------------------------------------------------------------
    def _gather_obj(self, obj):
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def _ensure_placement(self, x, write=True):
        """COLLECTIVE, before a batch is routed, until settled: decide which shard owns which row.
          * a placement stored next to the shard files -> that one (rows are where it put them);
          * an EMPTY matrix -> planned from this first batch (plan_placement);
          * rows but no stored placement (files written with equal ranges) -> equal ranges, plus whatever
            rows are found away from their range."""
        if self._placed:
            return
        if not (hasattr(self.shard, "row_count") and hasattr(self.part, "set_placement")):
            self._placed = True
            return
        if self._known_empty:
            if not write:
                return                      # reads of an empty matrix find nothing wherever they are routed
            stored, rows = [None] * self.world, [0] * self.world
        else:
            stored = None
            if self.placement_file and os.path.exists(self.placement_file):
                stored = open(self.placement_file).read()
            stored = self._gather_obj(stored)
            rows = self._gather_obj(self.shard.row_count())
            if not sum(rows):
                self._known_empty = True
                if not write:
                    return
        self._placed = True
        if any(t is not None for t in stored) and sum(rows):
            text = next(t for t in stored if t is not None)
            if any(t is not None and t != text for t in stored):
                raise RuntimeError("the shards' stored placements differ")
            pl = Placement.from_json(text)
            if pl.world != self.world:
                raise RuntimeError("stored placement is for %d shards, this group has %d" % (pl.world, self.world))
        elif sum(rows):
            place = {}
            for r, xs in enumerate(self._gather_obj(self.shard.displaced_rows(self.rank, self.world))):
                for v in xs:
                    place[int(v)] = r
            if len(place) > 512:
                raise RuntimeError("%d rows live away from their equal-range owner and no stored placement was found "
                                   "(%s): shard files written under a planned placement need their .placement file"
                                   % (len(place), self.placement_file))
            pl = Placement(self.world, None, place)
        elif self.auto_place and self.world > 1:
            if x.numel():
                ux, cnt = torch.unique(x, return_counts=True)
                top = torch.topk(cnt, min(self.hot_rows, ux.numel())).indices
                mine = {int(a) & 0xFFFFFFFF: int(c) for a, c in zip(ux[top].tolist(), cnt[top].tolist())}
            else:
                mine = {}
            counts, total = {}, 0
            for part, n in self._gather_obj((mine, int(x.numel()))):
                total += n
                for a, c in part.items():
                    counts[a] = counts.get(a, 0) + c
            pl = plan_placement(counts, total, self.world, self.hot_rows)
        else:
            pl = Placement(self.world)
        self.placement = pl
        self.part.set_placement(pl)

    def _a2a(self, out, inp, out_splits=None, in_splits=None):
        """all_to_all_single; with SMATRIX_SHARD_HOST_STAGED=1 the payload is staged through host
        memory (lets two ranks share ONE GPU over gloo -- test rigs only, RCCL refuses duplicate GPUs)."""
        if self.host_staged and inp.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    @property
    def local(self):
        """the rank's own SparseMatrix (stats / profiling)"""
        return self.shard.m

    def owner(self, x):
        return self.placement.owner(x)

    def apply_dev(self, op, x, y, v, out):
        """x, y, v, out: 1-D int32 tensors on this rank's device (v None for get).
        COLLECTIVE: every rank must call it with the same op (batch sizes may differ)."""
        vv = None if op == OP_GET else v
        self._ensure_placement(x, op != OP_GET)
        packed_path = self.packed and hasattr(self.part, "partition_packed")
        if packed_path:
            counts, perm, po = self.part.partition_packed(x, y, vv, self.world)
        else:
            counts, perm, xo, yo, vo = self.part.partition(x, y, vv, self.world)
        send = torch.tensor(counts, dtype=torch.int64, device=x.device)
        recv = torch.empty(self.world, dtype=torch.int64, device=x.device)
        self._a2a(recv, send)
        rcounts = [int(c) for c in recv.tolist()]
        nr = sum(rcounts)
        if packed_path:
            # one collective for the whole op record (splits count rows of the [n, w] tensor)
            pr = torch.empty((nr, po.shape[1]), dtype=x.dtype, device=x.device)
            self._a2a(pr, po, rcounts, counts)
            if not hasattr(self.shard, "apply_packed"):
                xr, yr, vr = self.part.unpack(pr)
        else:
            xr = torch.empty(nr, dtype=x.dtype, device=x.device)
            yr = torch.empty(nr, dtype=x.dtype, device=x.device)
            self._a2a(xr, xo, rcounts, counts)
            self._a2a(yr, yo, rcounts, counts)
            vr = None
            if op != OP_GET:
                vr = torch.empty(nr, dtype=x.dtype, device=x.device)
                self._a2a(vr, vo, rcounts, counts)
        outr = torch.empty(nr, dtype=x.dtype, device=x.device)
        if packed_path and hasattr(self.shard, "apply_packed"):
            self.shard.apply_packed(op, pr, outr)            # the op kernels read the records in place
        else:
            self.shard.apply(op, xr, yr, vr, outr)
        back = torch.empty(x.numel(), dtype=x.dtype, device=x.device)
        self._a2a(back, outr, counts, rcounts)
        self.part.gather(back, perm, out)
        self.exchanged_ops += nr
        return out

    def rowlen_dev(self, xs, out):
        """rowlen of arbitrary rows: routed to the owners like an op batch (rows are shard-local, so a
        scan of a rank's OWN rows -- local.rowlen_batch / local.getrow_batch -- needs no exchange at all).
        COLLECTIVE."""
        self._ensure_placement(xs, False)
        counts, perm, xo, _, _ = self.part.partition(xs, xs, None, self.world)
        send = torch.tensor(counts, dtype=torch.int64, device=xs.device)
        recv = torch.empty(self.world, dtype=torch.int64, device=xs.device)
        self._a2a(recv, send)
        rcounts = [int(c) for c in recv.tolist()]
        xr = torch.empty(sum(rcounts), dtype=xs.dtype, device=xs.device)
        self._a2a(xr, xo, rcounts, counts)
        lr = torch.empty_like(xr)
        self.shard.rowlen(xr, lr)
        back = torch.empty(xs.numel(), dtype=xs.dtype, device=xs.device)
        self._a2a(back, lr, counts, rcounts)
        self.part.gather(back, perm, out)
        return out

    def getrow_dev(self, xs):
        """getrow of arbitrary rows (include/smatrix.h smatrix_getrow, batched): every row is read on its owner in
        slot order and sent back.  -> (offsets int64 [n+1], pairs int32 [total, 2] = {y, value}, counts int32 [n]);
        row i's pairs are pairs[offsets[i] : offsets[i] + counts[i]] (counts[i] == offsets[i+1] - offsets[i]).
        COLLECTIVE.  A scan of a rank's OWN rows (local.getrow_batch) needs no exchange at all."""
        dev = xs.device
        self._ensure_placement(xs, False)
        counts, perm, xo, _, _ = self.part.partition(xs, xs, None, self.world)
        send = torch.tensor(counts, dtype=torch.int64, device=dev)
        recv = torch.empty(self.world, dtype=torch.int64, device=dev)
        self._a2a(recv, send)
        rcounts = [int(c) for c in recv.tolist()]
        nr = sum(rcounts)
        xr = torch.empty(nr, dtype=xs.dtype, device=dev)
        self._a2a(xr, xo, rcounts, counts)
        # owners: lengths, then the rows themselves into one buffer (capacity = rowlen: all pairs, no Q5 overrun)
        lr = torch.empty(nr, dtype=torch.int32, device=dev)
        self.shard.rowlen(xr, lr)
        offr = torch.zeros(nr + 1, dtype=torch.int64, device=dev)
        torch.cumsum(lr.to(torch.int64), 0, out=offr[1:])
        tot_r = int(offr[-1].item()) if nr else 0
        pr = torch.empty((tot_r, 2), dtype=torch.int32, device=dev)
        cr = torch.zeros(nr, dtype=torch.int32, device=dev)
        self.shard.getrow(xr, offr, pr, cr)
        # back: the lengths (row order of the partition), then the pairs, split by requesting rank
        lens_p = torch.empty(xs.numel(), dtype=torch.int32, device=dev)
        self._a2a(lens_p, cr, counts, rcounts)
        bounds = [0]
        for c in rcounts:
            bounds.append(bounds[-1] + c)
        psend = [int((offr[bounds[i + 1]] - offr[bounds[i]]).item()) for i in range(self.world)] if nr else [0] * self.world
        ps = torch.tensor(psend, dtype=torch.int64, device=dev)
        prcv = torch.empty(self.world, dtype=torch.int64, device=dev)
        self._a2a(prcv, ps)
        precv = [int(c) for c in prcv.tolist()]
        pairs_p = torch.empty((sum(precv), 2), dtype=torch.int32, device=dev)
        self._a2a(pairs_p, pr, precv, psend)
        # un-permute: row i of the request sits at position perm[i] of the partitioned order
        lens = torch.empty_like(lens_p)
        self.part.gather(lens_p, perm, lens)
        offsets = torch.zeros(xs.numel() + 1, dtype=torch.int64, device=dev)
        torch.cumsum(lens.to(torch.int64), 0, out=offsets[1:])
        off_p = torch.zeros(xs.numel() + 1, dtype=torch.int64, device=dev)
        torch.cumsum(lens_p.to(torch.int64), 0, out=off_p[1:])
        total = int(offsets[-1].item()) if xs.numel() else 0
        pairs = torch.empty((total, 2), dtype=torch.int32, device=dev)
        if total:
            # pair k of the output belongs to request row i = row_of[k]; it is pair (k - offsets[i]) of partitioned row perm[i]
            row_of = torch.repeat_interleave(torch.arange(xs.numel(), device=dev), lens.to(torch.int64))
            src = off_p[perm.long()[row_of]] + (torch.arange(total, device=dev) - offsets[row_of])
            pairs.copy_(pairs_p[src])
        return offsets, pairs, lens

    # ---- split-phase form: lets the exchange of one batch overlap the op kernels of another -------
    #
    #   h = route(op, x, y, v)     partition + all_to_all of the op records   (communication stream)
    #   apply_routed(h)            the owner's op kernels on what arrived      (compute stream)
    #   finish(h, out)             all_to_all of the results back + gather     (communication stream)
    #   wait(h)                    compute stream waits for `out`
    #
    # route() of batch B may be issued before apply_routed() of batch A returns to the host: the
    # library blocks the host only on the compute stream, so the records of B travel while A's kernels
    # run.  bench.py pipelines  incr(s) | get(s) | incr(s+1)  this way.  On CPU tensors (gloo tests)
    # there are no streams and the phases simply run in order.
    class _Routed:
        pass

    def _comm(self, like):
        if not like.is_cuda:
            return None
        if getattr(self, "_comm_stream", None) is None:
            self._comm_stream = torch.cuda.Stream(device=like.device)
        return self._comm_stream

    def route(self, op, x, y, v=None, inputs_ready=False):
        """inputs_ready: the caller guarantees that x, y, v are complete (e.g. produced before an earlier
        synchronisation) -- the exchange then does not wait for whatever the compute stream is running now."""
        h = self._Routed()
        h.op, h.n, h.x_dev = op, x.numel(), x.device
        comm = self._comm(x)
        vv = None if op == OP_GET else v
        self._ensure_placement(x, op != OP_GET)
        ctx = torch.cuda.stream(comm) if comm is not None else _null_ctx()
        if comm is not None and not inputs_ready:
            comm.wait_stream(torch.cuda.current_stream())         # inputs were produced on the compute stream
        with ctx:
            packed_path = self.packed and hasattr(self.part, "partition_packed")
            if packed_path:
                h.counts, h.perm, po = self.part.partition_packed(x, y, vv, self.world)
            else:
                h.counts, h.perm, xo, yo, vo = self.part.partition(x, y, vv, self.world)
            send = torch.tensor(h.counts, dtype=torch.int64, device=x.device)
            recv = torch.empty(self.world, dtype=torch.int64, device=x.device)
            self._a2a(recv, send)
            h.rcounts = [int(c) for c in recv.tolist()]
            nr = sum(h.rcounts)
            if packed_path:
                pr = torch.empty((nr, po.shape[1]), dtype=x.dtype, device=x.device)
                self._a2a(pr, po, h.rcounts, h.counts)
                h.pr = pr if hasattr(self.shard, "apply_packed") else None
                h.xr, h.yr, h.vr = (None, None, None) if h.pr is not None else self.part.unpack(pr)
                h.keep = (x, y, v, po, pr)
            else:
                h.xr = torch.empty(nr, dtype=x.dtype, device=x.device)
                h.yr = torch.empty(nr, dtype=x.dtype, device=x.device)
                self._a2a(h.xr, xo, h.rcounts, h.counts)
                self._a2a(h.yr, yo, h.rcounts, h.counts)
                h.vr = None
                if vv is not None:
                    h.vr = torch.empty(nr, dtype=x.dtype, device=x.device)
                    self._a2a(h.vr, vo, h.rcounts, h.counts)
                h.pr = None
                h.keep = (x, y, v, xo, yo, vo)
            h.outr = torch.empty(nr, dtype=x.dtype, device=x.device)
            h.ev_routed = comm.record_event() if comm is not None else None
        self.exchanged_ops += nr
        return h

    def apply_routed(self, h):
        if h.ev_routed is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(h.ev_routed)
            for t in (h.xr, h.yr, h.vr, h.pr, h.outr):
                if t is not None:
                    t.record_stream(cur)                          # allocated under the communication stream
        if h.pr is not None:
            self.shard.apply_packed(h.op, h.pr, h.outr)           # the op kernels read the records in place
        else:
            self.shard.apply(h.op, h.xr, h.yr, h.vr, h.outr)
        h.ev_applied = torch.cuda.current_stream().record_event() if h.ev_routed is not None else None

    def apply_routed_get(self, h):
        """a get on the SAME keys right behind the write of apply_routed(h) -- the benchmark's step (incr batch, then get
        batch, src/smatrix_benchmark.c:226-230) with ONE partition and ONE exchange of the op records: the owner reads
        {x, y} of the records it already holds.  finish(h, out, out_get) then returns both result sets."""
        assert h.op != OP_GET
        dev = h.outr.device
        h.outg = torch.empty(h.outr.numel(), dtype=h.outr.dtype, device=dev)
        if h.ev_routed is not None:
            h.outg.record_stream(torch.cuda.current_stream())
        if h.pr is not None:
            self.shard.apply_packed(OP_GET, h.pr, h.outg)         # width-3 records: the get kernel skips the value word
        else:
            self.shard.apply(OP_GET, h.xr, h.yr, None, h.outg)
        h.ev_applied = torch.cuda.current_stream().record_event() if h.ev_routed is not None else None
        self.exchanged_ops += h.outg.numel()

    def finish(self, h, out, out_get=None):
        comm = self._comm(out)
        ctx = torch.cuda.stream(comm) if comm is not None else _null_ctx()
        with ctx:
            if comm is not None:
                comm.wait_event(h.ev_applied)
                out.record_stream(comm)
            back = torch.empty(h.n, dtype=out.dtype, device=out.device)
            self._a2a(back, h.outr, h.counts, h.rcounts)
            self.part.gather(back, h.perm, out)
            h.keep = h.keep + (back,)
            if out_get is not None:
                if comm is not None:
                    out_get.record_stream(comm)
                back2 = torch.empty(h.n, dtype=out.dtype, device=out.device)
                self._a2a(back2, h.outg, h.counts, h.rcounts)
                self.part.gather(back2, h.perm, out_get)
                h.keep = h.keep + (back2,)
            h.ev_done = comm.record_event() if comm is not None else None
        return h

    def wait(self, h):
        if getattr(h, "ev_done", None) is not None:
            torch.cuda.current_stream().wait_event(h.ev_done)
        h.keep = None

    def close(self):
        if self.placement_file and self._placed:
            with open(self.placement_file, "w") as f:
                f.write(self.placement.to_json())
        self.shard.close()


class _null_ctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class NativeShardedMatrix:
    """The router that lives in the C library (include/smatrix_shard.h, csrc/smx_shard.inc): the library partitions,
    exchanges counts / records / results (RCCL, or the shared-memory test transport: SMATRIX_SHARD_TRANSPORT=shm),
    plans and persists the placement and applies -- Python only hands the 128-byte id from rank 0 to the other ranks
    (here through torch.distributed when it is initialised; any launcher can do that) and passes device pointers.
    Every call is collective."""

    def __init__(self, filename=None, rank=None, world=None, unique_id=None):
        self._lib = _lib.load()
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
            world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank, self.world = rank, world
        if world > 1 and unique_id is None:
            box = [None]
            if rank == 0:
                buf = C.create_string_buffer(128)
                if self._lib.smatrix_shard_unique_id(buf) == 0:
                    box[0] = bytes(buf.raw)
            dist.broadcast_object_list(box, src=0)               # (None when rank 0 has no RCCL: the others must not wait for an id that never comes)
            unique_id = box[0]
            if unique_id is None:
                raise RuntimeError("RCCL is not available on rank 0 (smatrix_shard_unique_id)")
        idbuf = C.create_string_buffer(unique_id, 128) if unique_id else None
        self._h = self._lib.smatrix_shard_open(filename.encode() if filename else None, rank, world, idbuf)
        if not self._h:
            raise ValueError("smatrix_shard_open() failed")
        self.local = _BorrowedMatrix(self._lib, self._lib.smatrix_shard_local(self._h))

    @staticmethod
    def _ok(rc, what):
        if rc:
            raise RuntimeError("%s failed (%d)" % (what, rc))

    @property
    def transport(self):
        return self._lib.smatrix_shard_transport(self._h).decode()

    @property
    def placement(self):
        """the placement in force (planned by the library from the first write batch, loaded from <file>.placement, or set)"""
        cuts = (C.c_uint32 * 64)()
        pairs = (C.c_uint32 * 2048)()
        nc, nr = C.c_uint32(0), C.c_uint32(0)
        self._lib.smatrix_shard_get_placement(self._h, cuts, C.byref(nc), pairs, 1024, C.byref(nr))
        return Placement(self.world, [int(cuts[i]) for i in range(nc.value)] if nc.value else None,
                         {int(pairs[2 * i]): int(pairs[2 * i + 1]) for i in range(min(nr.value, 1024))})

    def apply_dev(self, op, x, y, v, out, stream=None):
        self._ok(self._lib.smatrix_shard_apply_dev(self._h, op, x.numel(), x.data_ptr(), y.data_ptr(),
                                                   v.data_ptr() if v is not None else None, out.data_ptr(), stream), "smatrix_shard_apply_dev")

    def apply_then_get_dev(self, op, x, y, v, out, out_get, stream=None):
        self._ok(self._lib.smatrix_shard_apply_then_get_dev(self._h, op, x.numel(), x.data_ptr(), y.data_ptr(), v.data_ptr(),
                                                            out.data_ptr(), out_get.data_ptr(), stream), "smatrix_shard_apply_then_get_dev")

    # split phases (include/smatrix_shard.h): h = route(...); apply_routed(h); finish(h, out[, out_get]); wait(h)
    def route(self, op, x, y, v=None, inputs_ready=False, stream=None):
        h = self._lib.smatrix_shard_route_dev(self._h, op, x.numel(), x.data_ptr(), y.data_ptr(),
                                              v.data_ptr() if v is not None else None, int(inputs_ready), stream)
        if not h:
            raise RuntimeError("smatrix_shard_route_dev failed")
        return (h, (x, y, v))                       # (the arrays must stay alive until the batch has been routed)

    def apply_routed(self, h, then_get=False, stream=None):
        self._ok(self._lib.smatrix_shard_apply_routed(self._h, h[0], int(then_get), stream), "smatrix_shard_apply_routed")

    def finish(self, h, out, out_get=None):
        self._ok(self._lib.smatrix_shard_finish(self._h, h[0], out.data_ptr(), out_get.data_ptr() if out_get is not None else None),
                 "smatrix_shard_finish")

    def wait(self, h, stream=None):
        self._ok(self._lib.smatrix_shard_wait(self._h, h[0], stream), "smatrix_shard_wait")

    def rowlen_dev(self, xs, out, stream=None):
        self._ok(self._lib.smatrix_shard_rowlen_dev(self._h, xs.numel(), xs.data_ptr(), out.data_ptr(), stream), "smatrix_shard_rowlen_dev")
        return out

    def getrow_dev(self, xs, offsets, pairs, counts, stream=None):
        """offsets int64 [n+1], pairs int32 [total, 2], counts int32 [n] (as SparseMatrix.getrow_batch_dev)"""
        self._ok(self._lib.smatrix_shard_getrow_dev(self._h, xs.numel(), xs.data_ptr(), offsets.data_ptr(), pairs.data_ptr(),
                                                    counts.data_ptr(), stream), "smatrix_shard_getrow_dev")

    def set_placement(self, placement):
        """a Placement chosen by the caller -> the library's device tables; identical on every rank"""
        import numpy as np
        cuts = np.array(placement.cuts, dtype=np.uint32) if placement.cuts is not None else None
        slots, table = 0, None
        if placement.place:
            slots = 16
            while slots < 2 * len(placement.place):
                slots *= 2
            table = np.zeros((slots, 2), dtype=np.uint32)
            for x, owner in sorted(placement.place.items()):
                i = self._lib.smatrix_place_slot(x, slots)
                while table[i, 1]:
                    i = (i + 1) % slots
                table[i] = (x, owner + 1)
        rc = self._lib.smatrix_shard_set_placement(
            self._h, cuts.ctypes.data_as(C.c_void_p) if cuts is not None else None,
            table.ctypes.data_as(C.c_void_p) if table is not None else None, slots)
        if rc:
            raise ValueError("smatrix_shard_set_placement failed")

    @property
    def exchanged_ops(self):
        return int(self._lib.smatrix_shard_ops_applied(self._h))

    def close(self):
        if self._h:
            self.local._h = None
            self._lib.smatrix_shard_close(self._h)
            self._h = None


class _BorrowedMatrix(SparseMatrix):
    """the local shard of a NativeShardedMatrix as a SparseMatrix (owned and closed by the shard handle)"""

    def __init__(self, lib, handle):
        self._lib, self._h, self.filename = lib, handle, None

    def close(self):
        self._h = None
