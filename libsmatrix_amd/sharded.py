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
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib
from .matrix import OP_GET, SparseMatrix


def _stream():
    return torch.cuda.current_stream().cuda_stream


class HipShard:
    """The local shard: a SparseMatrix driven with device pointers on torch's current stream."""

    def __init__(self, filename=None):
        self.m = SparseMatrix(filename)

    def apply(self, op, x, y, v, out):
        n = x.numel()
        if n:
            self.m.apply_batch_dev(op, n, x.data_ptr(), y.data_ptr(), v.data_ptr() if v is not None else None,
                                   out.data_ptr(), _stream())

    def rowlen(self, x, out):
        if x.numel():
            self.m.rowlen_batch_dev(x.numel(), x.data_ptr(), out.data_ptr(), _stream())

    def close(self):
        self.m.close()


class HipPartitioner:
    """partition-by-owner and result gather on the GPU (include/smatrix_shard.h)."""

    def __init__(self, device):
        self.lib = _lib.load()
        self.work = torch.zeros(128, dtype=torch.int64, device=device)

    def partition(self, x, y, v, world):
        n = x.numel()
        counts = (C.c_uint64 * world)()
        perm = torch.empty(n, dtype=torch.int32, device=x.device)
        xo, yo = torch.empty_like(x), torch.empty_like(y)
        vo = torch.empty_like(v) if v is not None else None
        rc = self.lib.smatrix_partition_dev(
            n, x.data_ptr(), y.data_ptr(), v.data_ptr() if v is not None else None, world,
            C.cast(counts, _lib.u64p), self.work.data_ptr(), perm.data_ptr(), xo.data_ptr(), yo.data_ptr(),
            vo.data_ptr() if vo is not None else None, _stream())
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
            C.cast(counts, _lib.u64p), self.work.data_ptr(), perm.data_ptr(), packed.data_ptr(), _stream())
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
    def __init__(self, group=None, shard=None, partitioner=None, device=None):
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
        self.packed = os.environ.get("SMATRIX_SHARD_PACKED", "1") != "0"   # one collective per op batch
        self.host_staged = os.environ.get("SMATRIX_SHARD_HOST_STAGED", "0") == "1"

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
        return _lib.load().smatrix_shard_of(int(x), self.world)

    def apply_dev(self, op, x, y, v, out):
        """x, y, v, out: 1-D int32 tensors on this rank's device (v None for get).
        COLLECTIVE: every rank must call it with the same op (batch sizes may differ)."""
        vv = None if op == OP_GET else v
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

    def route(self, op, x, y, v=None):
        h = self._Routed()
        h.op, h.n, h.x_dev = op, x.numel(), x.device
        comm = self._comm(x)
        vv = None if op == OP_GET else v
        ctx = torch.cuda.stream(comm) if comm is not None else _null_ctx()
        if comm is not None:
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
                h.xr, h.yr, h.vr = self.part.unpack(pr)
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
                h.keep = (x, y, v, xo, yo, vo)
            h.outr = torch.empty(nr, dtype=x.dtype, device=x.device)
            h.ev_routed = comm.record_event() if comm is not None else None
        self.exchanged_ops += nr
        return h

    def apply_routed(self, h):
        if h.ev_routed is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(h.ev_routed)
            for t in (h.xr, h.yr, h.vr, h.outr):
                if t is not None:
                    t.record_stream(cur)                          # allocated under the communication stream
        self.shard.apply(h.op, h.xr, h.yr, h.vr, h.outr)
        h.ev_applied = torch.cuda.current_stream().record_event() if h.ev_routed is not None else None

    def finish(self, h, out):
        comm = self._comm(out)
        ctx = torch.cuda.stream(comm) if comm is not None else _null_ctx()
        with ctx:
            if comm is not None:
                comm.wait_event(h.ev_applied)
                out.record_stream(comm)
            back = torch.empty(h.n, dtype=out.dtype, device=out.device)
            self._a2a(back, h.outr, h.counts, h.rcounts)
            self.part.gather(back, h.perm, out)
            h.ev_done = comm.record_event() if comm is not None else None
            h.keep = h.keep + (back,)
        return h

    def wait(self, h):
        if getattr(h, "ev_done", None) is not None:
            torch.cuda.current_stream().wait_event(h.ev_done)
        h.keep = None

    def close(self):
        self.shard.close()


class _null_ctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
