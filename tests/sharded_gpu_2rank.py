"""Run by tests/test_gpu_parity.py::test_sharded_two_ranks_one_gpu under torch.distributed.run (2 ranks, gloo,
BOTH on cuda:0, payload staged through the host: SMATRIX_SHARD_HOST_STAGED=1).  The real HIP partitioner,
the real HIP shards and the pipelined split-phase routing with world_size 2, checked against a single
un-sharded HIP matrix fed with the ops of both ranks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ["SMATRIX_SHARD_HOST_STAGED"] = "1"
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_INCR
from libsmatrix_amd.sharded import ShardedMatrix

B, S = 1 << 18, 4
gen = Stream("zipf", 999 + rank, 100000, 1.1, 1)
xs = torch.empty((S, B), dtype=torch.int32, device=dev); ys = torch.empty_like(xs)
st = torch.cuda.current_stream().cuda_stream
for s in range(S):
    gen.fill_device(s * B, B, xs[s].data_ptr(), ys[s].data_ptr(), st)
ones = torch.ones(B, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
sm, direct = ShardedMatrix(), SparseMatrix()
pending = {}
for s in range(S):
    oi = torch.full((B,), -7, dtype=torch.int32, device=dev); og = torch.full((B,), -7, dtype=torch.int32, device=dev)
    h_i = pending.pop(s, None) or sm.route(OP_INCR, xs[s], ys[s], ones)
    h_g = sm.route(OP_GET, xs[s], ys[s])
    sm.apply_routed(h_i); sm.finish(h_i, oi)
    sm.apply_routed(h_g)                                 # bench.py's order: the get kernel runs under the next route()
    if s + 1 < S:
        pending[s + 1] = sm.route(OP_INCR, xs[s + 1], ys[s + 1], ones)
    sm.finish(h_g, og)
    sm.wait(h_i); sm.wait(h_g)
    torch.cuda.synchronize()
    # the un-sharded matrix sees the step's ops of ALL ranks, then this rank's gets
    allx = [torch.empty(B, dtype=torch.int32) for _ in range(world)]; ally = [torch.empty(B, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(allx, xs[s].cpu()); dist.all_gather(ally, ys[s].cpu())
    for r in range(world):
        ax, ay = allx[r].to(dev), ally[r].to(dev); tmp = torch.empty_like(ax)
        direct.apply_batch_dev(OP_INCR, B, ax.data_ptr(), ay.data_ptr(), ones.data_ptr(), tmp.data_ptr(), st)
    dg = torch.empty(B, dtype=torch.int32, device=dev)
    direct.apply_batch_dev(OP_GET, B, xs[s].data_ptr(), ys[s].data_ptr(), None, dg.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(og, dg), "rank %d step %d: sharded get != un-sharded get" % (rank, s)
    assert int((oi <= og).all()), "an incr return exceeds the value the following get saw"
rows = torch.unique(torch.cat([xs.reshape(-1)]))[:20000]
owners = torch.tensor([sm.owner(int(v)) for v in rows[:2000].tolist()])
mine = rows[:2000][owners == rank]
if mine.numel():
    l = torch.empty(mine.numel(), dtype=torch.int32, device=dev)
    sm.local.rowlen_batch_dev(mine.numel(), mine.contiguous().data_ptr(), l.data_ptr(), st); torch.cuda.synchronize()
    assert int((l > 0).all()), "an owned row is missing from the local shard"
# getrow of arbitrary rows, routed to the owners: the same cells as the un-sharded matrix holds (slot order is the owner's)
qrows = rows[:300].contiguous()
off, prs, cnt = sm.getrow_dev(qrows)
torch.cuda.synchronize()
doff, dpairs, dcnt = direct.getrow_batch(qrows.cpu().numpy().view("uint32"))
assert cnt.cpu().tolist() == dcnt.tolist()
pc, oc = prs.cpu().numpy().view("uint32"), off.cpu().tolist()
for i in range(qrows.numel()):
    a = sorted(map(tuple, pc[oc[i]: oc[i] + int(cnt[i])].tolist()))
    b = sorted(map(tuple, dpairs[int(doff[i]): int(doff[i]) + int(dcnt[i])].tolist()))
    assert a == b, "getrow of row %d differs" % int(qrows[i])
# skew-aware placement was planned from the first batch: hot rows placed one by one, unequal hash ranges
assert sm.placement.place and sm.placement.cuts is not None
share = torch.tensor([float(sm.exchanged_ops)]); tot_ops = share.clone(); dist.all_reduce(tot_ops)
assert abs(float(share) * world / float(tot_ops) - 1.0) < 0.15, "shard load %.3f of the mean" % (float(share) * world / float(tot_ops))
tot = torch.tensor([sm.local.stats()["rows"]]); dist.all_reduce(tot)
assert int(tot) == direct.stats()["rows"], (int(tot), direct.stats()["rows"])
# file-backed shards: close (files + stored placement), reopen, same answers, further writes land on the same owners
import tempfile
from libsmatrix_amd.sharded import HipShard
d = [tempfile.mkdtemp(prefix="smx2r_") if rank == 0 else None]
dist.broadcast_object_list(d, src=0)
path = os.path.join(d[0], "shard%d.smx" % rank)
fm = ShardedMatrix(shard=HipShard(path))
oi = torch.empty(B, dtype=torch.int32, device=dev)
fm.apply_dev(OP_INCR, xs[0], ys[0], ones, oi)
want = torch.empty(B, dtype=torch.int32, device=dev)
fm.apply_dev(OP_GET, xs[0], ys[0], None, want)
pl_text = fm.placement.to_json()
fm.close()
assert os.path.exists(path + ".placement")
fm = ShardedMatrix(shard=HipShard(path))
got = torch.empty(B, dtype=torch.int32, device=dev)
fm.apply_dev(OP_GET, xs[0], ys[0], None, got)
assert fm.placement.to_json() == pl_text, "the stored placement was not taken over"
assert torch.equal(got, want), "values differ after close / reopen of the shard files"
fm.apply_dev(OP_INCR, xs[0], ys[0], ones, oi)
fm.apply_dev(OP_GET, xs[0], ys[0], None, got)
torch.cuda.synchronize()
assert torch.equal(got, want * 2), "writes after the reopen did not reach the rows' owners"
fm.close()
# files written with EQUAL ranges and no stored placement (e.g. by an earlier version): reopened from where the rows are
path2 = os.path.join(d[0], "eq%d.smx" % rank)
os.environ["SMATRIX_SHARD_PLACE"] = "0"
em = ShardedMatrix(shard=HipShard(path2))
em.apply_dev(OP_INCR, xs[1], ys[1], ones, oi)
em.apply_dev(OP_GET, xs[1], ys[1], None, want)
em.close()
os.remove(path2 + ".placement")
os.environ["SMATRIX_SHARD_PLACE"] = "1"
em = ShardedMatrix(shard=HipShard(path2))
em.apply_dev(OP_GET, xs[1], ys[1], None, got)
assert em.placement.cuts is None and not em.placement.place and torch.equal(got, want)
em.close()
dist.barrier()
if rank == 0:
    import shutil
    shutil.rmtree(d[0], ignore_errors=True)
    print("SHARDED_2RANK_OK rows=%d" % int(tot))
sm.close(); direct.close(); dist.destroy_process_group()
