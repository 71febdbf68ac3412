"""Run by tests/test_gpu_parity.py::test_sharded_pipeline_matches_direct (one rank, RCCL): the pipelined
split-phase ShardedMatrix (separate communication stream) must give exactly the direct path's results."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29588", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_INCR
from libsmatrix_amd.sharded import ShardedMatrix

B, S = 1 << 20, 6
gen = Stream("zipf", 4242, 200000, 1.1, 1)
xs = torch.empty((S, B), dtype=torch.int32, device=dev); ys = torch.empty_like(xs)
st = torch.cuda.current_stream().cuda_stream
for s in range(S):
    gen.fill_device(s * B, B, xs[s].data_ptr(), ys[s].data_ptr(), st)
ones = torch.ones(B, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
direct, sm = SparseMatrix(), ShardedMatrix()
pending = {}
for s in range(S):
    di = torch.empty(B, dtype=torch.int32, device=dev); dg = torch.empty_like(di)
    direct.apply_batch_dev(OP_INCR, B, xs[s].data_ptr(), ys[s].data_ptr(), ones.data_ptr(), di.data_ptr(), st)
    direct.apply_batch_dev(OP_GET, B, xs[s].data_ptr(), ys[s].data_ptr(), None, dg.data_ptr(), st)
    oi = torch.full((B,), -7, dtype=torch.int32, device=dev); og = torch.full((B,), -7, dtype=torch.int32, device=dev)
    h_i = pending.pop(s, None) or sm.route(OP_INCR, xs[s], ys[s], ones)
    h_g = sm.route(OP_GET, xs[s], ys[s])
    sm.apply_routed(h_i); sm.finish(h_i, oi)
    sm.apply_routed(h_g)
    if s + 1 < S:
        pending[s + 1] = sm.route(OP_INCR, xs[s + 1], ys[s + 1], ones)
    sm.finish(h_g, og)
    sm.wait(h_i); sm.wait(h_g)
    torch.cuda.synchronize()
    assert torch.equal(og, dg), "get mismatch at step %d" % s
    # incr returns: same per-key multiset (order inside a key is a serialisation choice)
    k = (xs[s].long() & 0xFFFFFFFF) << 32 | (ys[s].long() & 0xFFFFFFFF)
    a = torch.stack([k, oi.long() & 0xFFFFFFFF], 1); b = torch.stack([k, di.long() & 0xFFFFFFFF], 1)
    a = a[torch.argsort(a[:, 1], stable=True)]; a = a[torch.argsort(a[:, 0], stable=True)]
    b = b[torch.argsort(b[:, 1], stable=True)]; b = b[torch.argsort(b[:, 0], stable=True)]
    assert torch.equal(a, b), "incr returns mismatch at step %d" % s
assert direct.stats()["rows"] == sm.local.stats()["rows"]
q = xs[0][:100000].contiguous(); l1 = torch.empty_like(q); l2 = torch.empty_like(q)
sm.rowlen_dev(q, l1); direct.rowlen_batch_dev(q.numel(), q.data_ptr(), l2.data_ptr(), st); torch.cuda.synchronize()
assert torch.equal(l1, l2)
print("SHARDED_PIPELINE_OK")
sm.close(); direct.close(); dist.destroy_process_group()
