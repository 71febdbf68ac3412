"""One rank of the C library's router (include/smatrix_shard.h) on the shared-memory test transport: N of these
processes share cuda:0.  Launched by tests/test_gpu_round3.py; writes what it saw to $SMX_OUT/rank<r>_<phase>.npz.
Env: SMX_RANK, SMX_WORLD, SMX_ID (shm object name), SMX_OUT, SMX_FILE (optional shard file prefix), SMX_PHASE
(build | reopen), SMX_STEPS, SMX_N (ops per rank and step)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from libsmatrix_amd import Stream
from libsmatrix_amd.sharded import NativeShardedMatrix

rank, world = int(os.environ["SMX_RANK"]), int(os.environ["SMX_WORLD"])
phase = os.environ.get("SMX_PHASE", "build")
steps, n = int(os.environ.get("SMX_STEPS", "3")), int(os.environ.get("SMX_N", "65536"))
out_dir = os.environ["SMX_OUT"]
prefix = os.environ.get("SMX_FILE")
fname = "%s.shard%d.smx" % (prefix, rank) if prefix else None
ident = os.environ["SMX_ID"].encode().ljust(128, b"\0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
u = lambda a: a.cpu().numpy().view(np.uint32)

sh = NativeShardedMatrix(fname, rank=rank, world=world, unique_id=ident if world > 1 else None)
assert sh.transport == ("shm" if world > 1 else "self"), sh.transport
if os.environ.get("SMX_SKEWED_CUTS") and phase == "build":
    # a placement chosen by the caller: hash ranges of very unequal width (a tenth of the space for every shard but the last)
    from libsmatrix_amd.sharded import Placement
    sh.set_placement(Placement(world, [int((k + 1) * 0.1 * 2 ** 32) for k in range(world - 1)], {}))
first = 0 if phase == "build" else 1000
gen = Stream("zipf", 777 + rank, 30000, 1.1, 1)
save = {}


def batch(s):
    x, y = gen.fill((first + s) * n, n)
    if rank == world - 1 and s == 1:
        x, y = x[: n // 3], y[: n // 3]                       # ranks may submit different n (here: a short batch)
    v = ((x + y) % 4 + 1).astype(np.uint32)
    return x, y, v


if phase == "reopen":
    # everything the build phase wrote must be where the stored placement says: get on the build phase's keys first
    prev = np.load(os.path.join(out_dir, "rank%d_build.npz" % rank))
    for s in range(int(prev["steps"])):
        x, y = prev["x%d" % s], prev["y%d" % s]
        o = torch.empty(x.size, dtype=torch.int32, device=dev)
        sh.apply_dev(0, t(x), t(y), None, o)
        torch.cuda.synchronize()
        save["reget%d" % s] = u(o)

# the benchmark's pipeline through the split phases: the records of step s+1 travel under the kernels of step s
arrs = [tuple(t(a) for a in batch(s)) for s in range(steps)]
host = [batch(s) for s in range(steps)]
torch.cuda.synchronize()
h = sh.route(2, *arrs[0], inputs_ready=True)
for s in range(steps):
    hn = sh.route(2, *arrs[s + 1], inputs_ready=True) if s + 1 < steps else None
    oi = torch.empty(arrs[s][0].numel(), dtype=torch.int32, device=dev); og = torch.empty_like(oi)
    sh.apply_routed(h, then_get=True)
    sh.finish(h, oi, og)
    sh.wait(h)
    torch.cuda.synchronize()
    save.update({"x%d" % s: host[s][0], "y%d" % s: host[s][1], "v%d" % s: host[s][2], "oi%d" % s: u(oi), "og%d" % s: u(og)})
    h = hn
# one blocking call of each kind as well: a set batch (the value is a function of the key: any winner among the ranks'
# duplicates holds the same), then a plain get.  (No decr: a cell that ends at value 0 is dropped by the reference's
# load rule at reopen, quirk Q4, and WHICH chains that cuts depends on the batch-built layout.)
x, y, v = batch(steps)
vs = ((x * 7 + y) % 5 + 1).astype(np.uint32)
od = torch.empty(x.size, dtype=torch.int32, device=dev); og = torch.empty_like(od)
sh.apply_dev(1, t(x), t(y), t(vs), od)
sh.apply_dev(0, t(x), t(y), None, og)
torch.cuda.synchronize()
save.update({"xd": x, "yd": y, "vd": vs, "od": u(od), "ogd": u(og)})

# rowlen / getrow of arbitrary rows (most of them owned by other ranks), absent ids included
rng = np.random.default_rng(5 + rank)
ids = np.concatenate([rng.choice(host[0][0], 3000), rng.integers(1, 1 << 31, 50, dtype=np.uint32)]).astype(np.uint32)
lens = torch.empty(ids.size, dtype=torch.int32, device=dev)
sh.rowlen_dev(t(ids), lens)
torch.cuda.synchronize()
ln = u(lens).astype(np.int64)
off = np.zeros(ids.size + 1, np.int64); np.cumsum(ln + 1, out=off[1:])
pairs = torch.zeros((int(off[-1]), 2), dtype=torch.int32, device=dev)
cnt = torch.empty(ids.size, dtype=torch.int32, device=dev)
sh.getrow_dev(t(ids), torch.from_numpy(off).to(dev), pairs, cnt)
torch.cuda.synchronize()
pl = sh.placement
save.update({"steps": steps, "ids": ids, "lens": u(lens), "off": off, "pairs": u(pairs.reshape(-1)).reshape(-1, 2), "cnt": u(cnt),
             "rows_local": int(sh.local.stats()["rows"]), "ops_applied": sh.exchanged_ops,
             "placed_rows": len(pl.place), "cuts": np.array(pl.cuts if pl.cuts is not None else [], np.uint64)})
np.savez(os.path.join(out_dir, "rank%d_%s.npz" % (rank, phase)), **save)
sh.close()
print("rank %d %s ok" % (rank, phase), flush=True)
