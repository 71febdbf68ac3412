#!/usr/bin/env python3
"""GPU probe: time the phases of ShardedMatrix.apply_dev with one rank (packed vs separate arrays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
from libsmatrix_amd import Stream, OP_INCR, OP_GET
from libsmatrix_amd.sharded import HipPartitioner
B = 1 << 24
gen = Stream("zipf", 12345, 1000000, 1.1, 1)
x = torch.empty(B, dtype=torch.int32, device=dev); y = torch.empty_like(x); v = torch.ones_like(x)
gen.fill_device(0, B, x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream)
part = HipPartitioner(dev)
def t(label, fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    print("%-40s %8.3f ms" % (label, (time.perf_counter() - t0) / reps * 1e3)); return r
counts, perm, xo, yo, vo = t("partition (3 arrays)", lambda: part.partition(x, y, v, 1))
# the same with 8 shards and a planned placement (256 hot rows + unequal ranges)
from libsmatrix_amd.sharded import plan_placement
ux, cnt = torch.unique(x, return_counts=True); top = torch.topk(cnt, 256).indices
pl = plan_placement({int(a) & 0xFFFFFFFF: int(c) for a, c in zip(ux[top].tolist(), cnt[top].tolist())}, B, 8, 256)
c8 = t("partition packed, 8 shards, equal ranges", lambda: part.partition_packed(x, y, v, 8))[0]
part.set_placement(pl)
c8p = t("partition packed, 8 shards, planned", lambda: part.partition_packed(x, y, v, 8))[0]
part.set_placement(None)
print("ops per shard / mean: equal ranges %s  planned %s" % ([round(c * 8 / B, 2) for c in c8], [round(c * 8 / B, 2) for c in c8p]))
counts, perm, po = t("partition packed [n,3]", lambda: part.partition_packed(x, y, v, 1))
pr = torch.empty_like(po); xr = torch.empty_like(xo)
t("all_to_all_single packed [n,3]", lambda: dist.all_to_all_single(pr, po, counts, counts))
t("all_to_all_single one array", lambda: dist.all_to_all_single(xr, xo, counts, counts))
t("unpack", lambda: part.unpack(pr))
out = torch.empty_like(x)
t("gather", lambda: part.gather(xr, perm, out))
t("torch.empty x4", lambda: [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)])
dist.destroy_process_group()
