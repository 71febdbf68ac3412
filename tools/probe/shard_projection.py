#!/usr/bin/env python3
"""A PROJECTION of the 8-shard step of config 4 from measurements on ONE GPU -- no 8-GPU node has ever run this code.

Config 4's stream (8 M x 8 M ids, Zipf(1.1)^2, scrambled, seed 12345 + rank: bench.py N_IDS_CONFIG4) is generated for `world` ranks;
the placement is planned from the ranks' first batches exactly as ShardedMatrix does (plan_placement: the 256 hottest rows one by
one, the tail in hash ranges of unequal width); every step's batches are partitioned by owner with the router's own partition
kernel; then, for each shard IN TURN, a fresh matrix receives the records that shard would receive -- `world` slices per step, one
from each rank, applied as ONE routed batch of packed {x, y, v} records like apply_routed does -- for `steps` steps: incr batch +
get batch.  Reported per shard: ops received per step, incr / get kernel ms (HIP events of the round-0 kernel), step ms (wall,
device-resident), and the projection

    speed-up(N) = N x direct step / (slowest shard's step + route + finish overhead)

where the direct step is the same stream un-routed on one matrix (one rank's batch per step) and the overhead is what one rank
pays around its shard's kernels per step: partition + result gather on the device, measured here; the exchange itself (xGMI) is NOT
measured and NOT in the figure (DESIGN.md 6 prices it at 0.2 ms per step, hidden under the previous batch's kernels).

  python tools/probe/shard_projection.py [steps=12] [world=8] [batch_lg=24] [n_ids=8000000]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_INCR
from libsmatrix_amd.sharded import HipPartitioner, plan_placement

def project(steps=12, world=8, blg=24, n_ids=8000000, out=print):
    """-> {"direct": {...}, "shards": [...], "overhead_ms": .., "max_over_mean": .., "ops_max_over_mean": .., "speedup": ..}"""
    B = 1 << blg
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream().cuda_stream
    gens = [Stream("zipf", 12345 + r, n_ids, 1.1, 1) for r in range(world)]
    part = HipPartitioner(dev)
    x = torch.empty(B, dtype=torch.int32, device=dev); y = torch.empty_like(x); ones = torch.ones_like(x)


    def batch(r, s):
        gens[r].fill_device(s * B, B, x.data_ptr(), y.data_ptr(), stream)
        return x, y


    # ---- the placement, from the ranks' first batches (ShardedMatrix.apply_dev: the first write batch is the sample)
    cnt = {}
    for r in range(world):
        bx, _ = batch(r, 0)
        ux, c = torch.unique(bx, return_counts=True)
        top = torch.topk(c, min(1024, c.numel())).indices
        for a, k in zip(ux[top].tolist(), c[top].tolist()):
            cnt[a & 0xFFFFFFFF] = cnt.get(a & 0xFFFFFFFF, 0) + k
    hot = dict(sorted(cnt.items(), key=lambda kv: -kv[1])[:256])
    pl = plan_placement(hot, world * B, world, 256)
    part.set_placement(pl)
    share1 = max(hot.values()) / float(world * B)
    hot_row = max(hot, key=hot.get)
    out("# config 4 stream: %d ranks x 2^%d ops per step over %d x %d ids; hottest row %.2f %% of all ops (fair share %.2f %%) -> shard %d" % (
        world, blg, n_ids, n_ids, 100 * share1, 100.0 / world, pl.owner(hot_row)))

    # ---- overhead of one rank per step around its shard's kernels: partition of its batch, gather of its results
    bx, by = batch(0, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        counts, perm, packed = part.partition_packed(bx, by, ones, world)
    torch.cuda.synchronize(); t_part = (time.perf_counter() - t0) / 5 * 1e3
    res = torch.empty(B, dtype=torch.int32, device=dev); outb = torch.empty_like(res)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        part.gather(res, perm, outb)
    torch.cuda.synchronize(); t_gather = (time.perf_counter() - t0) / 5 * 1e3
    overhead = 2 * t_part + 2 * t_gather          # (incr batch and get batch each: route + finish)
    out("# one rank's routing work per step (partition + gather, incr and get): %.3f ms" % overhead)


    def run(apply_step, label):
        m = SparseMatrix()
        m.reserve(24 << 30)
        m.profile(True)
        ts, nops = [], []
        for s in range(steps):
            recs = apply_step(s)
            n = recs.shape[0]
            out = torch.empty(n, dtype=torch.int32, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.apply_packed_dev(OP_INCR, n, recs.data_ptr(), 3, out.data_ptr(), stream)
            m.apply_packed_dev(OP_GET, n, recs.data_ptr(), 3, out.data_ptr(), stream)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); nops.append(n)
            del recs, out
        st = m.stats()
        m.close()
        warm = min(2, steps - 1)
        step_ms = sum(ts[warm:]) / len(ts[warm:])
        return {"label": label, "ops": sum(nops[warm:]) / len(nops[warm:]), "step_ms": step_ms,
                "incr_ms": st["kernel_ms_incr"] / max(st["kernel_launches_incr"], 1), "get_ms": st["kernel_ms_get"] / max(st["kernel_launches_get"], 1),
                "rows": int(st["rows"])}


    def direct_step(s):
        bx, by = batch(0, s)
        return torch.stack([bx, by, ones], dim=1).contiguous()


    def shard_step(k):
        def f(s):
            pieces = []
            for r in range(world):
                bx, by = batch(r, s)
                counts, perm, packed = part.partition_packed(bx, by, ones, world)
                lo = sum(counts[:k])
                pieces.append(packed[lo:lo + counts[k]].clone())
            return torch.cat(pieces, dim=0).contiguous()
        return f


    d = run(direct_step, "direct (one rank's batch, un-routed)")
    out("%-44s ops/step %10.0f  step %7.3f ms  incr kernel %6.3f  get kernel %6.3f  rows %d" % (d["label"], d["ops"], d["step_ms"], d["incr_ms"], d["get_ms"], d["rows"]))
    rs = []
    for k in range(world):
        r = run(shard_step(k), "shard %d%s" % (k, "  (holds the hottest row)" if pl.owner(hot_row) == k else ""))
        rs.append(r)
        out("%-44s ops/step %10.0f  step %7.3f ms  incr kernel %6.3f  get kernel %6.3f  rows %d" % (r["label"], r["ops"], r["step_ms"], r["incr_ms"], r["get_ms"], r["rows"]), flush=True)
    mx = max(r["step_ms"] for r in rs); mean = sum(r["step_ms"] for r in rs) / world
    out("per-shard step ms: max %.3f  mean %.3f  max/mean %.3f ; ops per shard max/mean %.3f" % (mx, mean, mx / mean, max(r["ops"] for r in rs) / (sum(r["ops"] for r in rs) / world)))
    out("PROJECTION (not a measurement of %d GPUs): speed-up = %d x %.3f / (%.3f + %.3f) = %.2fx  (exchange over xGMI not included)" % (
        world, world, d["step_ms"], mx, overhead, world * d["step_ms"] / (mx + overhead)))
    for g in gens:
        g.close()
    return {"direct": d, "shards": rs, "overhead_ms": overhead, "max_over_mean": mx / mean,
            "ops_max_over_mean": max(r["ops"] for r in rs) / (sum(r["ops"] for r in rs) / world), "speedup": world * d["step_ms"] / (mx + overhead), "hot_share": share1}



if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:5]]
    project(*a)
