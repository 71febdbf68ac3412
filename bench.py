#!/usr/bin/env python3
"""bench.py -- mixed incr+get throughput of the HIP path on BASELINE.json's config 2.

Workload (N=1): the 1M-row x 1M-col Zipf(1.1) stream with scrambled ids, seed 12345
(SURVEY.md 8d), cut into batches of 2^24 ops.  One STEP = one pass of the hot path
over one batch: smatrix_incr on the batch, then smatrix_get on the same keys (the
order of the reference benchmark: incr test, then get test,
src/smatrix_benchmark.c:226-230) = 2 * 2^24 ops.  Batches are generated on the GPU
before the timed region, so inputs are resident in HBM when timing starts.
warmup + steps = 24 batches cover the whole 4e8-op stream (100.4M nnz in 1M rows).

N>1: one process per GPU (torch.distributed, backend nccl = RCCL); every rank draws its
own slice of the stream, ops are routed to the row's owner shard with all_to_all
(libsmatrix_amd/sharded.py) and results routed back; the exchange of get(s) overlaps the
incr kernels of step s and the exchange of incr(s+1) overlaps get(s) (split-phase API on a
separate communication stream).  scaling = weak.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH_LG = 24
N_IDS = 1000000
ZIPF_S = 1.1
SEED = 12345
# algorithmic bytes per op, SURVEY.md 8(d): get = 8 in + 12 cell + 4 out; incr (with return) = 12 + 12 + 4 + 4
BYTES_GET, BYTES_INCR = 24, 32
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(sample_ops, torch, dev):
    """The reference's CPU path timed on this host, one thread, on the first `sample_ops` ops of the
    same stream (generated on the GPU, copied to the host): incr batch then get batch, in chunks of
    2^24 like the GPU run.  Uses the real reference (oracle/_ref, kind 'reference') when the prebuilt
    library is present, else the port.  ~10-30 s of CPU work."""
    import numpy as np
    from libsmatrix_amd import Stream
    from oracle import oracle as O
    gen = Stream("zipf", SEED, N_IDS, ZIPF_S, 1)
    xd = torch.empty(sample_ops, dtype=torch.int32, device=dev)
    yd = torch.empty_like(xd)
    gen.fill_device(0, sample_ops, xd.data_ptr(), yd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    x, y = xd.cpu().numpy().view(np.uint32), yd.cpu().numpy().view(np.uint32)
    del xd, yd
    kind = "reference" if O.have_reference() else "port"
    m = O.Reference() if kind == "reference" else O.Oracle()
    chunk = 1 << 24
    t_incr = t_get = 0.0
    for a in range(0, sample_ops, chunk):
        xs, ys = x[a:a + chunk], y[a:a + chunk]
        ones = np.ones(xs.size, np.uint32)
        t0 = time.perf_counter()
        m.apply(O.OP_INCR, xs, ys, ones)
        t1 = time.perf_counter()
        m.apply(O.OP_GET, xs, ys)
        t2 = time.perf_counter()
        t_incr += t1 - t0
        t_get += t2 - t1
    rows = m.num_rows()
    m.close()
    res = {
        "value": round(2 * sample_ops / (t_incr + t_get) / 1e6, 3), "unit": "Mops/s", "cores": 1, "kind": kind,
        "sample": "first %d ops of the same Zipf stream in batches of 2^24 (incr batch then get batch): "
                  "incr %.2fs + get %.2fs, 1 thread, %d rows at the end; host has %d cores"
                  % (sample_ops, t_incr, t_get, rows, os.cpu_count()),
    }
    if kind == "reference":
        # the reference is thread-safe (per-row spin RW locks, src/smatrix.c:843-889): the same ops split in
        # contiguous slices over 8 threads, thread start inside the timed region as in smatrix_benchmark.c:109-122
        import threading
        T, n8 = 8, min(sample_ops, 1 << 25)
        m = O.Reference()
        t8 = 0.0
        for a in range(0, n8, chunk):
            xs, ys = x[a:a + chunk], y[a:a + chunk]
            ones = np.ones(xs.size, np.uint32)
            per = (xs.size + T - 1) // T
            for op in (O.OP_INCR, O.OP_GET):
                th = [threading.Thread(target=m.apply, args=(op, xs[i * per:(i + 1) * per], ys[i * per:(i + 1) * per],
                                                             ones[i * per:(i + 1) * per])) for i in range(T)]
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                t8 += time.perf_counter() - t0
        m.close()
        res["threads_8"] = {"value": round(2 * n8 / t8 / 1e6, 3), "unit": "Mops/s", "cores": T,
                            "sample": "first %d ops, each batch split over 8 threads: %.2fs" % (n8, t8)}
    return res


def random_access_roofline(torch, dev, gib=4, touches=1 << 27):
    """R_rand of this chip, measured now (not a datasheet number): pseudo-random
    8-byte touches over a table-sized buffer, one kernel per mode, torch events on the stream
    the probe kernel is launched on."""
    from libsmatrix_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(gib << 27, dtype=torch.int64, device=dev)       # gib GiB
    sink = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    out = {"buffer_gib": gib, "touches": touches}
    for mode, name in ((0, "read8"), (3, "chain2"), (1, "atomic_ret"), (2, "atomic_noret")):
        best = None
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.smx_probe_random_dev(buf.data_ptr(), buf.numel() * 8, touches, mode, 99 + rep, sink.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        out[name + "_gtouch_per_s"] = touches / (best * 1e-3) / 1e9
    del buf
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=22)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-lg", type=int, default=BATCH_LG)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-sample-lg", type=int, default=26)
    ap.add_argument("--no-profile", action="store_true", help="do not time kernels with HIP events (A/B of the overhead)")
    ap.add_argument("--no-overlap", action="store_true", help="sharded path: one blocking apply_dev per op batch")
    ap.add_argument("--no-comm-thread", action="store_true", help="sharded path: issue the exchange from the main thread")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for test rigs)")
    ap.add_argument("--single-device", action="store_true",
                    help="test rig: every rank uses cuda:0 and the exchange is staged through the host (gloo)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="route through ShardedMatrix even with one rank (exercises the exchange path)")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with python -m torch.distributed.run --nproc-per-node %d" % args.gpus)
    if args.single_device:
        local = 0
        os.environ["SMATRIX_SHARD_HOST_STAGED"] = "1"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:          # single process, --force-sharded
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from libsmatrix_amd import SparseMatrix, Stream, OP_GET, OP_INCR
    B = 1 << args.batch_lg
    total_steps = args.warmup + args.steps
    gen = Stream("zipf", SEED + (rank if world > 1 else 0), N_IDS, ZIPF_S, 1)

    # inputs live in HBM before the timed region; beyond 32 batches (4 GB) the pre-generated batches are
    # reused cyclically (the stream then repeats: later passes over a batch are all hits)
    ring = min(total_steps, 32)
    xs_all = torch.empty((ring, B), dtype=torch.int32, device=dev)
    ys_all = torch.empty((ring, B), dtype=torch.int32, device=dev)

    class _Ring:
        def __init__(self, t):
            self.t = t

        def __getitem__(self, s):
            return self.t[s % ring]
    xs, ys = _Ring(xs_all), _Ring(ys_all)
    ones = torch.ones(B, dtype=torch.int32, device=dev)
    out_i = torch.empty(B, dtype=torch.int32, device=dev)
    out_g = torch.empty(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for s in range(ring):
        gen.fill_device(s * B, B, xs[s].data_ptr(), ys[s].data_ptr(), stream)
    torch.cuda.synchronize()

    if sharded:
        from libsmatrix_amd.sharded import ShardedMatrix
        m = ShardedMatrix()
    else:
        m = SparseMatrix()

    pending = {}       # sharded: batches whose incr records are already travelling (routed ahead)

    # Sharded + overlap: everything that talks to the other ranks (partition, the collectives, gather) runs on
    # ONE helper thread, in a fixed order, so that the host-side waits of the exchange (split sizes must reach
    # the host before a collective can be issued) run under the host-driven rounds of the op kernels instead of
    # in front of them.  The library calls release the GIL; the op kernels stay on the main thread.
    comm = None
    if sharded and not args.no_overlap and not args.no_comm_thread:
        from concurrent.futures import ThreadPoolExecutor
        comm = ThreadPoolExecutor(max_workers=1, initializer=lambda: torch.cuda.set_device(local))

    READY = True       # the batches were generated before the timed region: route() need not wait for the compute stream
    lagging = []       # handles whose results the compute stream has not been told to wait for yet

    def step_threaded(s):
        f_i = pending.pop(s, None) or comm.submit(m.route, OP_INCR, xs[s], ys[s], ones, READY)
        f_g = comm.submit(m.route, OP_GET, xs[s], ys[s], None, READY)           # travels under the incr kernels of s
        h_i = f_i.result()
        m.apply_routed(h_i)
        f_fi = comm.submit(m.finish, h_i, out_i)
        h_g = f_g.result()
        m.apply_routed(h_g)
        if s + 1 < total_steps and s + 1 != args.warmup:           # never across the timing fence
            pending[s + 1] = comm.submit(m.route, OP_INCR, xs[s + 1], ys[s + 1], ones, READY)
        f_fg = comm.submit(m.finish, h_g, out_g)
        f_fi.result(); f_fg.result()
        # results are consumed after the timed region only: the compute stream is made to wait for the result
        # exchange of the PREVIOUS step (long finished), not of this one (same box: 3.93 -> 3.88 ms, one rank)
        for h in lagging:
            m.wait(h)
        lagging[:] = [h_i, h_g]

    def step(s):
        if comm is not None:
            return step_threaded(s)
        if sharded and not args.no_overlap:
            # exchange of get(s) overlaps the incr kernels of s; exchange of incr(s+1) overlaps get(s)
            h_i = pending.pop(s, None) or m.route(OP_INCR, xs[s], ys[s], ones, READY)
            h_g = m.route(OP_GET, xs[s], ys[s], None, READY)
            m.apply_routed(h_i)                                   # host-driven rounds; get(s) records travel meanwhile
            m.finish(h_i, out_i)
            m.apply_routed(h_g)                                   # one async launch: runs while the host sits in the next route()
            if s + 1 < total_steps and s + 1 != args.warmup:      # never across the timing fence
                pending[s + 1] = m.route(OP_INCR, xs[s + 1], ys[s + 1], ones, READY)
            m.finish(h_g, out_g)
            m.wait(h_i); m.wait(h_g)
        elif sharded:
            m.apply_dev(OP_INCR, xs[s], ys[s], ones, out_i)
            m.apply_dev(OP_GET, xs[s], ys[s], None, out_g)
        else:
            m.apply_batch_dev(OP_INCR, B, xs[s].data_ptr(), ys[s].data_ptr(), ones.data_ptr(),
                              out_i.data_ptr(), stream)
            m.apply_batch_dev(OP_GET, B, xs[s].data_ptr(), ys[s].data_ptr(), None, out_g.data_ptr(), stream)

    def fence():
        for h in lagging:
            m.wait(h)
        del lagging[:]
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        step(s)
    local_m = m.local if sharded else m
    local_m.profile(not args.no_profile)          # HIP events around the op kernels, on the stream they run on
    fence()
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        step(s)
    fence()
    dt = time.perf_counter() - t0
    st = local_m.stats()
    local_m.profile(False)
    shard_info = None
    if sharded:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # how evenly the placement spread the ops (ops each shard applied over the whole run / mean)
        loads = [None] * world
        dist.all_gather_object(loads, int(m.exchanged_ops))
        mean = max(sum(loads) / world, 1)
        shard_info = {"rows_placed_by_load": len(m.placement.place),
                      "hash_range_widths": ([round((b - a) / 2.0 ** 32, 4) for a, b in
                                             zip([0] + m.placement.cuts, m.placement.cuts + [1 << 32])]
                                            if m.placement.cuts is not None else "equal"),
                      "ops_applied_over_mean": [round(v / mean, 3) for v in loads]}

    # informative extra (not `value`): the same step on the finished table -- every incr is a hit
    steady = None
    if not sharded:
        fence()
        t1 = time.perf_counter()
        for s in range(total_steps - 4, total_steps):
            step(s)
        fence()
        steady = {"steps": 4, "ms_per_step": (time.perf_counter() - t1) / 4 * 1e3,
                  "Mops_per_s": 2 * B * 4 / (time.perf_counter() - t1) / 1e6,
                  "note": "last 4 batches replayed on the 100.4M-nnz table: no inserts, no growth"}

    # parity spot-check inside the bench: the last get batch must equal the last incr returns' per-key max
    ok = bool((out_g >= 1).all().item())

    total_ops = 2 * B * args.steps * world
    res = {
        "metric": "mixed incr+get ops/s", "value": total_ops / dt / 1e6, "unit": "Mops/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "config-2: Zipf(1.1) x Zipf(1.1) over 1M x 1M scrambled ids, seed 12345, "
                               "batches of 2^%d ops, step = incr batch + get batch on the same keys; "
                               "table grows from %d to %d batches of the 4e8-op stream during the timed steps"
                               % (args.batch_lg, args.warmup, total_steps),
                   "batch_ops": B, "distinct_batches": ring, "parallelism": "row-hash shards x%d" % world if sharded else "single GPU"},
        "sanity_all_gets_positive": ok,
    }
    if shard_info:
        res["config"]["placement"] = shard_info
    if rank == 0:
        ki = st["kernel_ms_incr"] / max(st["kernel_launches_incr"], 1)
        kg = st["kernel_ms_get"] / max(st["kernel_launches_get"], 1)
        # per-launch units: rank 0's local launches (N=1: exactly B ops per launch)
        ops_i = st["kernel_ops_incr"] / max(st["kernel_launches_incr"], 1)
        ops_g = st["kernel_ops_get"] / max(st["kernel_launches_get"], 1)
        ach_i = ops_i * BYTES_INCR / (ki * 1e-3) / 1e9 if ki else 0.0
        ach_g = ops_g * BYTES_GET / (kg * 1e-3) / 1e9 if kg else 0.0
        res["roofline"] = {"bound": "hbm", "kernel": "k_apply_agg<INCR> (round 0)", "achieved": ach_i,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_i / HBM_PEAK_GBS, "traffic": None,
                           "avg_launch_ms": ki, "ops_per_launch": ops_i, "bytes_per_op": BYTES_INCR,
                           "gops_per_s": ops_i / (ki * 1e-3) / 1e9 if ki else 0.0}
        res["roofline_get"] = {"bound": "hbm", "kernel": "k_apply<GET>", "achieved": ach_g, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": ach_g / HBM_PEAK_GBS, "traffic": None,
                               "avg_launch_ms": kg, "ops_per_launch": ops_g, "bytes_per_op": BYTES_GET,
                               "gops_per_s": ops_g / (kg * 1e-3) / 1e9 if kg else 0.0}
        # HBM-side bytes per launch from the committed PMC passes of this same command (rocprofv3
        # cannot be run from inside the measured process); null when the profile is absent
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
            if world == 1 and args.batch_lg == BATCH_LG:
                res["roofline"]["traffic"] = pmc["k_apply_agg_incr"]["bytes_per_launch"]
                res["roofline_get"]["traffic"] = pmc["k_apply_get"]["bytes_per_launch"]
                res["roofline"]["traffic_source"] = res["roofline_get"]["traffic_source"] = "profiles/r01_pmc_summary.txt"
        except Exception:
            pass
        if steady:
            res["steady_state_all_hits"] = steady
        res["table"] = {k: st[k] for k in ("rows", "dir_slots", "arena_units", "arena_mapped", "batches",
                                           "rounds", "deferred_ops", "rows_grown", "dir_grown")}
        if world == 1:
            ra = random_access_roofline(torch, dev)
            g_get = res["roofline_get"]["gops_per_s"]; g_inc = res["roofline"]["gops_per_s"]
            ra.update({
                "get_gops_per_s": g_get, "incr_gops_per_s": g_inc,
                # a get is a directory touch + a dependent cell touch; an incr adds one returning atomic
                "get_frac_of_chain2": g_get / ra["chain2_gtouch_per_s"],
                "get_frac_of_read8": g_get / ra["read8_gtouch_per_s"],
                "incr_frac_of_atomic_ret": g_inc / ra["atomic_ret_gtouch_per_s"],
                "note": "ceilings are uniform-random over the buffer; the Zipf stream re-touches hot lines "
                        "in L2/Infinity Cache, so fractions above 1 are cache assistance, not an error",
            })
            res["random_access"] = ra
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(1 << args.cpu_sample_lg, torch, dev)
        print(json.dumps(res))
    if comm is not None:
        comm.shutdown()
    m.close()
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
