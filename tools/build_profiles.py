#!/usr/bin/env python3
"""Turns the raw outputs of tools/refresh_profiles.sh (gpurun_out/prof_*) into the files committed under profiles/:
r01_bench.json, r01_rocprof_kernel_stats.txt, r01_pmc_summary.txt, r01_pmc.json."""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
raw = open(os.path.join(G, "prof_pmc_raw.txt")).read()
AGG, GET = "smx::k_apply_agg<2, 1u>", "smx::k_apply<0>"


def val(kern, grid, ctr):
    m = re.search(r"%s\s+grid=%s\s+%s\s+dispatches=\s*(\d+) mean=\s*([\d.]+)" % (re.escape(kern), grid, ctr), raw)
    return float(m.group(2))


agg = {k: val(AGG, "8388608", c) for k, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("hit", "TCC_HIT_sum"),
                                                ("miss", "TCC_MISS_sum"), ("atom", "TCC_EA0_ATOMIC_sum"))}
get = {k: val(GET, "16777216", c) for k, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("hit", "TCC_HIT_sum"),
                                                 ("miss", "TCC_MISS_sum"))}
p0 = val("smx::k_probe_random<0>", "2097152", "FETCH_SIZE")
b = json.load(open(os.path.join(G, "prof_bench.json")))
ra = b["random_access"]; R, A = ra["read8_gtouch_per_s"], ra["atomic_ret_gtouch_per_s"]
N = 1 << 24
bytes_agg = (agg["fetch"] + agg["write"]) * 1024; bytes_get = (get["fetch"] + get["write"]) * 1024
rd_agg = agg["miss"] - agg["atom"]
t_agg = rd_agg / (R * 1e9) + agg["atom"] / (A * 1e9); t_get = get["miss"] / (R * 1e9)
ki, kg = b["roofline"]["avg_launch_ms"], b["roofline_get"]["avg_launch_ms"]
hdr = """# rocprofv3 --kernel-trace --pmc <counter> --output-format csv -- python3 bench.py --no-cpu   (three separate passes: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum; tools/refresh_profiles.sh + tools/build_profiles.py)
# MI355X, round 1 (final code of the round), config 2 = 24 batches of 2^24 ops (+ 4 replayed batches of the steady-state extra).  FETCH_SIZE/WRITE_SIZE are KiB per dispatch, means over the dispatches listed.
# Calibration in our own access pattern, same runs: k_probe_random<0> = 2^27 random 8-byte loads over 4 GiB -> %.0f KiB = %.1f B per touch
#   (a 64 B line per touch; no 1/2 correction for this shape); k_probe_random<1>/<2> = 2^27 scattered 32-bit atomics -> WRITE_SIZE 32 B and TCC_EA0_ATOMIC 1.0 per atomic.
#   (the guide's gfx950 correction -- FETCH_SIZE tallies a 128-B coalesced streaming request at 64 B -- applies only to the streamed op arrays,
#    12 B/op in for incr and 8 B/op for get: at most +6 / +4 B/op on the figures below; the random 8-16 B touches are counted in full, as calibrated)
# HBM-side traffic per launch of the op kernels (full-batch dispatches, 2^24 ops each):
#   k_apply_agg<INCR>: fetch %.1f KiB + write %.1f KiB = %.3e B = %.0f B/op (algorithmic 32 B/op); %.1f M memory-side atomics; L2 hit %.0f %%
#   k_apply<GET>     : fetch %.1f KiB + write %.1f KiB = %.3e B = %.0f B/op (algorithmic 24 B/op); L2 hit %.0f %%
#   (each op touches two 64-byte lines -- directory slot, cell -- of which ~half are served by L2; the 8-byte cell costs a 64-byte line)
# The memory-side bound of these transaction counts, priced with the random-touch rates measured in the bench run of the same box
# (read8 %.1f G/s, returning atomic %.1f G/s, uniform over 4 GiB).  TCC_MISS counts an atomic as a miss (probe<1>: 1.0 per atomic),
# so the read misses of the incr kernel are TCC_MISS - TCC_EA0_ATOMIC; they include the streamed op arrays (12 B/op in = 3.1 M lines).
#   k_apply_agg<INCR>: %.1f M read misses / %.1f G/s + %.1f M atomics / %.1f G/s = %.3f ms;  measured %.3f ms per launch (HIP events, growing table;
#                      1.30 ms when every key is a hit) -> %.0f %% of that bound.  The atomics are 80 %% of it: one returning atomic per distinct key of a
#                      2048-op tile (+ a ticket and a claim per new cell) -- the LDS fold removes the duplicates inside a tile, not across tiles.
#   k_apply<GET>     : %.1f M misses / %.1f G/s = %.3f ms;  measured %.3f ms -> %.0f %%
#   i.e. the kernels run within 15-30 %% of the chip's random-transaction rates for what they touch; the rest of the gap to the byte roofline is the COUNT
#   (two lines per op, 64 B moved per 8-16 B used).
""" % (p0, p0 * 1024 / (1 << 27), agg["fetch"], agg["write"], bytes_agg, bytes_agg / N, agg["atom"] / 1e6,
       100 * agg["hit"] / (agg["hit"] + agg["miss"]), get["fetch"], get["write"], bytes_get, bytes_get / N,
       100 * get["hit"] / (get["hit"] + get["miss"]), R, A, rd_agg / 1e6, R, agg["atom"] / 1e6, A, t_agg * 1e3, ki,
       100 * t_agg * 1e3 / ki, get["miss"] / 1e6, R, t_get * 1e3, kg, 100 * t_get * 1e3 / kg)
open(os.path.join(P, "r01_pmc_summary.txt"), "w").write(hdr + raw)
json.dump({"source": "profiles/r01_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python3 bench.py --no-cpu`, calibrated on k_probe_random in the same runs: 64 B per random 8-byte load, 32 B per atomic)",
           "k_apply_agg_incr": {"fetch_kib": agg["fetch"], "write_kib": agg["write"], "bytes_per_launch": int(bytes_agg), "ops_per_launch": N,
                                "l2_misses": agg["miss"], "memory_side_atomics": agg["atom"]},
           "k_apply_get": {"fetch_kib": get["fetch"], "write_kib": get["write"], "bytes_per_launch": int(bytes_get), "ops_per_launch": N,
                           "l2_misses": get["miss"]}}, open(os.path.join(P, "r01_pmc.json"), "w"), indent=1)
ur = json.load(open(os.path.join(G, "prof_bench_under_rocprof.json")))
open(os.path.join(P, "r01_rocprof_kernel_stats.txt"), "w").write(
    "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu   (MI355X, round 1 final, tools/refresh_profiles.sh; bench line of this "
    "same profiled run: %.0f Mops/s, k_apply_agg<INCR> round-0 avg %.3f ms and k_apply<GET> %.3f ms by HIP events; the full-batch grids below include the 4 all-hit "
    "replays of the steady-state extra and the first batches of the empty table)\n" % (ur["value"], ur["roofline"]["avg_launch_ms"], ur["roofline_get"]["avg_launch_ms"])
    + open(os.path.join(G, "prof_kernel_stats.txt")).read()
    + "\n# per-step spans of the same trace (tools/step_spans.py): span, busy, idle, kernels, prep rounds | top kernels (us)\n"
    + open(os.path.join(G, "prof_step_spans.txt")).read() + "\n# kernel timeline of step 23 (tools/timeline.py)\n"
    + open(os.path.join(G, "prof_timeline_step23.txt")).read())
# the bench line must carry the traffic of THIS refresh: bench.py reads profiles/r01_pmc.json, which was just rewritten
b["roofline"]["traffic"] = int(bytes_agg); b["roofline_get"]["traffic"] = int(bytes_get)
json.dump(b, open(os.path.join(P, "r01_bench.json"), "w"))
print(hdr)
