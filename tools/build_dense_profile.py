#!/usr/bin/env python3
"""Turns gpurun_out/prof_dense_* (tools/refresh_dense_profile.sh) into profiles/r06_dense_kernels.txt and profiles/pmc_dense.json."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.build_profiles_common import kernel_source_sha16
RND = "r06"
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
rd = lambda n: open(os.path.join(G, n)).read()
run = rd("prof_dense_run.txt").strip().splitlines()
last8, early, seq = rd("prof_dense_kernels_last8.txt"), rd("prof_dense_kernels_steps2to5.txt"), rd("prof_dense_sequence.txt")
pmc = {t: rd("prof_dense_pmc_%s.txt" % t) for t in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum")}


def mean(txt, kern, ctr):
    m = re.search(r"%s\S*\s+%s\s+mean\s+([\d.]+)\s+x(\d+)" % (re.escape(kern), ctr), txt)
    return (float(m.group(1)), int(m.group(2))) if m else (None, 0)


def us(txt, kern):
    m = re.search(r"%s\S*\s+([\d.]+) us\s+x([\d.]+)" % re.escape(kern), txt)
    return float(m.group(1)) if m else None


PASS = "smx::k_apply_wpo_far<2>"
fetch, n1 = mean(pmc["FETCH_SIZE"], PASS, "FETCH_SIZE"); write, _ = mean(pmc["WRITE_SIZE"], PASS, "WRITE_SIZE")
hit, _ = mean(pmc["TCC_HIT_sum"], PASS, "TCC_HIT_sum"); miss, _ = mean(pmc["TCC_HIT_sum"], PASS, "TCC_MISS_sum")
scan_f, _ = mean(pmc["FETCH_SIZE"], "smx::k_far_scan", "FETCH_SIZE")
out = ["# dense-id config 2 (ids = Zipf ranks, unscrambled): tools/probe/dense_steps.py 24 = bench.py's dense_ids leg, under rocprofv3 (tools/refresh_dense_profile.sh)",
       "# " + run[-1] if run else "", "# " + (run[-2] if len(run) > 1 else ""),
       "# ---- kernel time by name, per step over the last 8 steps (rocprofv3 --kernel-trace)", last8.rstrip(),
       "# ---- the same over steps 2..5 (the young table: long deferred lists, many doublings)", early.rstrip(),
       "# ---- the kernel sequence of the last two incr batches (name us | ...)", seq.rstrip(),
       "# ---- PMC means per launch (separate --pmc passes; FETCH_SIZE / WRITE_SIZE in KiB)"] + [pmc[t].rstrip() for t in pmc]
out.append("# the pass in front of prep (k_apply_wpo_far, all launches of the run): fetch %.0f KiB + write %.0f KiB per launch = %.3f GB moved (round 4's walking pass: 1.06 GB fetched per launch for 260 000 far ops);"
           % (fetch or 0, write or 0, ((fetch or 0) + (write or 0)) * 1024 / 1e9))
out.append("# the join's scan (k_far_scan): %.0f KiB fetched per launch = every row of >= 512 cells once" % (scan_f or 0))
open(os.path.join(P, RND + "_dense_kernels.txt"), "w").write("\n".join(out) + "\n")
json.dump({"summary": "profiles/%s_dense_kernels.txt" % RND, "kernel_source_sha16": kernel_source_sha16(), "kernel": "k_apply_wpo_far<INCR> (the pass in front of prep)",
           "avg_launch_ms_last8": (us(last8, PASS) or 0) / 1e3, "fetch_bytes_per_launch": (fetch or 0) * 1024, "write_bytes_per_launch": (write or 0) * 1024,
           "tcc_hit": hit, "tcc_miss": miss, "r04_walking_pass_fetch_bytes_per_launch": 1.06e9,
           "busy_us_per_step_last8": float(re.search(r"busy ([\d.]+) us", last8).group(1))},
          open(os.path.join(P, "pmc_dense.json"), "w"), indent=1)
print(open(os.path.join(P, "pmc_dense.json")).read())
