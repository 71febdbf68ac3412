#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats CSV directory into a short text summary
(the file committed under profiles/).  usage: prof_summary.py <dir> [<out.txt>]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    out.write("# %s\n%-58s %6s %11s %11s %7s %10s %10s\n" % (os.path.basename(f), "kernel", "calls", "total_ms", "avg_us", "pct", "min_us", "max_us"))
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "")
        out.write("%-58s %6s %11.3f %11.1f %7.2f %10.1f %10.1f\n" % (
            name[:58], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
            float(r["Percentage"]), int(r["MinNs"]) / 1e3, int(r["MaxNs"]) / 1e3))
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    tr = list(csv.DictReader(open(f)))
    # round-0 op kernels = full-batch grids
    big = {}
    for t in tr:
        n = t["Kernel_Name"]
        if "k_apply" not in n:
            continue
        key = (n.split("(")[0].replace("void ", ""), t["Grid_Size_X"])
        big.setdefault(key, []).append(int(t["End_Timestamp"]) - int(t["Start_Timestamp"]))
    out.write("# op-kernel launches by (kernel, grid): count, avg_us (steady state = the full-batch grid)\n")
    for (k, g), v in sorted(big.items(), key=lambda kv: -len(kv[1]))[:6]:
        out.write("%-40s grid=%-10s n=%4d avg_us=%10.1f min_us=%10.1f max_us=%10.1f\n" % (k, g, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
