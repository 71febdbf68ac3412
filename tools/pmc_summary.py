#!/usr/bin/env python3
"""Condenses rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per
dispatch (full-batch op kernels are separated by grid size).  usage: pmc_summary.py <dir>..."""
import csv
import glob
import os
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            grid = r.get("Grid_Size", r.get("Grid_Size_X", ""))
            key = (name[:44], grid, r["Counter_Name"])
            acc[key][0] += float(r["Counter_Value"])
            acc[key][1] += 1
        print("# %s" % f)
        for (name, grid, ctr), (tot, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:40]:
            print("%-44s grid=%-10s %-22s dispatches=%4d mean=%16.1f" % (name, grid, ctr, n, tot / n))
