#!/usr/bin/env python3
"""Per-step span, busy time and idle time of a bench run from a rocprofv3 kernel-trace dir."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda t: int(t['Start_Timestamp']))
idx = [i for i, t in enumerate(tr) if ('k_apply_agg<2' in t['Kernel_Name'] or 'k_apply_agg_clu<2' in t['Kernel_Name']) and t['Grid_Size_X'] == '8388608']
idx.append(len(tr))
prev_end = None
for s in range(len(idx) - 1):
    seg = tr[idx[s]:idx[s + 1]]
    # the step ends with its get kernel
    e = max(i for i, t in enumerate(seg) if 'k_apply<0' in t['Kernel_Name']) if any('k_apply<0' in t['Kernel_Name'] for t in seg) else len(seg) - 1
    seg = seg[:e + 1]
    t0 = int(seg[0]['Start_Timestamp']); t1 = int(seg[-1]['End_Timestamp'])
    busy = sum(int(t['End_Timestamp']) - int(t['Start_Timestamp']) for t in seg)
    by = {}
    for t in seg:
        n = t['Kernel_Name'].split('(')[0].replace('void ', '')
        by[n] = by.get(n, 0) + (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3
    rounds = sum(1 for t in seg if 'k_prep' in t['Kernel_Name'])
    top = sorted(by.items(), key=lambda kv: -kv[1])[:5]
    # `since` = from the end of the previous step's get kernel to this step's first kernel: the host's time between two steps
    print("step %2d: since %6.1f  span %8.1f us  busy %8.1f  idle %7.1f  kernels %4d rounds %2d | %s" % (
        s, (t0 - prev_end) / 1e3 if prev_end else 0.0, (t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3, len(seg), rounds,
        "  ".join("%s %.0f" % (k[:18], v) for k, v in top)))
    prev_end = t1
