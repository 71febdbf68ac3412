#!/usr/bin/env python3
"""Prints the kernel timeline of one bench step from a rocprofv3 kernel-trace dir."""
import csv, glob, sys
d = sys.argv[1]; which = sys.argv[2] if len(sys.argv) > 2 else "-1"
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda t: int(t['Start_Timestamp']))
idx = [i for i, t in enumerate(tr) if ('k_apply_agg<2' in t['Kernel_Name'] or 'k_apply_agg_clu<2' in t['Kernel_Name']) and t['Grid_Size_X'] == '8388608']
if which == "last-growing":
    # the last step of the growing table: the last full-grid launch of the folding kernel that is followed by a growth round
    # (the all-hit replays behind it have none; since round 4 the first batch of an empty matrix has no such launch at all)
    ends = idx[1:] + [len(tr)]
    us = lambda t: (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3
    grow = [k for k, (a, b2) in enumerate(zip(idx, ends)) if sum(us(t) for t in tr[a:b2] if 'k_grow_' in t['Kernel_Name']) >= 100.0]
    which = grow[-1]
    print("# step %d of the trace (the last one whose growth kernels ran for 100 us or more)" % which)
which = int(which)
start = idx[which]; t0 = int(tr[start]['Start_Timestamp'])
agg = {}
for t in tr[start:start + 200]:
    n = t['Kernel_Name'].split('(')[0].replace('void ', '')
    dur = (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3
    print("%9.1f us  +%8.1f  %-28s grid=%s" % ((int(t['Start_Timestamp']) - t0) / 1e3, dur, n[:28], t['Grid_Size_X']))
    agg[n] = agg.get(n, 0) + dur
    if 'k_apply<0' in n or 'k_get_clu' in n:
        end = int(t['End_Timestamp']); break
print("step span %.1f us; kernel time by name:" % ((end - t0) / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]): print("   %-30s %9.1f us" % (k, v))
