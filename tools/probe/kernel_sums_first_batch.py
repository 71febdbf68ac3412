"""Kernel time by name for the first write batch of a rocprofv3 kernel trace (up to the first get kernel)."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda t: int(t['Start_Timestamp']))
# (since round 4 a large batch into an empty matrix starts with k_iota, not with the folding kernel)
i0 = next(i for i, t in enumerate(tr) if 'k_apply_agg<2' in t['Kernel_Name'] or 'k_iota' in t['Kernel_Name'])
i1 = next(i for i, t in enumerate(tr) if i > i0 and 'k_apply<0' in t['Kernel_Name'])
agg = {}; cnt = {}
for t in tr[i0:i1]:
    n = t['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    agg[n] = agg.get(n, 0) + (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3; cnt[n] = cnt.get(n, 0) + 1
span = (int(tr[i1 - 1]['End_Timestamp']) - int(tr[i0]['Start_Timestamp'])) / 1e3
print("span %.1f us, busy %.1f us, %d launches" % (span, sum(agg.values()), i1 - i0))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:24]: print("   %-42s %9.1f us  x%d" % (k, v, cnt[k]))
if len(sys.argv) > 2:
    print("launches of", sys.argv[2], ":", " ".join("%.0f" % ((int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3) for t in tr[i0:i1] if sys.argv[2] in t['Kernel_Name']))
