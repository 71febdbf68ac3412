"""Kernel time by name over a whole rocprofv3 kernel trace (argv[1] = its directory): what a run's first batch spends where
(python tools/probe/dense_steps.py 1 under --kernel-trace = the cold start of the dense-id stream)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
agg = {}; cnt = {}; mx = {}
for t in csv.DictReader(open(f)):
    n = t['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
    if n.startswith('at::') or 'rocclr' in n or 'elementwise' in n: continue
    d = (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3
    agg[n] = agg.get(n, 0) + d; cnt[n] = cnt.get(n, 0) + 1; mx[n] = max(mx.get(n, 0), d)
print("busy %.1f us" % sum(agg.values()))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:24]: print("   %-46s %10.1f us  x%-4d longest %9.1f" % (k, v, cnt[k], mx[k]))
if len(sys.argv) > 2:      # argv[2] = a kernel name: its launches in order (us, grid)
    tr = sorted(csv.DictReader(open(f)), key=lambda t: int(t['Start_Timestamp']))
    print(sys.argv[2] + ":", " ".join("%.0f/%s" % ((int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3, t['Grid_Size_X']) for t in tr if sys.argv[2] in t['Kernel_Name']))
