"""Kernel time and gaps inside the LAST pipelined incr call of tools/probe/host_trace.py under rocprofv3 --kernel-trace:
python tools/probe/host_trace_kernels.py <trace dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda t: int(t['Start_Timestamp']))
# the chunks' round-0 kernels: k_apply_agg with a grid for 2^21 ops (1024 tiles x 1024 lanes)
idx = [i for i, t in enumerate(tr) if 'k_apply_agg<2' in t['Kernel_Name'] and t['Grid_Size_X'] == str(1024 * 1024)]
last8 = idx[-8:]
i0, i1 = last8[0], len(tr)
t0 = int(tr[i0]['Start_Timestamp']); t1 = max(int(t['End_Timestamp']) for t in tr[i0:i1])
busy = sum(int(t['End_Timestamp']) - int(t['Start_Timestamp']) for t in tr[i0:i1])
print("last call: %d kernels, span %.2f ms, sum of kernel times %.2f ms" % (i1 - i0, (t1 - t0) / 1e6, busy / 1e6))
agg = {}
for t in tr[i0:i1]:
    n = t['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
    a = agg.setdefault(n, [0, 0]); a[0] += (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3; a[1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]: print("   %-46s %8.1f us  x%d" % (k, v[0], v[1]))
# per chunk: span from its round-0 kernel to the next chunk's
for a, b in zip(last8, last8[1:] + [i1]):
    s0 = int(tr[a]['Start_Timestamp']); e = max(int(t['End_Timestamp']) for t in tr[a:b]); nxt = int(tr[b]['Start_Timestamp']) if b < len(tr) else e
    print("   chunk: kernels %3d, busy %.0f us, first start to last end %.0f us, to the next chunk's start %.0f us" % (
        b - a, sum(int(t['End_Timestamp']) - int(t['Start_Timestamp']) for t in tr[a:b]) / 1e3, (e - s0) / 1e3, (nxt - s0) / 1e3))
