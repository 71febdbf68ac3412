"""Kernel time by name over the last N write batches of a rocprofv3 kernel trace (N = argv[2], default 8); with argv[3] = M only the
first M of those N (a window in the middle of the run: N = 22, M = 4 are steps 2..5 of a 24-step run)."""
import csv, glob, sys
d = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda t: int(t['Start_Timestamp']))
idx = [i for i, t in enumerate(tr) if ('k_apply_agg<2' in t['Kernel_Name'] or 'k_apply_agg_clu<2' in t['Kernel_Name']) and t['Grid_Size_X'] == '8388608']
i0 = idx[-last]; agg = {}; cnt = {}
i1 = len(tr)
if len(sys.argv) > 3:
    m = int(sys.argv[3]); i1 = idx[-last + m] if m < last else len(tr); last = min(m, last)
for t in tr[i0:i1]:
    n = t['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    if n.startswith('at::') or 'rocclr' in n: continue
    agg[n] = agg.get(n, 0) + (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3; cnt[n] = cnt.get(n, 0) + 1
print("per step over the last %d steps: busy %.1f us" % (last, sum(agg.values()) / last))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:16]: print("   %-42s %9.1f us  x%.1f" % (k, v / last, cnt[k] / last))
