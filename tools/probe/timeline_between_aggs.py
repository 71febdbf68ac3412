"""Kernel timeline between two consecutive full-batch launches of the folding kernel (a whole step) from a rocprofv3 trace."""
import csv, glob, sys
d = sys.argv[1]; which = int(sys.argv[2])
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda t: int(t['Start_Timestamp']))
idx = [i for i, t in enumerate(tr) if 'k_apply_agg<2' in t['Kernel_Name'] and t['Grid_Size_X'] == '8388608']
start = idx[which]; t0 = int(tr[start]['Start_Timestamp']); prev_end = t0
for t in tr[start:idx[which + 1] + 1]:
    n = t['Kernel_Name'].split('(')[0].replace('void ', '')
    a, b = int(t['Start_Timestamp']), int(t['End_Timestamp'])
    print("%9.1f -> %9.1f us  +%8.1f  gap %6.1f  %-28s grid=%s q=%s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, (a - prev_end) / 1e3, n[:28], t['Grid_Size_X'], t.get('Queue_Id', '?')))
    prev_end = max(prev_end, b)
