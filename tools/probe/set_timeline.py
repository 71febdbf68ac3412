import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda t: int(t['Start_Timestamp']))
idx = [i for i, t in enumerate(tr) if 'k_set_fold' in t['Kernel_Name']]
i0 = idx[1]; t0 = int(tr[i0]['Start_Timestamp'])
for t in tr[i0:i0 + 24]:
    n = t['Kernel_Name'].split('(')[0].replace('void ', '')
    print("%9.1f us  +%8.1f  %-30s grid=%s" % ((int(t['Start_Timestamp']) - t0) / 1e3, (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3, n[:30], t['Grid_Size_X']))
