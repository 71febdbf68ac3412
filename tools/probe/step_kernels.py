"""The kernels of the last four write steps of a rocprofv3 kernel trace (argv[1]), in launch order: name and duration in us."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda t: int(t['Start_Timestamp']))
idx = [i for i, t in enumerate(tr) if ('k_apply_agg<2' in t['Kernel_Name'] or 'k_apply_agg_clu<2' in t['Kernel_Name']) and t['Grid_Size_X'] == '8388608']
for i0 in idx[-4:]:
    out = []
    for t in tr[i0:i0 + 60]:
        n = t['Kernel_Name'].split('(')[0].replace('void ', '').replace('smx::', '')[:28]
        if n.startswith('at::') or 'rocclr' in n: continue
        if 'k_apply<0' in n: break
        out.append("%s %.0f" % (n, (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3))
    print(" | ".join(out))
