cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ktc -- python3 $GRAFT_REPO_ROOT/tools/probe/cold_start.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/ktc/**/*kernel_trace.csv', recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda t: int(t['Start_Timestamp']))
i0 = next(i for i, t in enumerate(tr) if 'k_iota' in t['Kernel_Name'])
i1 = next(i for i, t in enumerate(tr) if i > i0 and 'k_apply<0>' in t['Kernel_Name'])
agg = {}; cnt = {}
for t in tr[i0:i1]:
    n = t['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    agg[n] = agg.get(n, 0) + (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3; cnt[n] = cnt.get(n, 0) + 1
span = (int(tr[i1 - 1]['End_Timestamp']) - int(tr[i0]['Start_Timestamp'])) / 1e3
print("first incr batch of config 2 (rocprofv3 kernel trace): span %.1f us, busy %.1f us, %d launches" % (span, sum(agg.values()), i1 - i0))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:26]: print("   %-42s %9.1f us  x%d" % (k, v, cnt[k]))
t0 = int(tr[i0]['Start_Timestamp'])
print("timeline (start us, duration us):")
for t in tr[i0:i1]:
    n = t['Kernel_Name'].split('(')[0].replace('void ', '')[:34]
    print("  %9.1f +%8.1f  %s" % ((int(t['Start_Timestamp']) - t0) / 1e3, (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3, n))
PY
rm -rf gpurun_out/ktc
