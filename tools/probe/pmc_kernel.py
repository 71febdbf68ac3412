"""Mean of every counter column per kernel name over a rocprofv3 --pmc run (counter_collection.csv)."""
import csv, glob, sys
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
    if pat not in n: continue
    k = (n, r['Counter_Name'])
    a = acc.setdefault(k, [0.0, 0]); a[0] += float(r['Counter_Value']); a[1] += 1
for (n, c), (s, k) in sorted(acc.items()): print("%-46s %-26s mean %14.1f  x%d" % (n, c, s / k, k))
