"""Summarise rocprofv3 --pmc CSV output per kernel (development aid)."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-28s %16.4g  (per dispatch %.4g)" % (c, v, v / max(cnt[(k, c)], 1)))
