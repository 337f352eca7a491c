"""Aggregate k_ring launches of a rocprofv3 kernel trace by grid size (= ring length class). Development aid."""
import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ring" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"][:24], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", "?"))
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[key][0] += 1
        agg[key][1] += dur
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-26s grid=%-8s wg=%-5s lds=%-7s calls=%5d avg_us=%9.1f share=%5.1f%%" % (k + (v[0], v[1] / v[0], 100 * v[1] / tot)))
