#!/bin/bash
# rocprofv3 kernel statistics of one command: tools/prof_stats.sh <tag> <program> [args...]
#   -> gpurun_out/<tag>_kernel_stats.csv (+ <tag>_kernel_trace.csv: one row per dispatch)
# (the program itself follows "--": never a shell wrapper, the profiler's preload initialises the GPU first)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
[ -z "$GRAFT_REPO_ROOT" ] && out=/root/repo/gpurun_out
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag --output-format csv -- "$@" > $out/${tag}_run.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats.csv
f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 - "$f" $out/${tag}_kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size", "Workgroup_Size", "LDS_Block_Size"]
cols = [c for c in keep if rows and c in rows[0]]
with open(sys.argv[2], "w") as o:
    w = csv.writer(o)
    w.writerow(cols)
    for r in rows[-4000:]:
        w.writerow([r[c][:60] if c == "Kernel_Name" else r[c] for c in cols])
PY
