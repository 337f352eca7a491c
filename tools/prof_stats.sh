#!/bin/bash
# rocprofv3 kernel statistics of one command: tools/prof_stats.sh <tag> <program> [args...]  -> gpurun_out/<tag>_kernel_stats.csv
# (the program itself follows "--": never a shell wrapper, the profiler's preload initialises the GPU first)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
[ -z "$GRAFT_REPO_ROOT" ] && out=/root/repo/gpurun_out
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag --output-format csv -- "$@" > $out/${tag}_run.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats.csv
