#!/bin/bash
# Round profiles on the GPU box: tools/collect_profiles.sh <rNN>
#   1. rocprofv3 --kernel-trace --stats of the bench command           -> gpurun_out/<rNN>_kernel_stats_bench.csv
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_* in small groups) -> gpurun_out/<rNN>_pmc_{traffic,sq}.json
# Counter passes never carry a trace option (gpurun refuses that combination); the program itself follows "--".
r=${1:-r02}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
rm -rf /tmp/${r}_*
rocprofv3 --kernel-trace --stats -d /tmp/${r}_stats -o st --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/${r}_stats_run.log 2>&1
cp $(find /tmp/${r}_stats -name "*kernel_stats.csv" | head -1) $out/${r}_kernel_stats_bench.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/${r}_$c -- $B > $out/${r}_pmc_$c.log 2>&1
done
python3 $root/tools/pmc_traffic.py /tmp/${r}_FETCH_SIZE /tmp/${r}_WRITE_SIZE $out/${r}_pmc_traffic.json
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d /tmp/${r}_sq$i -- $B > $out/${r}_pmc_sq$i.log 2>&1
done
python3 $root/tools/pmc_sq.py $out/${r}_pmc_sq.json /tmp/${r}_sq1 /tmp/${r}_sq2 /tmp/${r}_sq3 /tmp/${r}_sq4 > $out/${r}_pmc_sq_summary.txt 2>&1
ls -la $out/${r}_*
