#!/bin/bash
# SQ / traffic counters of one command in separate rocprofv3 --pmc passes (never with a trace option; the program itself
# follows "--"): tools/pmc_run.sh <tag> <program> [args...]   -> gpurun_out/<tag>_pmc_sq.json, <tag>_pmc_sq_summary.txt,
# <tag>_pmc_traffic.json
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/${tag}_*
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d /tmp/${tag}_sq$i -- "$@" > $out/${tag}_pmc_sq$i.log 2>&1
done
python3 $root/tools/pmc_sq.py $out/${tag}_pmc_sq.json /tmp/${tag}_sq1 /tmp/${tag}_sq2 /tmp/${tag}_sq3 /tmp/${tag}_sq4 > $out/${tag}_pmc_sq_summary.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/${tag}_$c -- "$@" > $out/${tag}_pmc_$c.log 2>&1
done
python3 $root/tools/pmc_traffic.py /tmp/${tag}_FETCH_SIZE /tmp/${tag}_WRITE_SIZE $out/${tag}_pmc_traffic.json > /dev/null 2>&1
tail -n 40 $out/${tag}_pmc_sq_summary.txt
