set -x
R=$GRAFT_REPO_ROOT
bash $R/tools/collect_profiles.sh r03 > $R/gpurun_out/r03_collect.log 2>&1
for c in cfg2 cfg4 cfg5; do bash $R/tools/prof_stats.sh r03_$c python3 $R/tools/cr_time.py $c > /dev/null 2>&1; done
bash $R/tools/prof_stats.sh r03_cfg3pol python3 $R/tools/cr_time.py cfg3 diagonal pol > /dev/null 2>&1
bash $R/tools/pmc_run.sh r03_cfg4 python3 $R/tools/cr_time.py cfg4 > $R/gpurun_out/r03_pmc_cfg4.log 2>&1
bash $R/tools/pmc_run.sh r03_cfg3pol python3 $R/tools/cr_time.py cfg3 diagonal pol > $R/gpurun_out/r03_pmc_cfg3pol.log 2>&1
ls $R/gpurun_out | grep r03_ | head -60
