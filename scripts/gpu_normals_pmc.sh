#!/bin/bash
# normals kernel: kernel time (rocprofv3 --kernel-trace --stats) and instruction mix (PMC passes) of scripts/normals_one.py
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/prof_nrm1
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_nrm1 -o s -- python $R/scripts/normals_one.py > $OUT/normals_one.log 2>/dev/null
python $R/scripts/prof_summary.py $OUT/prof_nrm1/s_results.db $OUT/rocprof_stats_normals_one.txt > /dev/null
cat $OUT/normals_one.log; grep -i "normals\|cell_count\|scatter" $OUT/rocprof_stats_normals_one.txt | cut -c1-60,100-170
cd $R
bash scripts/pmc_cmd.sh normals1 python scripts/normals_one.py 2>&1 | grep "P4f" | grep -v finish | cut -c20-200 > $OUT/pmc_normals_one.txt; cat $OUT/pmc_normals_one.txt
