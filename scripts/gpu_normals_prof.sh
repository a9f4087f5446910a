#!/bin/bash
# normals kernel: correctness (check_normals.py), kernel time per cell-size scale (rocprofv3 --kernel-trace --stats on normals_ab.py),
# instruction mix (PMC) with "pmc" as first argument
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 300 python scripts/check_normals.py > $OUT/check_normals.log 2>&1; grep -c "points != oracle: 0 " $OUT/check_normals.log; tail -1 $OUT/check_normals.log
cd /tmp; export TMPDIR=/tmp
for sc in ${SCALES:-1.0}; do
  rm -rf $OUT/prof_nrm
  O3DS_NRM_CELL_SCALE=$sc timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_nrm -o s -- python $R/scripts/normals_ab.py > $OUT/normals_ab_$sc.log 2>/dev/null
  python $R/scripts/prof_summary.py $OUT/prof_nrm/s_results.db $OUT/rocprof_stats_normals_$sc.txt > /dev/null
  echo "== cell scale $sc"; grep -i "normals" $OUT/rocprof_stats_normals_$sc.txt | cut -c1-60,100-170; cat $OUT/normals_ab_$sc.log
done
cd $R
if [ "$1" = "pmc" ]; then bash scripts/pmc_cmd.sh normals python scripts/normals_ab.py 2>&1 | grep "P4f" | grep "INSTS\|WAVES\|WAVE_CYCLES\|GUI" | cut -c20-200 > $OUT/pmc_normals.txt; cat $OUT/pmc_normals.txt; fi
