#!/bin/bash
# normals kernel: two builds of the backend (LIBS) on scripts/normals_one.py under rocprofv3 (kernel time), then the bitwise check of the last one
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
for lib in $LIBS; do
  rm -rf $OUT/prof_nl
  O3DS_BACKEND_LIB=$R/$lib REPS=20 timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_nl -o s -- python $R/scripts/normals_one.py > $OUT/nl.log 2>/dev/null
  python $R/scripts/prof_summary.py $OUT/prof_nl/s_results.db $OUT/nl_stats.txt > /dev/null
  echo "$lib: $(grep 'normals_kernel' $OUT/nl_stats.txt | cut -c100-160) | $(cat $OUT/nl.log)"
done
done
cd $R
O3DS_BACKEND_LIB=$R/$lib timeout 300 python scripts/check_normals.py 2>&1 | tail -1
