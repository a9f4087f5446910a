#!/bin/bash
# per-pass kernel durations of the configs[1] registration under the timing switches of O3DS_DEBUG_ACC (A/B libraries only):
# 0 = as shipped, 2 = no search at all, 16 = no stage 3, 32 = stage-3 loop without its search, 3 = no winner gather
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in ${LIBS}; do for dbg in ${DBG:-0 2 16 32}; do
  rm -rf $OUT/prof_dbg
  O3DS_DEBUG_ACC=$dbg O3DS_BACKEND_LIB=$R/open3d_slam_amd/lib/libo3ds_backend_$v.so timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $OUT/prof_dbg -o m1 -- python $R/scripts/icp_trace.py --one > /dev/null 2>&1
  python $R/scripts/prof_summary.py $OUT/prof_dbg/m1_results.db $OUT/dbg.txt > /dev/null
  echo "$v debug=$dbg: $(grep icp_fused $OUT/dbg.txt | tail -24 | head -12 | awk '{printf "%s ", $(NF-6)}')"
done; done
rm -rf $OUT/prof_dbg
