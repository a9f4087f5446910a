#!/bin/bash
# candidate-set tuning (A/B library: O3DS_SET_CAP / O3DS_SET_GAIN / O3DS_SET_MIN): per-pass durations of the configs[1] registration, point-to-plane and GICP
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
LIB=$R/open3d_slam_amd/lib/libo3ds_backend_${LIBV:-ab}.so
for meth in ${METHODS:-p2l gicp}; do for cfg in ${CFGS}; do
  cap=${cfg%%:*}; gain=${cfg##*:}
  rm -rf $OUT/prof_sets
  METHOD=$meth O3DS_SET_CAP=$cap O3DS_SET_GAIN=$gain O3DS_BACKEND_LIB=$LIB timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $OUT/prof_sets -o m1 -- python $R/scripts/icp_trace.py --one > /dev/null 2>&1
  python $R/scripts/prof_summary.py $OUT/prof_sets/m1_results.db $OUT/sets.txt > /dev/null
  echo "$meth cap=$cap gain=$gain: $(grep icp_fused $OUT/sets.txt | tail -24 | head -12 | awk '{s+=$(NF-6); printf "%s ", $(NF-6)} END {printf "| sum %.1f", s}')"
done; done
rm -rf $OUT/prof_sets
