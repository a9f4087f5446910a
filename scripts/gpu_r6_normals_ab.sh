#!/bin/bash
# the normals kernel's average over the 120-frame stream (rocprofv3 --kernel-trace --stats) and the stream rate, per library
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for v in $LIBS; do
  lib=$R/open3d_slam_amd/lib/libo3ds_backend_$v.so; [ "$v" = default ] && lib=$R/open3d_slam_amd/lib/libo3ds_backend.so
  rm -rf $OUT/prof_n
  O3DS_BACKEND_LIB=$lib timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_n -o s -- python $R/scripts/bench_stream.py --frames ${FRAMES:-120} > $OUT/nab.json 2>/dev/null
  python $R/scripts/prof_summary.py $OUT/prof_n/s_results.db $OUT/nab.txt > /dev/null
  echo "$v: $(grep normals_kernel $OUT/nab.txt | head -2 | awk '{print $(NF-1)}' | tr '\n' ' ') us avg (search, finish) | $(python -c "import json; print(round(json.load(open('$OUT/nab.json'))['scans_per_sec'],1))") scans/s under the profiler"
done; done
rm -rf $OUT/prof_n
