#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for m in 0 1 2; do
  rm -rf $OUT/prof_m$m
  O3DS_DEBUG_UPDATE=$m O3DS_PASS_BLOCK=512 O3DS_PASS_ROWS=768 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_m$m -o bench -- python $R/bench.py --steps 10 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== debug mode $m"; python $R/scripts/prof_summary.py $OUT/prof_m$m/bench_results.db | sed -n 3,4p
done
