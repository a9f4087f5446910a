#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for m in 0 1 2 3; do
  rm -rf $OUT/prof_a$m
  O3DS_DEBUG_ACC=$m timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_a$m -o bench -- python $R/bench.py --steps 10 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== acc debug mode $m"; python $R/scripts/prof_summary.py $OUT/prof_a$m/bench_results.db | sed -n 3,4p; python $R/scripts/prof_summary.py $OUT/prof_a$m/bench_results.db | grep -A6 "per-dispatch" | tail -5 | cut -c1-75
done
