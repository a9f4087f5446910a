#!/bin/bash
# timing experiments on the pass kernel (O3DS_DEBUG_ACC): per-dispatch durations of the steady passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for dbg in 0 2 4 5; do
  rm -rf $OUT/prof_$dbg
  O3DS_DEBUG_ACC=$dbg O3DS_ICP_MODE=launch timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$dbg -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --m2-frames 0 --concurrent 0 > /dev/null 2> $OUT/rocprof_$dbg.err
  echo "== debug $dbg"; python $R/scripts/prof_summary.py $OUT/prof_$dbg/bench_results.db /dev/null | grep "icp_accumulate" | sed -n 2,8p | awk '{print $(NF-6)}' | tr '\n' ' '; echo
done 2>&1 | tee $OUT/exp_acc.txt
