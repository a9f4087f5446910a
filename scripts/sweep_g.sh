#!/bin/bash
# steady-pass duration of the two-launch pass kernel for G = 2/4/8 (rocprofv3 per-dispatch)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for g in 2 4 8; do
  rm -rf $OUT/prof_g$g
  O3DS_PASS_GROUP=$g O3DS_ICP_MODE=launch timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_g$g -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/g$g.json 2> /dev/null
  echo "== G $g  $(python -c "import json;d=json.loads(open('$OUT/g$g.json').readline());print('%.0f it/s'%d['value'])")"; python $R/scripts/prof_summary.py $OUT/prof_g$g/bench_results.db /dev/null | grep "icp_accumulate" | sed -n 2,12p | awk '{print $(NF-6)}' | tr '\n' ' '; echo
done 2>&1 | tee $OUT/sweep_g.txt
