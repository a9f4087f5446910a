#!/bin/bash
# A/B of an ICP kernel knob (env var NAME, values LIST): bench.py value + result checksum, then the ICP parity tests under the last value
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
: > $OUT/icp_ab.txt
for v in $LIST; do
  for rep in 1 2; do
    env $NAME=$v timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --m2-frames 0 --concurrent 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$NAME=$v', round(d['value']), 'it/s', round(d['ms_per_step'],4), 'ms/step  kernel', round(d['roofline']['avg_launch_us'],2), 'us  pose', d['pose_error_vs_truth'])" | tee -a $OUT/icp_ab.txt
  done
done
env $NAME=$v timeout 300 python -m pytest tests/test_icp_gpu.py tests/test_sharded_gpu.py -m gpu -q -x --timeout 150 2>&1 | tail -3 | tee -a $OUT/icp_ab.txt
