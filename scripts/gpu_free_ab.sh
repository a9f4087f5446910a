#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 200 python scripts/stress.py 2>&1 | tail -3
NODL=1 timeout 100 python scripts/debug_stream.py 2>&1 | tail -3
timeout 420 python -m pytest tests -m gpu -q --timeout 200 2>&1 | tail -3
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export O3DS_SYNC_ON_FREE=1; else unset O3DS_SYNC_ON_FREE; fi
  timeout 200 python scripts/bench_stream.py --frames 40 --cpu-frames 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('sync_on_free=$v', round(d['gpu_scans_per_sec_mapping_only'],1), round(d['gpu_scans_per_sec_odometry_plus_mapping'],1), d['map_points'], d['final_pose_error_vs_truth'])"
done
