#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 420 python -m pytest tests -m gpu -q --timeout 150 2>&1 | tail -4 | tee $OUT/pytest_tail.txt
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export O3DS_NO_BOX_CACHE=1; else unset O3DS_NO_BOX_CACHE; fi
  timeout 200 python scripts/bench_stream.py --frames 40 --cpu-frames 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('no_box_cache=$v', round(d['gpu_scans_per_sec_mapping_only'],1), round(d['gpu_scans_per_sec_odometry_plus_mapping'],1), d['gpu_ms_per_scan'], d['map_points'], d['final_pose_error_vs_truth'])" | tee -a $OUT/box_ab.txt
done
unset O3DS_NO_BOX_CACHE
timeout 100 python scripts/stream_calls.py 2>&1 | tail -17 | head -8 | tee $OUT/stream_calls.txt
