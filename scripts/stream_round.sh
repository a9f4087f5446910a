#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -4
timeout 600 python scripts/bench_stream.py --frames 40 --cpu-frames 5 2>/dev/null | tee $OUT/bench_stream.json | python -c "
import json,sys; d=json.loads(sys.stdin.readline())
for k in ('gpu_scans_per_sec_mapping_only','gpu_scans_per_sec_odometry_plus_mapping','gpu_ms_per_scan','cpu_scans_per_sec_mapping_only','cpu_scans_per_sec_odometry_plus_mapping','cpu_ms_per_scan','map_points','final_pose_error_vs_truth'): print(k, d.get(k))"
cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/prof_stream
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stream -o s -- python $R/scripts/bench_stream.py --frames 20 --cpu-frames 0 > /dev/null 2>&1
python $R/scripts/prof_summary.py $OUT/prof_stream/s_results.db $OUT/rocprof_stats_stream.txt | head -28 | cut -c1-160
