#!/bin/bash
# normal-estimation kernel: time + checksums (scripts/normals_ab.py), work counters (scripts/normals_stats.py, needs the -DO3DS_NRM_STATS
# build lib/libo3ds_backend_stats.so), its parity tests, the config-2 stream
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 120 python scripts/normals_ab.py 2>&1 | tail -6 | tee $OUT/normals_ab.txt
[ -f open3d_slam_amd/lib/libo3ds_backend_stats.so ] && timeout 60 python scripts/normals_stats.py 2>&1 | tail -22 | tee $OUT/normals_stats.txt
timeout 200 python -m pytest tests/test_preprocess_map_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x --timeout 150 -k "normal or pipeline or odometry or voxel" 2>&1 | tail -3 | tee -a $OUT/normals_ab.txt
timeout 200 python scripts/bench_stream.py --frames 40 --cpu-frames 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline())
for k in ('gpu_scans_per_sec_mapping_only','gpu_scans_per_sec_odometry_plus_mapping','gpu_ms_per_scan'): print(k, d.get(k))" | tee -a $OUT/normals_ab.txt
