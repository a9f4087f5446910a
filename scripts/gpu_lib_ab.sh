#!/bin/bash
# A/B of two builds of the backend (LIBS: paths relative to the repository; O3DS_BACKEND_LIB selects one): configs[1] rate and the stream
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
: > $OUT/lib_ab.txt
for rep in 1 2; do
  for lib in $LIBS; do
    O3DS_BACKEND_LIB=$R/$lib timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-f64 --concurrent 0 --m2-frames ${M2:-100} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); s=d.get('scans_per_sec') or {}
print('$lib', round(d['value']), 'it/s  kernel', round(d['roofline']['avg_launch_us'],2), 'us  pose', d['pose_error_vs_truth']['dt_m'], ' stream', round(s.get('scans_per_sec',0),1), round(s.get('mapping_only_scans_per_sec',0),1), (s.get('calls') or {}).get('icp_fused_kernel launches of the stream'))" | tee -a $OUT/lib_ab.txt
  done
done
