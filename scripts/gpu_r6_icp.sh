#!/bin/bash
# Round 6 A/B of the search levers (variant libraries: scripts/build_variant.sh): for every library in LIBS the configs[1] registration
# under rocprofv3 --kernel-trace (per-pass durations of the first registration) and the bench value without the profiler; then the ICP
# parity tests on the default library.  LIBS="default v_base ..." (default = open3d_slam_amd/lib/libo3ds_backend.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
M1="${CELL:+--cell $CELL} --no-cpu-baseline --no-f64 --concurrent 0 --m2-frames 0 --large-map ${LARGE:-0} --no-gicp --no-host-seam"
: > $OUT/r6_icp.txt
for v in ${LIBS:-default}; do
  lib=$R/open3d_slam_amd/lib/libo3ds_backend_$v.so; [ "$v" = default ] && lib=$R/open3d_slam_amd/lib/libo3ds_backend.so
  for rep in 1 2; do
    O3DS_BACKEND_LIB=$lib O3DS_BENCH_DETAIL=$OUT/detail_$v.json timeout -k 5 200 python bench.py $M1 --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v', round(d['value']), 'it/s', round(d['ms_per_step']*1e3,1), 'us/step  per pass', round(d['roofline']['avg_launch_us'],2), 'us  frac', round(d['roofline']['frac'],4), 'large', (d.get('m1_large_map') or {}).get('value'), 'pose', d['pose_error_vs_truth'])" | tee -a $OUT/r6_icp.txt
  done
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/prof_$v && O3DS_BACKEND_LIB=$lib O3DS_BENCH_DETAIL=$OUT/detail_prof_$v.json timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o m1 -- python $R/bench.py $M1 --steps 20 --warmup 3 > /dev/null 2>&1
    python $R/scripts/prof_summary.py $OUT/prof_$v/m1_results.db $OUT/rocprof_m1_$v.txt > /dev/null; rm -f $OUT/prof_$v/*.db )
  echo "$v passes: $(grep icp_fused $OUT/rocprof_m1_$v.txt | tail -24 | head -12 | awk '{printf \"%s \", $(NF-6)}')" | tee -a $OUT/r6_icp.txt
done
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 5 ${TEST_TIMEOUT:-600} python -m pytest ${TESTS:-tests/test_icp_gpu.py tests/test_sharded_gpu.py tests/test_edge_parity_gpu.py} -m gpu -q -x --timeout 200 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -8 | tee -a $OUT/r6_icp.txt
fi
