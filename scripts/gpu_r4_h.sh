#!/bin/bash
# carving without a library sort (partition of the merge's layout; the layout kept through the carve): tests, then the stream with O3DS_CARVE_SORT=1 / default
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4h; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_preprocess_map_gpu.py tests/test_pipeline_gpu.py tests/test_reference_golden_gpu.py tests/test_repro_gpu.py tests/test_patched_reference_gpu.py tests/test_host_adapter.py -m gpu -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -4
for v in sort part sort part; do
  if [ $v = sort ]; then export O3DS_CARVE_SORT=1; else unset O3DS_CARVE_SORT; fi
  python scripts/bench_stream.py --frames 200 2>/dev/null | tail -1 > $OUT/stream_$v.json; python -c "
import json; d=json.load(open('$OUT/stream_$v.json')); print('carve $v', {k: (round(v,1) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,(dict,list))}, d.get('ms_per_scan'), d.get('final_pose_error_vs_truth'))"; done
unset O3DS_CARVE_SORT
cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o s -- python $R/scripts/bench_stream.py --frames 60 > /dev/null 2>&1
python $R/scripts/prof_summary.py $OUT/prof/s_results.db $OUT/stats.txt > /dev/null; echo "library sort kernels in the stream:"; grep -c -i "rocprim" $OUT/stats.txt; grep -i "rocprim" $OUT/stats.txt | cut -c60-110,120-175 | head -5; rm -rf $OUT/prof
