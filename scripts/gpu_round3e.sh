#!/bin/bash
# round 3, fifth GPU visit: 8-way merge passes of the hand-written sort (tests + A/B against rocPRIM), counter passes over the stream's kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_preprocess_map_gpu.py tests/test_pipeline_gpu.py tests/test_repro_gpu.py -m gpu -q -rA --durations=5 \
  --deselect tests/test_pipeline_gpu.py::test_full_length_stream_200_frames_matches_oracle > $O/pytest_gpu_e.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_e.log
grep -E "passed|failed" $O/pytest_gpu_e.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_gpu_e.log | head -20
for lib in 0 1 0 1; do
  if [ $lib -eq 1 ]; then export O3DS_MERGE_LIBRARY_SORT=1; else unset O3DS_MERGE_LIBRARY_SORT; fi
  timeout 200 python scripts/bench_stream.py --frames 120 --profile > $O/stream_libsort_$lib.json 2> $O/stream_libsort_$lib.err; echo "stream library_sort=$lib rc=$?"
  python -c "
import json;d=json.load(open('$O/stream_libsort_$lib.json'))
print({k:round(d[k],1) for k in ('scans_per_sec','mapping_only_scans_per_sec','map_points')}, {k[:24]:round(v['avg_us'],1) for k,v in d['calls'].items()})"
done
unset O3DS_MERGE_LIBRARY_SORT
SETS="" STREAM_SETS="dram wr" timeout 400 bash scripts/gpu_pmc_traffic.sh > $O/pmc_stream_run.log 2>&1; echo "pmc stream rc=$?"; cat $O/pmc_stream_run.log; grep -A60 "configs\[2\] stream" $O/pmc_traffic.txt | cut -c1-200 | head -80
