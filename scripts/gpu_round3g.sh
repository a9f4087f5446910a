#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_preprocess_map_gpu.py tests/test_pipeline_gpu.py tests/test_repro_gpu.py -m gpu -q \
  --deselect tests/test_pipeline_gpu.py::test_full_length_stream_200_frames_matches_oracle > $O/pytest_gpu_g.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu_g.log
for lib in 0 1 0 1; do
  if [ $lib -eq 1 ]; then export O3DS_MERGE_LIBRARY_SORT=1; else unset O3DS_MERGE_LIBRARY_SORT; fi
  timeout 200 python scripts/bench_stream.py --frames 120 --profile > $O/stream_libsort_$lib.json 2> $O/stream_libsort_$lib.err; echo "stream library_sort=$lib rc=$?"
  python -c "
import json;d=json.load(open('$O/stream_libsort_$lib.json'))
print({k:round(d[k],1) for k in ('scans_per_sec','mapping_only_scans_per_sec','map_points')}, {k[:24]:round(v['avg_us'],1) for k,v in d['calls'].items()})"
done
unset O3DS_MERGE_LIBRARY_SORT
bash scripts/gpu_stream_prof.sh | grep -E "sort_|rocprim|kernel  " | cut -c1-200
