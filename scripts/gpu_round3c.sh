#!/bin/bash
# round 3, third GPU visit: the tile kernel for normals (A/B against the ranking kernel), the tests that exercise it, the host-seam A/B of the
# narrowed staging copies
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0


timeout 900 python -m pytest tests -m gpu -q -rA --durations=8 --deselect tests/test_sharded_gpu.py::test_rccl_one_rank_collectives_and_stream_ordering \
  --deselect tests/test_pipeline_gpu.py::test_full_length_stream_200_frames_matches_oracle > $O/pytest_gpu_c.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_c.log
grep -E "passed|failed" $O/pytest_gpu_c.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_gpu_c.log | head -20
for sel in 0 1; do
  O3DS_NRM_SELECT=$sel timeout 200 python scripts/bench_stream.py --frames 100 > $O/stream_nrmselect_$sel.json 2> $O/stream_nrmselect_$sel.err; echo "stream select=$sel rc=$?"
  python -c "import json;d=json.load(open('$O/stream_nrmselect_$sel.json'));print({k:d[k] for k in ('scans_per_sec','mapping_only_scans_per_sec','ms_per_scan','map_points')})"
done
for nn in 0 1; do
  if [ $nn -eq 1 ]; then export O3DS_NO_HOST_NARROW=1; else unset O3DS_NO_HOST_NARROW; fi
  timeout 200 python scripts/stream_integration.py --frames 100 --mode serial > $O/host_seam_nonarrow_$nn.json 2> $O/host_seam_$nn.err; echo "host seam no_narrow=$nn rc=$?"; cat $O/host_seam_nonarrow_$nn.json | cut -c1-700
done
unset O3DS_NO_HOST_NARROW
