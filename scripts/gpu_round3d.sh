#!/bin/bash
# round 3, fourth GPU visit: hand-written sort in the map merge (tests + A/B), the N > 1 bench line with configs 3u and 4 riding along
# (two gloo ranks on the one GPU), counter passes of the stream's kernels (3 frames), instruction mix and phase trace of icp_fused_kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -rA --durations=8 --deselect tests/test_sharded_gpu.py::test_rccl_one_rank_collectives_and_stream_ordering \
  --deselect tests/test_pipeline_gpu.py::test_full_length_stream_200_frames_matches_oracle > $O/pytest_gpu_d.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_d.log
grep -E "passed|failed" $O/pytest_gpu_d.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_gpu_d.log | head -20
for lib in 0 1; do
  if [ $lib -eq 1 ]; then export O3DS_MERGE_LIBRARY_SORT=1; else unset O3DS_MERGE_LIBRARY_SORT; fi
  timeout 200 python scripts/bench_stream.py --frames 120 --profile > $O/stream_libsort_$lib.json 2> $O/stream_libsort_$lib.err; echo "stream library_sort=$lib rc=$?"
  python -c "
import json;d=json.load(open('$O/stream_libsort_$lib.json'))
print({k:d[k] for k in ('scans_per_sec','mapping_only_scans_per_sec','map_points')}, {k[:24]:round(v['avg_us'],1) for k,v in d['calls'].items()})"
done
unset O3DS_MERGE_LIBRARY_SORT
O3DS_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 8 --warmup 2 > $O/bench_n2_gloo_all_configs.json 2> $O/bench_n2_gloo.err; echo "bench --gpus 2 (gloo, auto) rc=$?"
python -c "
import json;d=json.load(open('$O/bench_n2_gloo_all_configs.json'))
print({k:d[k] for k in ('value','n_gpus','ms_per_step')}, {k:(v.get('value'),v.get('error')) for k,v in d.get('also',{}).items()})"
SETS="" STREAM_SETS="dram wr" timeout 500 bash scripts/gpu_pmc_traffic.sh > $O/pmc_stream_run.log 2>&1; echo "pmc stream rc=$?"; cat $O/pmc_stream_run.log
timeout 400 bash scripts/pmc_cmd.sh icp_r03 python bench.py --steps 20 --warmup 3 --m2-frames 0 --no-cpu-baseline --concurrent 0 --no-f64 --large-map 0 > $O/pmc_icp_r03.txt 2>&1; echo "pmc instruction mix rc=$?"; tail -30 $O/pmc_icp_r03.txt | cut -c1-200
O3DS_FUSED_TRACE=$O/fused_trace_r03.txt timeout 200 python bench.py --steps 3 --warmup 1 --m2-frames 0 --no-cpu-baseline --concurrent 0 --no-f64 --large-map 0 > /dev/null 2> $O/fused_trace.err; python scripts/fused_trace.py $O/fused_trace_r03.txt > $O/fused_trace_r03_summary.txt 2>&1; cat $O/fused_trace_r03_summary.txt
