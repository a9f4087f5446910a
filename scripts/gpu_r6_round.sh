#!/bin/bash
# (round 6: bench.py prints the compact line; the detail records go to gpurun_out/*_detail.json)
# One GPU-box visit that refreshes the evidence kept under profiles/: the GPU suite, smoke, the bench line (+ rocprofv3 kernel stats of the
# same command and of configs[1] alone), the stream's kernel stats and the dispatch sequence of its frames, the N = 2 line over gloo.
# Outputs under gpurun_out/.  (Counter passes: scripts/gpu_pmc_traffic.sh; the phase trace: O3DS_FUSED_TRACE + scripts/fused_trace.py.)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
{ echo "nproc=$(nproc)"; lscpu | grep -E "Model name|Socket|Core|Thread|MHz"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showproductname 2>/dev/null | grep -i -E "card series|gfx" | head -4; } > $OUT/host.txt 2>&1
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -rA --durations=10 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -260 > $OUT/pytest_gpu.log
  echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
fi
O3DS_BENCH_DETAIL=$OUT/bench_detail.json timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/bench.err
O3DS_BENCH_DETAIL=$OUT/bench_n2_gloo_detail.json O3DS_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 8 --warmup 2 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_n2_gloo_all_configs.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof
O3DS_BENCH_DETAIL=$OUT/rocprof_bench_detail.json timeout 1200 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-host-seam > $OUT/rocprof_bench.json 2> $OUT/rocprof.err
echo "rocprof rc=$?" >> $OUT/rocprof.err
python $R/scripts/prof_summary.py $OUT/prof/bench_results.db $OUT/rocprof_stats.txt > /dev/null
# the stream alone: which kernels a frame is made of, and the order and timing of its dispatches (GPU-busy fraction of a frame)
rm -rf $OUT/prof_stream
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stream -o s -- python $R/scripts/bench_stream.py --frames 100 > $OUT/stream_prof.json 2>/dev/null
python $R/scripts/prof_summary.py $OUT/prof_stream/s_results.db $OUT/rocprof_stats_stream.txt > /dev/null
python $R/scripts/prof_sequence.py $OUT/prof_stream/s_results.db $OUT/stream_frame_sequence.txt 65 > /dev/null
# configs[1] alone (the registrations `value` and `roofline` are quoted on): the fused kernel's average here is the bench line's avg_launch_us
cd /tmp; rm -rf $OUT/prof_m1
O3DS_BENCH_DETAIL=$OUT/rocprof_m1_detail.json timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_m1 -o m1 -- python $R/bench.py --no-cpu-baseline --no-f64 --concurrent 0 --m2-frames 0 --large-map 0 --no-gicp > $OUT/rocprof_m1.json 2> /dev/null
python $R/scripts/prof_summary.py $OUT/prof_m1/m1_results.db $OUT/rocprof_stats_m1.txt > /dev/null
cd $R
rm -rf $OUT/prof/*.db $OUT/prof_m1/*.db $OUT/prof_stream/*.db  # the summaries are what travels back
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3; tail -2 $OUT/smoke.log; tail -1 $OUT/bench.err; head -14 $OUT/rocprof_stats.txt | cut -c1-80,100-170
python - <<PY
import json
d=json.load(open("$OUT/bench_detail.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["avg_launch_us"])
s=d["scans_per_sec"]; print({k:s[k] for k in ("scans_per_sec","mapping_only_scans_per_sec")}, s["pipelined"]["scans_per_sec"], s.get("host_seam"), s.get("parity_vs_cpu"))
print(d.get("m1_large_map")); print(d.get("m1_gicp"))
print(s.get("map_insert_scan_by_map_size"))
PY
# the shipped configuration's stream alone: kernel stats and the dispatch sequence of a frame
cd /tmp; rm -rf $OUT/prof_shipped
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_shipped -o s -- python $R/scripts/shipped_stream.py 100 > $OUT/shipped_stream.json 2>/dev/null
python $R/scripts/prof_summary.py $OUT/prof_shipped/s_results.db $OUT/shipped_stats.txt > /dev/null
python $R/scripts/prof_sequence.py $OUT/prof_shipped/s_results.db $OUT/shipped_sequence.txt 65 > /dev/null
rm -rf $OUT/prof_shipped/*.db; tail -2 $OUT/shipped_sequence.txt
cd $R
