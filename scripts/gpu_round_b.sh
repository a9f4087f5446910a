#!/bin/bash
# The short second half of an evidence round (what scripts/gpu_round.sh does after its bench line, without the profiler over the whole bench):
# the GPU suite, a bench line with fewer steps / frames and no CPU legs, the stream's kernel stats + dispatch sequence, configs[1] alone under
# rocprofv3.  Every step under its own tight timeout (with -k: a hung profiler child is killed, not waited for).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 5 300 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -25 > $OUT/pytest_gpu_b.log
  echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/pytest_gpu_b.log
fi
FRAMES=${FRAMES:-200} EXTRA="${EXTRA:---no-gicp --no-host-seam}" bash scripts/gpu_r5_quick.sh > $OUT/quick.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_stream
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $OUT/prof_stream -o s -- python $R/scripts/bench_stream.py --frames 100 > $OUT/stream_prof.json 2>/dev/null
python $R/scripts/prof_summary.py $OUT/prof_stream/s_results.db $OUT/rocprof_stats_stream.txt > /dev/null
python $R/scripts/prof_sequence.py $OUT/prof_stream/s_results.db $OUT/stream_frame_sequence.txt 65 > /dev/null
rm -rf $OUT/prof_m1
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $OUT/prof_m1 -o m1 -- python $R/bench.py --no-cpu-baseline --no-f64 --concurrent 0 --m2-frames 0 --large-map 0 --no-gicp --no-host-seam > $OUT/rocprof_m1.json 2> /dev/null
python $R/scripts/prof_summary.py $OUT/prof_m1/m1_results.db $OUT/rocprof_stats_m1.txt > /dev/null
cd $R; rm -rf $OUT/prof_m1/*.db $OUT/prof_stream/*.db
tail -4 $OUT/pytest_gpu_b.log; cat $OUT/quick.log | cut -c1-600; tail -3 $OUT/stream_frame_sequence.txt; head -6 $OUT/rocprof_stats_m1.txt | cut -c1-60,110-170
