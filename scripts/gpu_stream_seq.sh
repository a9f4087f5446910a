#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/prof_stream
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stream -o s -- python $R/scripts/bench_stream.py --frames 100 > $OUT/stream_prof.json 2>/dev/null
python $R/scripts/prof_summary.py $OUT/prof_stream/s_results.db $OUT/rocprof_stats_stream.txt > /dev/null
python $R/scripts/prof_sequence.py $OUT/prof_stream/s_results.db $OUT/stream_frame_sequence.txt 60 | tail -120
rm -rf $OUT/prof_stream
