#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 420 python -m pytest tests -m gpu -q --timeout 150 2>&1 | tail -4 | tee $OUT/pytest_tail.txt
timeout 100 python scripts/stream_calls.py 2>&1 | tail -26 | tee $OUT/stream_calls.txt
cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/prof_stream
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_stream -o s -- python $R/scripts/bench_stream.py --frames 20 --cpu-frames 0 > /dev/null 2>&1
python $R/scripts/prof_summary.py $OUT/prof_stream/s_results.db $OUT/rocprof_stats_stream.txt | head -34 | cut -c1-175
rm -rf $OUT/prof_stream
