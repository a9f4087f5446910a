#!/bin/bash
# round-2 visit: new normals kernel -- parity (bitwise f64), tightened pipeline tests, stream kernel profile
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 400 python scripts/check_normals.py > $OUT/check_normals.log 2>&1; echo "rc=$?" >> $OUT/check_normals.log
timeout 900 python -m pytest tests/test_preprocess_map_gpu.py tests/test_pipeline_gpu.py -m gpu -q -s --timeout 400 -k "normals or loop or full_size_stream" > $OUT/pytest_normals.log 2>&1; echo "rc=$?" >> $OUT/pytest_normals.log
cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/prof_stream
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_stream -o s -- python $R/scripts/bench_stream.py --frames 20 --cpu-frames 0 > $OUT/stream_prof.json 2>/dev/null
python $R/scripts/prof_summary.py $OUT/prof_stream/s_results.db $OUT/rocprof_stats_stream.txt > /dev/null
cd $R
timeout 300 python scripts/bench_stream.py --frames 40 --cpu-frames 0 > $OUT/bench_stream.json 2> $OUT/bench_stream.err
tail -30 $OUT/check_normals.log; tail -60 $OUT/pytest_normals.log | cut -c1-300; head -12 $OUT/rocprof_stats_stream.txt | cut -c1-200; cat $OUT/bench_stream.json
