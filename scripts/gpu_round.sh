#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats, PMC traffic.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
{ echo "nproc=$(nproc)"; lscpu | grep -E "Model name|Socket|Core|Thread|MHz" ; } > $OUT/host.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q -rA --timeout 120 2>&1 | tail -150 > $OUT/pytest_gpu.log
echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/rocprof_bench.json 2> $OUT/rocprof.err
echo "rocprof rc=$?" >> $OUT/rocprof.err
python $R/scripts/prof_summary.py $OUT/prof/bench_results.db $OUT/rocprof_stats.txt > /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python $R/bench.py --steps 10 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python $R/bench.py --steps 10 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/scripts/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json > /dev/null
cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic_latest.json  # bench.py reports roofline.traffic from this file
cd $R
timeout 300 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/bench.err
timeout 400 python scripts/bench_stream.py --frames 40 --cpu-frames 5 > $OUT/bench_stream.json 2> $OUT/bench_stream.err
O3DS_FUSED_TRACE=$OUT/fused_trace.txt timeout 60 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
O3DS_FUSED_TRACE_LAUNCH=0 O3DS_FUSED_TRACE=$OUT/fused_trace0.txt timeout 60 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd /tmp; rm -rf $OUT/prof_stream
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_stream -o s -- python $R/scripts/bench_stream.py --frames 20 --cpu-frames 0 > /dev/null 2>&1
python $R/scripts/prof_summary.py $OUT/prof_stream/s_results.db $OUT/rocprof_stats_stream.txt > /dev/null
cd $R
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3; tail -2 $OUT/smoke.log; cat $OUT/bench.json; tail -2 $OUT/bench.err; head -5 $OUT/rocprof_stats.txt; cat $OUT/pmc_traffic.json
