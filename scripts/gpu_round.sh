#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench line (+ rocprofv3 kernel stats of the same command), stream profile.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
{ echo "nproc=$(nproc)"; lscpu | grep -E "Model name|Socket|Core|Thread|MHz" ; } > $OUT/host.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -rA --timeout 400 2>&1 | tail -150 > $OUT/pytest_gpu.log
echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline > $OUT/rocprof_bench.json 2> $OUT/rocprof.err
echo "rocprof rc=$?" >> $OUT/rocprof.err
python $R/scripts/prof_summary.py $OUT/prof/bench_results.db $OUT/rocprof_stats.txt > /dev/null
cd $R
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3; tail -2 $OUT/smoke.log; cat $OUT/bench.json; tail -2 $OUT/bench.err; head -14 $OUT/rocprof_stats.txt | cut -c1-80,100-170
