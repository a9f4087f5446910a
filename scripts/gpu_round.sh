#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
{ echo "nproc=$(nproc)"; lscpu | grep -E "Model name|Socket|Core|Thread|MHz" ; } > $OUT/host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rA --timeout 900 2>&1 | tail -150 > $OUT/pytest_gpu.log
echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/bench.err
for c in 0.125 0.175 0.35 0.5; do
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --cell $c > $OUT/bench_cell_$c.json 2>> $OUT/bench.err
done
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --precision f64 > $OUT/bench_f64.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/rocprof_bench.json 2> $OUT/rocprof.err
echo "rocprof rc=$?" >> $OUT/rocprof.err
python $R/scripts/prof_summary.py $OUT/prof/bench_results.db $OUT/rocprof_stats.txt > /dev/null
tail -5 $OUT/pytest_gpu.log; cat $OUT/bench.json; tail -3 $OUT/bench.err; head -8 $OUT/rocprof_stats.txt
