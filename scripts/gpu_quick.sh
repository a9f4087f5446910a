#!/bin/bash
# Short GPU-box visit: parity tests (incl. the C++ host programs), smoke, one bench line.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 420 python -m pytest tests -m gpu -q -rA --timeout 150 2>&1 | tail -150 > $OUT/pytest_gpu.log
echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
timeout 240 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/bench.err
grep -E "passed|failed|FAILED|CHECK failed" $OUT/pytest_gpu.log | tail -8; tail -2 $OUT/smoke.log; cut -c1-400 $OUT/bench.json; tail -2 $OUT/bench.err
