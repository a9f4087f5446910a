#!/bin/bash
# Round 4, GPU visit A: the candidate-set fused kernel against the round-3 library (kept as lib/libo3ds_backend_r3.so, untracked):
# the ICP tests, configs[1] alone per variant (rate, phase trace of launch 5, per-launch counters of how queries were served).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4a
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
M1="python bench.py --no-cpu-baseline --no-f64 --concurrent 0 --m2-frames 0 --large-map 0 --no-host-seam"
one() {  # tag, env...
  tag=$1; shift
  env "$@" $M1 --steps 100 --warmup 10 2>$OUT/$tag.err | grep '^{' | tail -1 > $OUT/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$tag.json")); print("$tag", round(d["value"]), "it/s", round(d["ms_per_step"]*1e3,1), "us/step frac", round(d["roofline"]["frac"],4))
except Exception as e: print("$tag FAILED", e)
PY
  env "$@" O3DS_FUSED_TRACE=$OUT/$tag.trace5 O3DS_FUSED_TRACE_LAUNCH=5 $M1 --steps 2 --warmup 1 >/dev/null 2>&1
  env "$@" O3DS_FUSED_TRACE=$OUT/$tag.trace0 O3DS_FUSED_TRACE_LAUNCH=0 $M1 --steps 2 --warmup 1 >/dev/null 2>&1
  for k in 0 5; do [ -f $OUT/$tag.trace$k ] && python scripts/fused_trace.py $OUT/$tag.trace$k > $OUT/$tag.trace$k.txt 2>&1; done
}
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_sharded_gpu.py -m gpu -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -60 > $OUT/pytest_icp.log
tail -3 $OUT/pytest_icp.log
one new_sets_on O3DS_ICP_SETS=1
O3DS_ICP_STATS=1 $M1 --steps 1 --warmup 1 2>&1 >/dev/null | grep "icp stats" | tail -2 > $OUT/stats_default.txt
cat $OUT/stats_*.txt
for t in new_sets_on; do echo "== $t trace5"; tail -12 $OUT/$t.trace5.txt; done
echo "== new_sets_on trace0"; tail -6 $OUT/new_sets_on.trace0.txt
cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/prof_m1 && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_m1 -o m1 -- python $R/bench.py --no-cpu-baseline --no-f64 --concurrent 0 --m2-frames 0 --large-map 0 --no-host-seam --steps 50 > /dev/null 2>&1
python $R/scripts/prof_summary.py $OUT/prof_m1/m1_results.db $OUT/rocprof_stats_m1.txt > /dev/null; rm -rf $OUT/prof_m1
sed -n 1,6p $OUT/rocprof_stats_m1.txt | cut -c1-60,110-170; sed -n 14,27p $OUT/rocprof_stats_m1.txt | cut -c60-120
