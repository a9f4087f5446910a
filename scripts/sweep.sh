#!/bin/bash
# quick A/B of launch geometry / cell size on the GPU box; prints one compact line per variant
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { # label, env..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --m2-frames 0 --concurrent 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-28s %8.0f it/s  %7.3f ms/step  kernel %6.1f us  frac %.4f  idx %.1f ms' % ('$label', d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], d['index_build_ms']))"
}
{
run "fused (default)" A=1 --
run "launch" O3DS_ICP_MODE=launch --
run "fused f64" A=1 -- --precision f64
run "fused cell .5" A=1 -- --cell 0.5
run "fused cell .35" A=1 -- --cell 0.35
run "fused r2048" O3DS_PASS_ROWS=2048 --
} | tee $OUT/sweep.txt
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp; rm -rf $OUT/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --m2-frames 0 --concurrent 0 > $OUT/rocprof_bench.json 2> $OUT/rocprof.err
python $R/scripts/prof_summary.py $OUT/prof/bench_results.db $OUT/rocprof_stats.txt | sed -n 14,30p
