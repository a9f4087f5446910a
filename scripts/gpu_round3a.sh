#!/bin/bash
# round 3, first GPU visit: new parity tests first, then the full suite, a bench line, the calibrated traffic passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -x -q -rA --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 600 $O/bench.err
O3DS_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --config 3u --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_selfspawn_2ranks_gloo.json 2> $O/bench_selfspawn.err; echo "selfspawn rc=$?"
timeout 900 bash scripts/gpu_pmc_traffic.sh > $O/pmc_traffic_run.log 2>&1; echo "pmc rc=$?"
tail -40 $O/pmc_traffic.txt
