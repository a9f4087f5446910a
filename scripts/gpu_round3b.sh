#!/bin/bash
# round 3, second GPU visit: the selection kernel for normals against the ranking kernel, the whole GPU suite, the DRAM-side counter passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0


timeout 1200 python -m pytest tests -m gpu -q -rA --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
timeout 1100 bash scripts/gpu_pmc_traffic.sh > $O/pmc_traffic_run.log 2>&1; echo "pmc rc=$?"; cat $O/pmc_traffic_run.log
