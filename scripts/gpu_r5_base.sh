#!/bin/bash
# round 5 baseline: GPU suite + default bench line on today's box (before the stream rework)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -40 > $OUT/base_pytest_gpu.log
echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/base_pytest_gpu.log
timeout 900 python bench.py > $OUT/base_bench.json 2> $OUT/base_bench.err
echo "bench rc=$?" >> $OUT/base_bench.err
tail -5 $OUT/base_pytest_gpu.log; tail -2 $OUT/base_bench.err
python - <<PY
import json
d=json.load(open("$OUT/base_bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"])
s=d["scans_per_sec"]; print({k:s[k] for k in ("scans_per_sec","mapping_only_scans_per_sec")}, s["pipelined"]["scans_per_sec"], s.get("host_seam"))
for k,v in s.get("calls",{}).items(): print(k[:60], round(v["avg_us"],1), v["calls"])
PY
