#!/bin/bash
# the registration's wait on the pinned stamp (O3DS_ICP_WATCH_STATE): tests, then configs[1] alone with the watch on / off, twice
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4e; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_sharded_gpu.py -m gpu -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -4
for v in 1 0 1 0; do O3DS_ICP_WATCH_STATE=$v timeout 300 python bench.py --no-cpu-baseline --no-f64 --concurrent 0 --m2-frames 0 --large-map 0 2>/dev/null | tail -1 > $OUT/m1_watch$v.json; python - <<PY
import json
d=json.load(open("$OUT/m1_watch$v.json")); print("watch=$v", round(d["value"]), d["ms_per_step"], d["roofline"]["frac"])
PY
done
