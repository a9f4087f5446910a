#!/bin/bash
# the boundary a maintainer links, one thread and two: the C++ host seam (integration/o3ds_open3d_slam.hpp, tests/cpp/stream_integration.cpp)
# and the patched reference (tests/test_patched_reference_gpu.py writes gpurun_out/patched_reference_stream.json)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
  for mode in serial threads; do echo "host seam $mode: $(timeout 300 python scripts/stream_integration.py --frames ${FRAMES:-200} --mode $mode 2>&1 | tail -1 | cut -c1-400)"; done
done
if [ -z "$SKIP_PATCHED" ]; then
  timeout 900 python -m pytest tests/test_patched_reference_gpu.py tests/test_host_adapter.py tests/test_integration_patch.py -m gpu -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -4
  python - <<PY
import json
d=json.load(open("$OUT/patched_reference_stream.json")); print("patched reference: serial", round(d["serial"]["scans_per_sec"]), "two threads", round(d["two_threads"]["scans_per_sec"]))
PY
fi
