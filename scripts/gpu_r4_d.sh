#!/bin/bash
# the integration header's shared pre-processing: C++ integration tests, the patched reference, host-seam rates shared / unshared
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4d; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_host_adapter.py tests/test_patched_reference_gpu.py tests/test_integration_patch.py -m gpu -q -s 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -40 > $OUT/pytest.log; tail -12 $OUT/pytest.log
cp gpurun_out/patched_reference_stream.json $OUT/patched_shared.json
O3DS_SHARE_PREPROCESS=0 timeout 600 python -m pytest tests/test_patched_reference_gpu.py -m gpu -q -k scans_per_second 2>&1 | tail -2
cp gpurun_out/patched_reference_stream.json $OUT/patched_unshared.json
python - <<PY
import json
for n in ("shared","unshared"):
    d=json.load(open("$OUT/patched_%s.json"%n)); print("patched reference", n, round(d["serial"]["scans_per_sec"]), round(d["two_threads"]["scans_per_sec"]))
PY
cp $OUT/patched_shared.json gpurun_out/patched_reference_stream.json
for v in 1 0; do O3DS_SHARE_PREPROCESS=$v timeout 600 python bench.py --no-cpu-baseline --no-f64 --large-map 0 --concurrent 0 2>/dev/null | tail -1 > $OUT/bench_share$v.json; python - <<PY
import json
d=json.load(open("$OUT/bench_share$v.json")); s=d["scans_per_sec"]; print("share=$v", round(d["value"]), round(s["scans_per_sec"]), s.get("host_seam"))
PY
done
