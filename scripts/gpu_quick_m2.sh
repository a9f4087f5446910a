# tests that cover the pre-processing chain, then the stream legs of the bench line
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_preprocess_map_gpu.py tests/test_edge_parity_gpu.py tests/test_pipeline_gpu.py tests/test_reference_golden_gpu.py -m gpu -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -15
timeout 600 python bench.py --no-cpu-baseline --large-map 0 --concurrent 0 --no-f64 > gpurun_out/bench_m2.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/bench_m2.json")); s=d["scans_per_sec"]
print(round(d["value"]), {k:round(s[k],1) for k in ("scans_per_sec","mapping_only_scans_per_sec")}, s.get("free_running"), round(s["pipelined"]["scans_per_sec"],1), {k:v for k,v in s["host_seam"].items() if k in ("scans_per_sec","two_threads_scans_per_sec")})
print({k:round(v["avg_us"],1) for k,v in s["calls"].items()})
PY
