# the default bench line (N = 1), then its summary
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["step_us"])
print(round(d["m1_f64"]["value"]), d["m1_f64"]["step_us"], d["m1_large_map"]["value"], d["concurrent"]["value"])
s=d["scans_per_sec"]; print({k:s[k] for k in ("scans_per_sec","mapping_only_scans_per_sec")}, s["pipelined"]["scans_per_sec"], s.get("host_seam"), s.get("parity_vs_cpu"))
print(d["cpu_baseline"])
PY
