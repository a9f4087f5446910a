#!/bin/bash
# a short bench line (every leg, few steps / frames) + the frame's dispatch sequence: what a change is checked with before a full round
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 800 python bench.py --steps 60 --m2-frames ${FRAMES:-80} --m2-cpu-frames 0 --no-cpu-baseline --large-map 0 --concurrent 0 --no-f64 ${EXTRA} > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
echo "bench rc=$?"; tail -3 gpurun_out/bench_quick.err | cut -c1-300
python - <<PY
import json
d=json.load(open("gpurun_out/bench_quick.json")); s=d["scans_per_sec"]
print("M1", round(d["value"]), d["ms_per_step"], "frac", round(d["roofline"]["frac"],4), "avg_us", round(d["roofline"]["avg_launch_us"],2))
print("gicp", {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.get("m1_gicp",{}).items() if k in ("value","ms_per_step","parity_vs_oracle","error")}, (d.get("m1_gicp",{}).get("roofline") or {}).get("frac"))
print("M2", {k:round(s[k],1) for k in ("scans_per_sec","mapping_only_scans_per_sec")}, "free", round(s["free_running"]["scans_per_sec"],1), "free w/o ahead", s["free_running"].get("scans_per_sec_without_preprocessing_ahead"), "pageable", round(s["pageable_ingest_at_frame_start"]["scans_per_sec"],1), "pipelined", s["pipelined"].get("scans_per_sec"))
print("shipped", {k:v for k,v in s.get("shipped_configuration",{}).items() if k in ("scans_per_sec","final_pose_error_vs_truth","error")})
print("sweep", s.get("map_insert_scan_by_map_size",{}).get("rows") or s.get("map_insert_scan_by_map_size"))
print("patched", {k:v for k,v in s.get("patched_reference",{}).items() if k!="what"})
print("host_seam", {k:v for k,v in (s.get("host_seam") or {}).items() if k in ("scans_per_sec","two_threads_scans_per_sec","error")})
print({k[:40]:round(v["avg_us"],1) for k,v in s["calls"].items()})
print("bitwise", s["free_running"]["pose_equals_staged_run_bitwise"], s["pageable_ingest_at_frame_start"]["pose_equals_bitwise"], s.get("pose_repeats_bitwise_between_the_two_runs"))
PY
