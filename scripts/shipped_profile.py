#!/usr/bin/env python
"""The shipped configuration's stream (GeneralizedIcp, downsampling_ratio 0.3) with the per-call table of hipEvent spans: where 1.6 ms per frame go."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open3d_slam_amd import backend
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
scans = bench.make_stream(frames)
be = backend.Backend(0); bench.run_stream(be, scans[:8], shipped=True); be.close()
be = backend.Backend(0)
out = bench.run_stream(be, scans, profile=True, shipped=True)
be.close()
print(json.dumps({k: out[k] for k in ("scans_per_sec", "ms_per_scan")}))
for k, v in sorted(out.get("calls", {}).items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("calls", 0)):
    print(f"{k[:70]:70s} calls {v.get('calls'):5d} avg {v.get('avg_us'):8.1f} us  total {v.get('avg_us') * v.get('calls') / 1e3:8.2f} ms")
