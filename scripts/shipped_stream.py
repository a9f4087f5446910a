#!/usr/bin/env python
"""The shipped configuration's stream alone (GeneralizedIcp, downsampling_ratio 0.3), as bench.py times it: for rocprofv3 / host traces."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open3d_slam_amd import backend
if os.environ.get("DEBUG_AB"):  # the library with the A/B switches (O3DS_SET_CAP, O3DS_SET_GAIN, ...)
    _load = backend.load
    backend.load = lambda ab=False: _load(True)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
scans = bench.make_stream(frames)
be = backend.Backend(0); bench.run_stream(be, scans[:8], shipped="bench" not in sys.argv); be.close()
be = backend.Backend(0)
out = bench.run_stream(be, scans, shipped="bench" not in sys.argv, stage_sync="free" not in sys.argv)
be.close()
print(json.dumps({k: out[k] for k in ("scans_per_sec", "ms_per_scan")}))
