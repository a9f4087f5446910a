#!/usr/bin/env python
"""A few frames of the configs[2] stream with no warm-up pass: short enough to run under `rocprofv3 --pmc` (every dispatch is serialised
there; 40 frames did not finish in five minutes).  python scripts/stream_few_frames.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open3d_slam_amd import backend
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
scans = [bench._scan_job(k) for k in range(frames)]  # no worker pool: forked children hang under rocprofv3
be = backend.Backend(0)
out = bench.run_stream(be, scans)
be.close()
print({k: out[k] for k in ("scans_per_sec", "map_points")})
