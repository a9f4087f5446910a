#!/usr/bin/env python
"""The two-thread / two-handle stream with and without the mapper taking the odometry worker's clouds through views; poses against the one-handle run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from open3d_slam_amd import backend
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
scans = bench.make_stream(frames)
be = backend.Backend(0); bench.run_stream(be, scans[:8]); be.close()
be = backend.Backend(0); ser = bench.run_stream(be, scans); be.close()
be = backend.Backend(0); free = bench.run_stream(be, scans, stage_sync=False); be.close()
print("one handle: staged %.0f scans/s, free-running %.0f" % (ser["scans_per_sec"], free["scans_per_sec"]))
for share, drain in ((False, True), (True, True), (True, False), (False, False), (True, True), (True, False)):
    bench.run_stream_pipelined(0, scans[:12], share=share, drain=drain)
    p = bench.run_stream_pipelined(0, scans, share=share, drain=drain)
    same = all(np.array_equal(a, b) for a, b in zip(ser["poses_per_frame"], p["poses_per_frame"]))
    print("two threads, share=%s drain=%s: %.0f scans/s, poses equal to the one-handle run bit for bit: %s, map %d" % (share, drain, p["scans_per_sec"], same, p["map_points"]))
