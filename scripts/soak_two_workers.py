#!/usr/bin/env python
"""Soak of the two-worker stream (views, deferred frees, device draws): N runs each of the bench and the shipped configuration, every pose of every run
against the one-handle run of the same configuration, bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from open3d_slam_amd import backend
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
scans = bench.make_stream(200)
bad = 0
for shipped in (False, True):
    be = backend.Backend(0); ref = bench.run_stream(be, scans, shipped=shipped); be.close()
    rates = []
    for r in range(runs):
        p = bench.run_stream_pipelined(0, scans, share=True, drain=bool(r & 1), shipped=shipped)
        same = len(p["poses_per_frame"]) == 200 and all(np.array_equal(a, b) for a, b in zip(ref["poses_per_frame"], p["poses_per_frame"])) and p["map_points"] == ref["map_points"]
        bad += 0 if same else 1
        rates.append(round(p["scans_per_sec"]))
    print("shipped" if shipped else "bench  ", "one handle %.0f scans/s; two workers:" % ref["scans_per_sec"], rates, "differing runs so far:", bad, flush=True)
sys.exit(1 if bad else 0)
