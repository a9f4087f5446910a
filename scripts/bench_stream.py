#!/usr/bin/env python
"""M2 alone (BASELINE.json configs[2], scans/s): the stream leg of bench.py for a chosen number of frames.
    python scripts/bench_stream.py --frames 40 [--profile] [--cpu-frames 6]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--cpu-frames", type=int, default=0)
ap.add_argument("--profile", action="store_true", help="per-call table from hipEvent spans (perturbs the rates slightly)")
ap.add_argument("--free", action="store_true", help="no stream drains between the stages (bench.py's free_running leg): what work queued ahead looks like")
args = ap.parse_args()
scans = bench.make_stream(args.frames)
from open3d_slam_amd import backend
be = backend.Backend(0)
bench.run_stream(be, scans[: min(8, len(scans))])
be.close()
be = backend.Backend(0)
out = bench.run_stream(be, scans, profile=args.profile, stage_sync=not args.free)
be.close()
out.pop("pose")
out.pop("poses_per_frame", None)
if args.cpu_frames > 1:
    out["cpu_baseline"], _ = bench.cpu_baseline_m2(scans, min(args.cpu_frames, len(scans)), min(16, os.cpu_count() or 1))
print(json.dumps(out))
