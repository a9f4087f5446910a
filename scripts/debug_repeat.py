"""Does the one-handle stream repeat bit for bit, run after run, before and after a two-handle run in the same process?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from open3d_slam_amd import backend
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
scans = bench.make_stream(frames)
def serial():
    be = backend.Backend(0); r = bench.run_stream(be, scans); be.close(); return r["poses_per_frame"]
def first(X, Y):
    for k in range(min(len(X), len(Y))):
        if not np.array_equal(X[k], Y[k]):
            return k, float(np.abs(X[k] - Y[k]).max())
    return None
ref = serial()
print("before:", [first(ref, serial()) for _ in range(5)])
P = [bench.run_stream_pipelined(0, scans)["poses_per_frame"] for _ in range(4)]
print("pipelined vs serial:", [first(ref, p) for p in P])
print("after:", [first(ref, serial()) for _ in range(5)])
