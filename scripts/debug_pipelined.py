"""Where the two-thread / two-handle stream (bench.run_stream_pipelined) first differs from the one-handle stream, and whether it repeats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from open3d_slam_amd import backend
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
scans = bench.make_stream(frames)
be = backend.Backend(0); ser = bench.run_stream(be, scans); be.close()
p1 = bench.run_stream_pipelined(0, scans); p2 = bench.run_stream_pipelined(0, scans)
S, A, B = ser["poses_per_frame"], p1["poses_per_frame"], p2["poses_per_frame"]
def first(X, Y):
    for k in range(min(len(X), len(Y))):
        if not np.array_equal(X[k], Y[k]):
            return k, float(np.abs(X[k] - Y[k]).max())
    return None
print("serial vs pipelined:", first(S, A), "| pipelined vs pipelined:", first(A, B))
print("odometry poses, pipelined vs pipelined:", first(p1["odometry_poses_per_frame"], p2["odometry_poses_per_frame"]))
k = first(A, B)
if k:
    print("frame", k[0], "pose A - B:\n", A[k[0]] - B[k[0]])
from open3d_slam_amd import pointcloud as pc
pc.SHARE_PREPROCESS = False
be = backend.Backend(0); ser2 = bench.run_stream(be, scans); be.close()
print("serial (pre-processing twice) vs pipelined:", first(ser2["poses_per_frame"], A), "| vs serial shared:", first(ser2["poses_per_frame"], S))
