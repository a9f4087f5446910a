"""The one-ulp wobble of a two-handle run: which quantity differs first -- the odometry poses the mapper reads, the
initial guess numpy composes from them, or the registration's result?  Logs all three per frame in one-handle and two-handle runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from open3d_slam_amd import backend
from open3d_slam_amd.scan_to_map_registration import ScanToMapIcp
from open3d_slam_amd.odometry import LidarOdometry

if os.environ.get("DEBUG_AB"):  # the library with the A/B switches, for every handle
    _load = backend.load
    backend.load = lambda ab=False: _load(True)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scans = bench.make_stream(frames)
LOG = {"guess": [], "result": [], "odom": []}
_reg, _get = ScanToMapIcp.scanToMapRegistration, LidarOdometry.getOdomToRangeSensor
def reg(self, scan, submap, pose, guess):
    LOG["guess"].append(np.array(guess, dtype=np.float64).copy())
    r = _reg(self, scan, submap, pose, guess)
    LOG["result"].append(np.array(r.transformation_).copy())
    return r
def get(self, t):
    T = _get(self, t)
    LOG["odom"].append((t, np.array(T).copy()))
    return T
ScanToMapIcp.scanToMapRegistration, LidarOdometry.getOdomToRangeSensor = reg, get
def run(two):
    for v in LOG.values():
        v.clear()
    if two:
        bench.run_stream_pipelined(0, scans)
    else:
        be = backend.Backend(0); bench.run_stream(be, scans); be.close()
    return {k: list(v) for k, v in LOG.items()}
def first(X, Y, key=lambda a: a):
    for k in range(min(len(X), len(Y))):
        a, b = key(X[k]), key(Y[k])
        if not np.array_equal(a, b):
            return k, float(np.abs(a - b).max()), np.argwhere(a != b).tolist()
    return None
ref = run(False)
for name, two in [("one handle", False)] * 2 + [("two handles", True)] * reps + [("one handle", False)]:
    r = run(two)
    if not two or os.environ.get("DEBUG_QUIET") is None or first(ref["result"], r["result"]):
      print(name, "| odom read:", first(ref["odom"], r["odom"], key=lambda a: a[1]), "| guess:", first(ref["guess"], r["guess"]), "| result:", first(ref["result"], r["result"]), flush=True)
    d = first(ref["guess"], r["guess"])
    if d:
        k = d[0]
        # the same numpy expression again, now, on the logged inputs: does it reproduce either value?
        print("   frame", k, "guess ref - run:\n", ref["guess"][k] - r["guess"][k])
