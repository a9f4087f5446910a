"""Submap::carve on a 1 M-point map with a full-size raw scan, a few times over -- run under rocprofv3 --kernel-trace --stats to see what a
carving costs (it runs every 10th insertion of the configs[2] stream)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_amd import backend, synthetic as syn  # noqa: E402

scene = syn.make_scene()
poses = syn.figure_eight_poses(200, 0.1)
mp, mn = syn.sample_map(scene, 1_000_000, seed=3)
be = backend.Backend(0)
T = poses[50]
raw = np.asarray(syn.os128_scan(scene, T, frame=50), dtype=np.float64)
crop = backend.make_crop(backend.CROP_MIN_MAX_RADIUS, center=T[:3, 3], rmin=2.0, rmax=30.0)
m = be.upload(mp, mn)
be.build_index(m, 1.0)
s = be.upload(raw)
be.map_carve(m, s, T, crop)
be.synchronize()
t0 = time.perf_counter()
n = 0
for _ in range(10):
    n += be.map_carve(m, s, T, crop)
be.synchronize()
print("ms per carve (scan already on the device): %.3f" % ((time.perf_counter() - t0) * 100), "removed after the first:", n, "map", be.size(m)[0])
