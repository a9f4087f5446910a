import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from open3d_slam_amd import backend, synthetic as syn
scene = syn.make_scene()
poses = syn.figure_eight_poses(200, 0.1)
mp, mn = syn.sample_map(scene, 1_000_000, seed=3)
be = backend.Backend(0)
T = poses[50]
raw = np.asarray(syn.os128_scan(scene, T, frame=50), dtype=np.float64)
crop = backend.make_crop(backend.CROP_MIN_MAX_RADIUS, center=T[:3, 3], rmin=2.0, rmax=30.0)
s = be.upload(raw)
for L, md in ((20.0, 0.5), (10.0, 0.5), (5.0, 0.5), (2.5, 0.5), (20.0, 2.0)):
    ts = []
    for rep in range(4):
        m = be.upload(mp, mn)
        be.synchronize(); t0 = time.perf_counter(); n = be.map_carve(m, s, T, crop, max_length=L, min_dot=md); be.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        be.free(m)
    print(f"max_length {L:5.1f} min_dot {md}: carve {min(ts):.3f} ms, removed {n}")
