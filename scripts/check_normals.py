"""GPU normals (8 lanes per point) vs the oracle on a voxel-filtered OS-128-like scan; prints agreement and time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, synthetic as syn
from oracle import pyoracle as po
scene = syn.make_scene()
scan = syn.os128_scan(scene, syn.make_pose((0.0, 0.0, 0.5), (0.0, 0.0, 0.0)))
be = backend.Backend(0, backend.PRECISION_F32)
c = be.upload(scan)
v = be.voxel_down_sample(c, 0.1)
pts = be.download(v)[0]
print("points", len(pts))
for rep in range(3):
    t0 = time.perf_counter(); be.estimate_normals(v, 3.0, 20); be.synchronize(); t1 = time.perf_counter()
    print("estimate_normals %.3f ms" % ((t1 - t0) * 1e3))
_, nrm = be.download(v)
sub = np.random.default_rng(0).choice(len(pts), 3000, replace=False)
ref = po.estimate_normals(pts, 3.0, 20)
dots = np.abs(np.sum(ref[sub] * nrm[sub], axis=1))
print("|n.n_ref| median %.6f  p1 %.4f  frac>0.999 %.4f  signs equal %.4f" % (np.median(dots), np.percentile(dots, 1), np.mean(dots > 0.999), np.mean(np.sum(ref[sub] * nrm[sub], axis=1) > 0)))
be.close()
