"""Debugging aid: one estimate_normals configuration on the voxel-filtered OS-128-like scan, with stage markers (O3DS_NRM_DEBUG)."""
import os, sys
os.environ["O3DS_NRM_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, synthetic as syn
from oracle import pyoracle as po
radius, knn, prec = float(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
scene = syn.make_scene()
scan = syn.os128_scan(scene, syn.make_pose((0.0, 0.0, 0.5), (0.0, 0.0, 0.0)))
be = backend.Backend(0, backend.PRECISION_F64 if prec == "f64" else backend.PRECISION_F32)
c = be.upload(scan)
v = be.voxel_down_sample(c, 0.1)
pts = be.download(v)[0]
print("points", len(pts), flush=True)
for rep in range(2):
    be.estimate_normals(v, radius, knn); be.synchronize()
_, nrm = be.download(v)
ref = po.estimate_normals(pts, radius, knn)
print("points != oracle:", int(np.sum(np.any(nrm != ref, axis=1))))
