"""the normal estimation of the config-2 stream, repeated: a frame of the figure-eight run, cropped (2 .. 30 m) and voxel-filtered (0.1 m) as the
stream's scan processing does, then EstimateNormals(r 3 m, knn 20) in f32 storage -- the profiling target beside scripts/normals_one.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, synthetic as syn
scene = syn.make_scene()
poses = syn.figure_eight_poses(200, 0.1)
k = int(os.environ.get("FRAME", "60"))
scan = syn.os128_scan(scene, poses[k], frame=k).astype(np.float32)
be = backend.Backend(0, backend.PRECISION_F32)
c = be.upload(scan.astype(np.float64))
v = be.crop_voxel_down_sample(c, backend.make_crop(backend.CROP_MIN_MAX_RADIUS, rmin=2.0, rmax=30.0), 0.1)
n = be.size(v)[0]
ts = []
for rep in range(int(os.environ.get("REPS", "12"))):
    be.synchronize(); t0 = time.perf_counter(); be.estimate_normals(v, 3.0, 20); be.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"stream frame {k}: n={n} estimate_normals call: min {min(ts):.3f} ms med {sorted(ts)[len(ts)//2]:.3f}")
