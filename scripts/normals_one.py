"""one configuration of the normal estimation, repeated: the profiling target of scripts/gpu_normals_pmc.sh (f32 storage, knn 20 / r 3 on the
voxel-filtered OS-128-like scan)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_amd import backend, synthetic as syn
scene = syn.make_scene()
scan = syn.os128_scan(scene, syn.make_pose((0.0, 0.0, 0.5), (0.0, 0.0, 0.0)))
be = backend.Backend(0, backend.PRECISION_F32)
c = be.upload(scan)
v = be.voxel_down_sample(c, 0.1)
n = be.size(v)[0]
ts = []
for rep in range(int(os.environ.get("REPS", "12"))):
    be.synchronize(); t0 = time.perf_counter(); be.estimate_normals(v, 3.0, 20); be.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"n={n} estimate_normals call: min {min(ts):.3f} ms med {sorted(ts)[len(ts)//2]:.3f}")
