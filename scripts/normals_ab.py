"""A/B of normal-estimation builds or settings (e.g. O3DS_NRM_CELL_SCALE, read once per process): time on the voxel-filtered OS-128-like scan
of the config-2 stream and a checksum of the result, so that variants can be compared bit for bit across processes."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, synthetic as syn
scene = syn.make_scene()
out = []
for name, pose in (("origin", syn.make_pose((0.0, 0.0, 0.5), (0.0, 0.0, 0.0))), ("offset", syn.make_pose((12.0, -7.0, 1.5), (1.0, -2.0, 40.0)))):
    scan = syn.os128_scan(scene, pose)
    for prec, pname in ((backend.PRECISION_F32, "f32"), (backend.PRECISION_F64, "f64")):
        be = backend.Backend(0, prec)
        c = be.upload(scan)
        v = be.voxel_down_sample(c, 0.1)
        n = be.size(v)[0]
        ts = []
        for rep in range(6):
            be.synchronize(); t0 = time.perf_counter(); be.estimate_normals(v, 3.0, 20); be.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        _, nrm = be.download(v)
        h = hashlib.sha1(np.ascontiguousarray(nrm).tobytes()).hexdigest()[:12]
        t5 = []
        for rep in range(3):
            be.synchronize(); t0 = time.perf_counter(); be.estimate_normals(v, 1.0, 5); be.synchronize(); t5.append((time.perf_counter() - t0) * 1e3)
        _, nrm5 = be.download(v)
        h5 = hashlib.sha1(np.ascontiguousarray(nrm5).tobytes()).hexdigest()[:12]
        out.append(f"cell_scale={os.environ.get('O3DS_NRM_CELL_SCALE','1')} {name} {pname} n={n} knn20/r3: min {min(ts):.3f} ms med {sorted(ts)[len(ts)//2]:.3f} sha {h} | knn5/r1: min {min(t5):.3f} sha {h5}")
        be.close()
print("\n".join(out))
