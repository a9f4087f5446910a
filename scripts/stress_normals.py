"""Development aid: repeats estimate_normals configurations around device-pool disturbances and reports hashes; with an instrumented
library (O3DS_BACKEND_LIB=..., built with -DO3DS_NRM_CHECK) and O3DS_NRM_DEBUG=1 the invariant-violation counters are printed."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, synthetic as syn
prec = backend.PRECISION_F64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else backend.PRECISION_F32
scene = syn.make_scene()
scan = syn.os128_scan(scene, syn.make_pose((0.0, 0.0, 0.5), (0.0, 0.0, 0.0)))
be = backend.Backend(0, prec)
c = be.upload(scan)
v = be.voxel_down_sample(c, 0.1)
seen = {}
for rnd in range(4):
    for radius, knn in ((3.0, 20), (1.0, 5), (0.5, 30), (2.0, 48)):
        for rep in range(3):
            be.estimate_normals(v, radius, knn)
            _, nrm = be.download(v)
            h = hashlib.sha1(np.ascontiguousarray(nrm).tobytes()).hexdigest()[:12]
            seen.setdefault((radius, knn), []).append(h)
        junk = [be.upload(np.random.default_rng(rnd * 10 + i).normal(size=(100_000 + 50_000 * ((rnd + i) % 4), 3))) for i in range(3)]
        for j in junk:
            be.free(j)
for k, hs in seen.items():
    print(k, "distinct hashes:", len(set(hs)), hs[:3], flush=True)
