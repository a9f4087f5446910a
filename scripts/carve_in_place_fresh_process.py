"""One persistent-map run of the carved-in-place sequence in a fresh process; prints the map sizes and what the carve said it removed (58 / 56
points at f64 / f32, every time).  The reproducer of the scalar-load / vector-store hazard of DESIGN.md 4.7: before the fix one run in ten removed 0."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from open3d_slam_amd import backend  # noqa: E402
import numpy as np  # noqa: E402
from open3d_slam_amd import synthetic as syn  # noqa: E402

prec = sys.argv[1]
p = backend.PRECISION_F64 if prec == "f64" else backend.PRECISION_F32
be = backend.Backend(0, p)
scene = syn.make_scene()
m = be.upload(np.zeros((0, 3)))
sizes, removed = [], None
for k in range(13):
    T = syn.make_pose([1.5 * k, 0.4 * k, 0.0], [0.0, 0.0, 4.0 * k])
    raw = syn.vlp16_scan(scene, T, frame=k, n_az=256)
    s = be.upload(raw)
    v = be.voxel_down_sample(s, 0.1)
    be.estimate_normals(v, 2.0, 10)
    crop = backend.make_crop(backend.CROP_MIN_MAX_RADIUS, center=T[:3, 3], rmin=0.0, rmax=12.0)
    if k == 8:
        removed = be.map_carve(m, s, T, crop, voxel=0.2)
    be.map_insert_scan(m, v, T, 0.2, crop, max_corr_hint=1.0)
    if k in (5, 12):
        sizes.append(len(be.download(m)[0]))
    be.free(s)
    be.free(v)
print(prec, "removed", removed, "sizes", sizes, flush=True)
