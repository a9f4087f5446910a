"""o3ds_map_insert_scan, span by span: the stream's pre-processed scans into a map seeded with N surface samples (bench.run_insert_sweep's loop,
every insertion's own time)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from open3d_slam_amd import backend, synthetic as syn
n_map = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
frames = 24
scans = bench.make_stream(frames)
if os.environ.get("O3DS_PM_STATS"):
    be = backend.Backend(0, ab=True)
else:
    be = backend.Backend(0)
scene = syn.make_scene(); poses = syn.figure_eight_poses(200, 0.1)
pts, nrm = syn.sample_map(scene, n_map, seed=syn.SEED_MAP + 3)
m = be.upload(pts, nrm)
crop_scan = backend.make_crop(backend.CROP_MIN_MAX_RADIUS, rmin=2.0, rmax=30.0)
pre = []
for k in range(frames):
    raw = be.upload_f32(np.ascontiguousarray(np.hstack([scans[k], np.zeros((len(scans[k]), 1), np.float32)])))
    v = be.crop_voxel_down_sample(raw, crop_scan, 0.1); be.estimate_normals(v, 3.0, 20); be.free(raw); be.size(v); pre.append(v)
be.profile_enable(True)
out = []
for k in range(frames):
    T = np.linalg.inv(poses[0]) @ poses[k]
    crop = backend.make_crop(backend.CROP_MIN_MAX_RADIUS, center=T[:3, 3], rmin=2.0, rmax=30.0)
    be.synchronize(); be.estimate_normals(pre[k], 3.0, 20)
    with be.span(0):
        be.map_insert_scan(m, pre[k], T, 0.1, crop, max_corr_hint=1.0)
    cnt, ms = be.span_read(0)
    out.append(round(1e3 * ms, 1))
print(n_map, be.size(m)[0], out)
