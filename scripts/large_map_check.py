"""One-off size check beyond the bench configurations: an 8 M-point map (index build, registration of a 64k scan, voxel merge of a cloud
without a known layout: the full-sort path) against the CPU oracle on the same inputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, synthetic as syn
from oracle import pyoracle as po

N = int(os.environ.get("N_MAP", "8000000"))
scene = syn.make_scene()
T_gt = syn.ground_truth_pose()
src = syn.vlp16_scan(scene, T_gt)
tgt, nrm = syn.sample_map(scene, N, seed=7)
be = backend.Backend(0)
t0 = time.perf_counter(); t_id = be.upload(tgt, nrm); be.synchronize(); t1 = time.perf_counter()
be.build_index(t_id, 1.0); be.synchronize(); t2 = time.perf_counter()
s_id = be.upload(src)
r = be.icp_point_to_plane_dev(s_id, t_id, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0); t3 = time.perf_counter()
ref = po.icp_point_to_plane(src, tgt.astype(np.float32).astype(np.float64), nrm.astype(np.float32).astype(np.float64), 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
dt, dr = syn.se3_error(r["transformation"], ref["transformation"])
print(f"map {N}: upload {1e3*(t1-t0):.0f} ms, index {1e3*(t2-t1):.1f} ms, 10-iteration registration {1e3*(t3-t2):.2f} ms; "
      f"n_corr {r['n_corr']} vs {ref['n_corr']}, pose vs oracle dt {dt:.2e} m dr {dr:.2e} rad")
assert r["n_corr"] == ref["n_corr"] or abs(r["n_corr"] - ref["n_corr"]) <= 2, "correspondence counts differ"
assert dt < 1e-6 and dr < 1e-6
# the map merge at this size (full-sort path) keeps every voxel once
crop = backend.make_crop(backend.CROP_MAX_RADIUS, center=(0.0, 0.0, 0.0), rmax=25.0)
be.voxelize_within_volume(t_id, 0.05, crop)
n_after = be.size(t_id)[0]
pts, _ = be.download(t_id)
inside = np.linalg.norm(pts, axis=1) <= 25.0
keys = np.floor(pts[inside] * (1.0 / 0.05)).astype(np.int64)
assert len(np.unique(keys, axis=0)) >= int(0.999 * inside.sum()), "voxels repeated after the merge"  # f32 means can leave a voxel by rounding
print(f"voxelize_within_volume(0.05): {N} -> {n_after} points, {inside.sum()} inside the volume, all in distinct voxels")
print("large map check: OK")
