"""Fixtures produced by the REFERENCE's own code: open3d_slam's croppers.cpp / helpers.cpp / Voxel.cpp / VoxelHashMap.cpp /
MotionCompensation.cpp compiled unchanged from the checkout (oracle/ref_build -> oracle/_ref/libo3dslam_ref.so) and run on seeded inputs.
Run where /root/reference exists:   python tests/golden/make_ref_golden.py   -> tests/golden/ref_units.npz
The GPU box has no reference checkout; tests/test_reference_golden_gpu.py compares the HIP path with this file there.
Inputs are stored as float32 (exactly representable, half the bytes) and widened by the reader; outputs whose order is the reference's hash
map's are stored in voxel-key order."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402

OUT = os.environ.get("O3DS_REF_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_units.npz")


def f32(a):
    return np.asarray(a, dtype=np.float32)


def key_order(p, voxel):
    k = np.floor(p * (1.0 / voxel)).astype(np.int64)
    return np.lexsort((k[:, 2], k[:, 1], k[:, 0]))


def main():
    rng = np.random.default_rng(20240925)
    g = {}
    # ---- croppers (a8): every volume, plain and inverted, with no-return points and boundary values
    n = 3000
    pts = f32(rng.normal(size=(n, 3)) * [12, 12, 3])
    pts[::97] = np.nan
    pts[5::131, 1] = np.inf
    t = f32([1.5, -2.0, 0.25])
    pts[1] = t + f32([10.0, 0, 0])
    pts[2] = t + f32([0, 2.0, 0])
    pose = np.eye(4)
    pose[:3, 3] = t
    g["crop_pts"], g["crop_center"] = pts, t
    g["crop_params"] = np.array([2.0, 10.0, -1.0, 2.0])  # rmin rmax zmin zmax
    for kind in range(5):
        for inv in (0, 1):
            idx, _, _, _ = ref.crop(pts.astype(np.float64), kind, rmin=2.0, rmax=10.0, zmin=-1.0, zmax=2.0, pose=pose, invert=bool(inv))
            g[f"crop_idx_{kind}_{inv}"] = idx.astype(np.int32)
    # ---- voxelizeWithinCroppingVolume (a9) incl. NaN normals and colours
    n = 2500
    vp = f32(rng.normal(size=(n, 3)) * [3, 3, 0.4])
    vn = rng.normal(size=(n, 3))
    vn = f32(vn / np.linalg.norm(vn, axis=1, keepdims=True))
    vn[::53] = np.nan
    vc = f32(rng.uniform(size=(n, 3)))
    g["vox_pts"], g["vox_nrm"], g["vox_col"] = vp, vn, vc
    g["vox_voxel"] = np.array([0.3])
    g["vox_crop"] = np.array([1.0, 5.0])  # MinMaxRadius about crop_center
    rp, rn, rc = ref.voxelize_within_cropping_volume(vp.astype(np.float64), 0.3, ref.CROP_MIN_MAX_RADIUS, rmin=1.0, rmax=5.0, pose=pose,
                                                     nrm=vn.astype(np.float64), col=vc.astype(np.float64))
    inside = ref.crop(vp.astype(np.float64), ref.CROP_MIN_MAX_RADIUS, rmin=1.0, rmax=5.0, pose=pose)[0]
    npass = n - len(inside)
    o = key_order(rp[npass:], 0.3)
    g["vox_npass"] = np.array([npass])
    g["vox_out_pts"] = np.vstack([rp[:npass], rp[npass:][o]])
    g["vox_out_nrm"] = np.vstack([rn[:npass], rn[npass:][o]])
    g["vox_out_col"] = np.vstack([rc[:npass], rc[npass:][o]])
    # ---- o3d_slam::transform (a3)
    A = np.eye(4)
    th = 0.3
    A[:3, :3] = [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]
    A[:3, 3] = [1.0, 2.0, 3.0]
    tp, tn, _ = ref.transform(A, vp[:1000].astype(np.float64), np.nan_to_num(vn[:1000]).astype(np.float64))
    g["tf_T"], g["tf_out_pts"], g["tf_out_nrm"] = A, tp, tn
    # ---- space carving of the sparse map (f2)
    m = 4000
    mp = f32(rng.uniform(-10, 10, size=(m, 3)) * [1, 1, 0.2])
    mn = rng.normal(size=(m, 3))
    mn = f32(mn / np.linalg.norm(mn, axis=1, keepdims=True))
    sensor = f32([0.5, -0.25, 0.75])
    scan = f32(rng.uniform(-9, 9, size=(1500, 3)) * [1, 1, 0.2])
    g["carve_map"], g["carve_map_nrm"], g["carve_scan"], g["carve_sensor"] = mp, mn, scan, sensor
    g["carve_params"] = np.array([0.2, 15.0, 0.3, 0.3])  # voxel max_length truncation min_dot
    sub = np.flatnonzero(np.linalg.norm(mp.astype(np.float64) - sensor.astype(np.float64), axis=1) <= 9.0)
    g["carve_crop_rmax"] = np.array([9.0])
    ids = ref.carved_idxs(scan.astype(np.float64), sensor.astype(np.float64), mp.astype(np.float64), mn.astype(np.float64), sub, voxel=0.2,
                          max_length=15.0, truncation=0.3, min_dot=0.3)
    g["carve_ids"] = ids.astype(np.int32)
    # ---- overlap (f3)
    so, to = ref.overlap_indices(scan.astype(np.float64), mp.astype(np.float64), A, 1.0, 2)
    g["overlap_params"] = np.array([1.0, 2.0])
    g["overlap_src"], g["overlap_tgt"] = so.astype(np.int32), to.astype(np.int32)
    # ---- dense voxel map (f2): three scans fused, then carved
    dp = f32(rng.normal(size=(3000, 3)) * [1.5, 1.5, 0.4])
    dn = f32(rng.normal(size=(3000, 3)))
    fp, fn, fc, fk = ref.dense_fuse(dp.astype(np.float64), dn.astype(np.float64), 0.25, batches=3)
    o = np.lexsort((fk[:, 2], fk[:, 1], fk[:, 0]))
    g["dense_pts"], g["dense_nrm"], g["dense_voxel"] = dp, dn, np.array([0.25])
    g["dense_out_pts"], g["dense_out_nrm"], g["dense_out_cnt"], g["dense_out_keys"] = fp[o], fn[o], fc[o], fk[o]
    # ... and moved as Submap::transform moves it (VoxelizedPointCloud::transform, Voxel.cpp:49-64: keys stay, the isometry is applied to the SUMS)
    tp_, tn_, tc_, tk_ = ref.dense_fuse(dp.astype(np.float64), dn.astype(np.float64), 0.25, batches=3, T_after=A)
    o2 = np.lexsort((tk_[:, 2], tk_[:, 1], tk_[:, 0]))
    assert np.array_equal(tk_[o2], fk[o])
    g["dense_tf_out_pts"], g["dense_tf_out_nrm"] = tp_[o2], tn_[o2]
    dscan = f32(rng.normal(size=(400, 3)) * [1.5, 1.5, 0.4])
    keys = ref.dense_carve_keys(dscan.astype(np.float64), np.zeros(3), dp.astype(np.float64), 0.25, radius=0.25, max_length=10.0, truncation=0.2,
                                dedup_scan=True)
    ko = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
    g["dense_carve_scan"], g["dense_carve_params"], g["dense_carve_keys"] = dscan, np.array([0.25, 10.0, 0.2]), keys[ko]
    # ---- constant-velocity de-skew (f4)
    up = f32(rng.normal(size=(1500, 3)) * [8, 8, 1])
    for cw in (0, 1):
        out, vel = ref.undistort(up.astype(np.float64), [0.12, -0.03, 0.01], [0.01, -0.02, 0.15], 0.1, 0.1, bool(cw))
        g[f"deskew_out_{cw}"] = out
    g["deskew_pts"], g["deskew_vel"], g["deskew_scan_duration"] = up, vel, np.array([0.1])
    np.savez_compressed(OUT, **g)
    print(OUT, os.path.getsize(OUT), "bytes;", {k: v.shape for k, v in g.items() if v.ndim > 1 and len(v) > 500})


if __name__ == "__main__":
    main()
