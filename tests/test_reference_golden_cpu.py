"""The CPU oracle against the committed fixtures that the REFERENCE's own code produced (tests/golden/ref_units.npz, made by
tests/golden/make_ref_golden.py from open3d_slam's sources compiled unchanged, oracle/ref_build).  Runs anywhere -- no reference checkout,
no GPU -- so the oracle stays pinned to those outputs wherever the suite runs.  The device path meets the same file in
tests/test_reference_golden_gpu.py."""
import os

import numpy as np

from oracle import pyoracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_units.npz")
g = {k: v for k, v in np.load(GOLDEN).items()}


def _f64(a):
    return np.asarray(a, dtype=np.float64)


def _key_order(p, voxel):
    k = np.floor(p * (1.0 / voxel)).astype(np.int64)
    return np.lexsort((k[:, 2], k[:, 1], k[:, 0]))


def test_oracle_croppers_match_the_reference_fixture():
    pts = _f64(g["crop_pts"])
    rmin, rmax, zmin, zmax = g["crop_params"]
    for kind in range(5):
        for inv in (0, 1):
            c = po.make_crop(kind, center=_f64(g["crop_center"]), rmin=rmin, rmax=rmax, zmin=zmin, zmax=zmax, invert=bool(inv))
            assert np.array_equal(po.crop_indices(pts, c), g[f"crop_idx_{kind}_{inv}"].astype(np.int64)), (kind, inv)


def test_oracle_map_merge_matches_the_reference_fixture():
    pts, nrm, col = _f64(g["vox_pts"]), _f64(g["vox_nrm"]), _f64(g["vox_col"])
    voxel = float(g["vox_voxel"][0])
    c = po.make_crop(po.CROP_MIN_MAX_RADIUS, center=_f64(g["crop_center"]), rmin=g["vox_crop"][0], rmax=g["vox_crop"][1])
    op, on, npass = po.voxelize_within_volume(pts, nrm, voxel, c)
    oc = po.voxelize_within_volume_colors(pts, col, voxel, c)
    assert npass == int(g["vox_npass"][0]) and len(op) == len(g["vox_out_pts"])
    o = np.concatenate([np.arange(npass), npass + _key_order(op[npass:], voxel)])
    assert np.array_equal(op[o], g["vox_out_pts"]) and np.array_equal(on[o], g["vox_out_nrm"], equal_nan=True) and np.array_equal(oc[o], g["vox_out_col"])


def test_oracle_transform_carving_overlap_match_the_reference_fixture():
    pts, nrm = _f64(g["vox_pts"][:1000]), np.nan_to_num(_f64(g["vox_nrm"][:1000]))
    assert np.array_equal(po.transform_points(pts, g["tf_T"]), g["tf_out_pts"]) and np.array_equal(po.transform_normals(nrm, g["tf_T"]), g["tf_out_nrm"])
    mp, mn, scan, sensor = _f64(g["carve_map"]), _f64(g["carve_map_nrm"]), _f64(g["carve_scan"]), _f64(g["carve_sensor"])
    voxel, max_len, trunc, min_dot = g["carve_params"]
    sub = np.flatnonzero(np.linalg.norm(mp - sensor, axis=1) <= float(g["carve_crop_rmax"][0]))
    f = po.carve_flags(scan, sensor, mp, mn, sub, voxel=voxel, max_length=max_len, truncation=trunc, min_dot=min_dot)
    assert np.array_equal(np.flatnonzero(f), g["carve_ids"].astype(np.int64))
    a, b = po.overlap_indices(scan, mp, g["tf_T"], float(g["overlap_params"][0]), int(g["overlap_params"][1]))
    assert np.array_equal(a, g["overlap_src"].astype(np.int64)) and np.array_equal(b, g["overlap_tgt"].astype(np.int64))


def test_oracle_dense_map_and_deskew_match_the_reference_fixture():
    voxel = float(g["dense_voxel"][0])
    op, on, oc = po.dense_fuse(_f64(g["dense_pts"]), _f64(g["dense_nrm"]), voxel)
    o = _key_order(op, voxel)
    assert np.array_equal(op[o], g["dense_out_pts"]) and np.array_equal(on[o], g["dense_out_nrm"]) and np.array_equal(oc[o], g["dense_out_cnt"])
    radius, max_len, trunc = g["dense_carve_params"]
    rem = po.dense_carve(_f64(g["dense_carve_scan"]), np.zeros(3), op, voxel, radius=float(radius), max_length=float(max_len), truncation=float(trunc))
    got = set(map(tuple, np.floor(op[rem] * (1.0 / voxel)).astype(np.int64)))
    assert got == set(map(tuple, g["dense_carve_keys"].astype(np.int64))) and len(got) > 100
    for cw in (0, 1):
        out = po.undistort(_f64(g["deskew_pts"]), g["deskew_vel"][:3], g["deskew_vel"][3:], float(g["deskew_scan_duration"][0]), bool(cw))
        assert np.abs(out - g[f"deskew_out_{cw}"]).max() <= 2e-14
