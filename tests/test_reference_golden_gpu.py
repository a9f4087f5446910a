"""The HIP path, through the C-ABI, against fixtures the REFERENCE's own code produced: tests/golden/ref_units.npz holds the outputs of
open3d_slam's croppers.cpp / helpers.cpp / Voxel.cpp / VoxelHashMap.cpp / MotionCompensation.cpp, compiled unchanged from the checkout
and run on seeded inputs (tests/golden/make_ref_golden.py; oracle/ref_build).  No oracle in between: device result == reference result.
Bit for bit at f64 storage wherever the reference's arithmetic is plain double sums and comparisons (rows a3, a8, a9, f2 carving, f3);
the dense voxel map keeps fixed-point sums (2^-30 m) and the de-skew goes through a rotation, so those two compare to 1e-8 m."""
import os

import numpy as np
import pytest

from open3d_slam_amd import backend

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_units.npz")


@pytest.fixture(scope="module")
def g():
    return {k: v for k, v in np.load(GOLDEN).items()}


def _f64(a):
    return np.asarray(a, dtype=np.float64)


def _key_order(p, voxel):
    k = np.floor(p * (1.0 / voxel)).astype(np.int64)
    return np.lexsort((k[:, 2], k[:, 1], k[:, 0]))


def test_croppers_keep_exactly_the_points_the_reference_keeps(backend_f64, backend_f32, g):
    """croppers.cpp:65-165 -- every volume, plain and inverted, NaN / inf points, values exactly on the radii"""
    pts = _f64(g["crop_pts"])
    rmin, rmax, zmin, zmax = g["crop_params"]
    for be in (backend_f64, backend_f32):  # the inputs are float32 values: both storages hold them exactly
        cid = be.upload(pts)
        for kind in range(5):
            for inv in (0, 1):
                crop = backend.make_crop(kind, center=_f64(g["crop_center"]), rmin=rmin, rmax=rmax, zmin=zmin, zmax=zmax, invert=bool(inv))
                out = be.crop_cloud(cid, crop)
                got = be.download(out)[0]
                want = pts[g[f"crop_idx_{kind}_{inv}"]]
                assert got.shape == want.shape, (kind, inv, got.shape, want.shape)
                assert np.array_equal(got, want, equal_nan=True), (kind, inv)
                be.free(out)
        be.free(cid)


def test_voxelize_within_cropping_volume_equals_the_reference_bit_for_bit(backend_f64, g):
    """helpers.cpp:115-183: pass-through points first and untouched, voxel means / re-normalised normals / last colour inside"""
    pts, nrm, col = _f64(g["vox_pts"]), _f64(g["vox_nrm"]), _f64(g["vox_col"])
    voxel = float(g["vox_voxel"][0])
    rmin, rmax = g["vox_crop"]
    crop = backend.make_crop(backend.CROP_MIN_MAX_RADIUS, center=_f64(g["crop_center"]), rmin=rmin, rmax=rmax)
    be = backend_f64
    m = be.upload(pts, nrm)
    be.set_colors(m, col)
    be.voxelize_within_volume(m, voxel, crop)
    gp, gn = be.download(m)
    gc = be.get_colors(m)
    npass = int(g["vox_npass"][0])
    assert len(gp) == len(g["vox_out_pts"])
    o = np.concatenate([np.arange(npass), npass + _key_order(gp[npass:], voxel)])
    assert np.array_equal(gp[o], g["vox_out_pts"])
    assert np.array_equal(gn[o], g["vox_out_nrm"], equal_nan=True)
    assert np.array_equal(gc[o], g["vox_out_col"])
    be.free(m)


def test_transform_equals_the_reference_bit_for_bit(backend_f64, g):
    """o3d_slam::transform (helpers.cpp:273-305) away from the identity"""
    pts, nrm = _f64(g["vox_pts"][:1000]), np.nan_to_num(_f64(g["vox_nrm"][:1000]))
    be = backend_f64
    c = be.upload(pts, nrm)
    t = be.transform_cloud(c, g["tf_T"])
    gp, gn = be.download(t)
    # (the cloud kernels are compiled without fused multiply-adds, as the reference's arithmetic is: the same bits)
    assert np.array_equal(gp, g["tf_out_pts"]) and np.array_equal(gn, g["tf_out_nrm"])
    be.free(c)
    be.free(t)


def test_space_carving_removes_exactly_the_reference_s_points(backend_f64, g):
    """Submap::carve (Submap.cpp:109-125) = getIndicesWithinVolume + getIdxsOfCarvedPoints (helpers.cpp:221-271) + removeByIds"""
    mp, mn, scan, sensor = _f64(g["carve_map"]), _f64(g["carve_map_nrm"]), _f64(g["carve_scan"]), _f64(g["carve_sensor"])
    voxel, max_len, trunc, min_dot = g["carve_params"]
    pose = np.eye(4)
    pose[:3, 3] = sensor
    raw = scan - sensor  # exact: float32 values; the device places it back with the identity rotation
    assert np.array_equal(raw + sensor, scan)
    be = backend_f64
    m, s = be.upload(mp, mn), be.upload(raw)
    crop = backend.make_crop(backend.CROP_MAX_RADIUS, center=sensor, rmax=float(g["carve_crop_rmax"][0]))
    removed, gone = be.map_carve_removed(m, s, pose, crop, voxel=voxel, max_length=max_len, truncation=trunc, min_dot=min_dot)
    ids = g["carve_ids"].astype(np.int64)
    keep = np.ones(len(mp), bool)
    keep[ids] = False
    assert removed == len(ids) > 100
    gp, gn = be.download(m)
    assert np.array_equal(gp, mp[keep]) and np.array_equal(gn, mn[keep])
    assert np.array_equal(be.download(gone)[0], mp[ids])  # toRemove_, in map order
    for c in (m, s, gone):
        be.free(c)


def test_overlap_indices_equal_the_reference(backend_f64, g):
    """computeIndicesOfOverlappingPoints (helpers.cpp:307-332)"""
    be = backend_f64
    s, t = be.upload(_f64(g["carve_scan"])), be.upload(_f64(g["carve_map"]))
    voxel, min_points = g["overlap_params"]
    i_s, i_t = be.overlap_indices(s, t, g["tf_T"], voxel=float(voxel), min_points=int(min_points))
    assert np.array_equal(i_s.astype(np.int64), g["overlap_src"].astype(np.int64))
    assert np.array_equal(i_t.astype(np.int64), g["overlap_tgt"].astype(np.int64))
    be.free(s)
    be.free(t)


def test_dense_voxel_map_fuses_and_carves_like_the_reference(backend_f64, g):
    """VoxelizedPointCloud::insert x 3 + toPointCloud (Voxel.cpp:66-114), then Submap::carve for the dense map (Submap.cpp:126-136):
    the same voxels, the same counts (checked through the means of integer-count sums), means to the fixed-point grain; the same keys gone"""
    be = backend_f64
    pts, nrm = _f64(g["dense_pts"]), _f64(g["dense_nrm"])
    voxel = float(g["dense_voxel"][0])
    dm = be.dense_map_create(voxel)
    n = len(pts)
    for b in range(3):  # three scans, as the fixture was fused
        lo, hi = n * b // 3, n * (b + 1) // 3
        c = be.upload(pts[lo:hi], nrm[lo:hi])
        be.dense_map_insert(dm, c)
        be.free(c)
    out = be.dense_map_to_cloud(dm)
    gp, gn = be.download(out)
    assert len(gp) == len(g["dense_out_pts"])
    o = _key_order(gp, voxel)  # the fixture's order: x-major voxel keys
    gp, gn = gp[o], gn[o]
    keys = np.floor(gp * (1.0 / voxel)).astype(np.int64)
    assert np.array_equal(keys, g["dense_out_keys"].astype(np.int64))
    assert np.abs(gp - g["dense_out_pts"]).max() < 1e-8 and np.abs(gn - g["dense_out_nrm"]).max() < 1e-8
    # Submap::transform's dense part (VoxelizedPointCloud::transform, Voxel.cpp:49-64) on a copy of the map: keys stay, sums are moved as points
    dm2 = be.dense_map_create(voxel)
    c = be.upload(pts, nrm)
    be.dense_map_insert(dm2, c)
    be.free(c)
    before_id = be.dense_map_to_cloud(dm2)
    o2 = _key_order(be.download(before_id)[0], voxel)
    be.dense_map_transform(dm2, g["tf_T"])
    moved_id = be.dense_map_to_cloud(dm2)
    mp_, mn_ = be.download(moved_id)
    assert np.abs(mp_[o2] - g["dense_tf_out_pts"]).max() < 1e-7 and np.abs(mn_[o2] - g["dense_tf_out_nrm"]).max() < 1e-7
    for cid in (before_id, moved_id):
        be.free(cid)
    be.dense_map_free(dm2)
    radius, max_len, trunc = g["dense_carve_params"]
    s = be.upload(_f64(g["dense_carve_scan"]))
    removed = be.dense_map_carve(dm, s, np.zeros(3), radius=float(radius), max_length=float(max_len), truncation=float(trunc))
    want = set(map(tuple, g["dense_carve_keys"].astype(np.int64)))
    assert removed == len(want) > 100
    after = be.dense_map_to_cloud(dm)
    left = set(map(tuple, np.floor(be.download(after)[0] * (1.0 / voxel)).astype(np.int64)))
    assert left == set(map(tuple, keys)) - want
    for c in (out, s, after):
        be.free(c)
    be.dense_map_free(dm)


@pytest.mark.parametrize("clockwise", [0, 1])
def test_constant_velocity_deskew_agrees_with_the_reference(backend_f64, g, clockwise):
    """ConstantVelocityMotionCompensation::undistortInputPointCloud (MotionCompensation.cpp:64-139) at the velocity the reference estimated"""
    be = backend_f64
    c = be.upload(_f64(g["deskew_pts"]))
    be.undistort(c, g["deskew_vel"][:3], g["deskew_vel"][3:], float(g["deskew_scan_duration"][0]), bool(clockwise))
    got = be.download(c)[0]
    assert np.abs(got - g[f"deskew_out_{clockwise}"]).max() < 1e-8
    be.free(c)


def test_the_reference_s_own_conversion_tests_hold_on_the_device_path(backend_f32, backend_f64):
    """The only tests the reference ships that touch a row of the path: open3d_utils/open3d_conversions/test/
    test_open3d_conversions.cpp :31 open3dToRos_uncolored, :50 open3dToRos_colored, :77 rosToOpen3d_uncolored, :110 rosToOpen3d_colored --
    five points (0.5 i, i^2, 10.5 i), colours (2 i, 5 i, 10 i) / 255, exact EXPECT_EQ on float fields and uint8 colours.  Here the same
    vectors go through the device ingest / egress of row f4 (o3ds_cloud_upload_f32, o3ds_cloud_set_colors_from_records,
    o3ds_cloud_download_f32), which stand where rosToOpen3d / open3dToRos stand."""
    i = np.arange(5, dtype=np.float64)
    pts = np.stack([0.5 * i, i * i, 10.5 * i], axis=1)
    col = np.stack([2 * i / 255.0, 5 * i / 255.0, 10 * i / 255.0], axis=1)
    for be in (backend_f32, backend_f64):
        # open3dToRos, uncolored and colored: float32 x y z at 0 / 4 / 8; "rgb" packed b g r in the bytes of the float at 16, step 32
        c = be.upload(pts)
        rec = be.download_f32(c, point_step=16)
        assert np.array_equal(rec.view(np.float32).reshape(5, 4)[:, :3], pts.astype(np.float32))
        be.set_colors(c, col)
        rec = be.download_f32(c, point_step=32, off_rgb=16, rgb_rounding=0)
        assert np.array_equal(rec.view(np.float32).reshape(5, 8)[:, :3], pts.astype(np.float32))
        assert np.array_equal(rec[:, 18], (2 * i).astype(np.uint8)) and np.array_equal(rec[:, 17], (5 * i).astype(np.uint8))  # r, g
        assert np.array_equal(rec[:, 16], (10 * i).astype(np.uint8))  # b
        be.free(c)
        # rosToOpen3d, uncolored and colored
        wire = np.zeros((5, 32), dtype=np.uint8)
        wire[:, :12] = pts.astype(np.float32).view(np.uint8).reshape(5, 12)
        wire[:, 18], wire[:, 17], wire[:, 16] = (2 * i).astype(np.uint8), (5 * i).astype(np.uint8), (10 * i).astype(np.uint8)
        c = be.upload_f32(wire)
        got = be.download(c)[0]
        assert np.array_equal(got, pts) and not be.has_colors(c)  # every value of the vector is a float32
        be.set_colors_from_records(c, wire, 16, backend.COLOR_FIELD_RGB)
        assert be.has_colors(c)
        want = col if be is backend_f64 else col.astype(np.float32).astype(np.float64)  # f32 storage keeps the colour as a float32
        assert np.array_equal(be.get_colors(c), want)
        be.free(c)
