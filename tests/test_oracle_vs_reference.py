"""The CPU oracle (oracle/o3d_oracle.c) against open3d_slam's OWN sources, compiled unchanged from the reference checkout and RUN here
(oracle/ref_build -> oracle/_ref/libo3dslam_ref.so, bound by oracle/ref.py): croppers.cpp (row a8), helpers.cpp --
voxelizeWithinCroppingVolume (a9), o3d_slam::transform (a3), getIdxsOfCarvedPoints / getKeysOfCarvedPoints (f2),
computeIndicesOfOverlappingPoints (f3) --, Voxel.cpp / VoxelHashMap.cpp (the dense voxel map, f2), MotionCompensation.cpp + math.cpp +
Transform.cpp (de-skew, f4).

What this pins: the reference's own logic on those rows -- inclusive / exclusive comparisons, inversion, NaN behaviour, the voxel index, what
is accumulated in which order, what is normalised, what is emitted.  What it cannot pin: Open3D's algorithms (VoxelDownSample,
EstimateNormals, RegistrationICP) and Eigen's rounding, which are in neither /root/reference nor this image; the stand-in arithmetic
(oracle/ref_build/shim/Eigen/mini_eigen.hpp) is plain left-to-right double arithmetic, as the oracle's is.  Where the reference's output
order is its hash map's, both sides are brought into one canonical order (by voxel key) first.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from oracle import ref  # noqa: E402
from open3d_slam_amd import synthetic as syn  # noqa: E402

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref is neither built nor buildable (no reference checkout)")

KINDS = [(ref.CROP_NONE, po.CROP_NONE), (ref.CROP_MAX_RADIUS, po.CROP_MAX_RADIUS), (ref.CROP_MIN_RADIUS, po.CROP_MIN_RADIUS),
         (ref.CROP_MIN_MAX_RADIUS, po.CROP_MIN_MAX_RADIUS), (ref.CROP_CYLINDER, po.CROP_CYLINDER)]


def _pose(t):
    T = np.eye(4)
    th = 0.4
    T[:3, :3] = [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]  # the croppers only read the translation
    T[:3, 3] = t
    return T


def _key_order(p, voxel):
    k = np.floor(p * (1.0 / voxel)).astype(np.int64)
    return np.lexsort((k[:, 2], k[:, 1], k[:, 0]))


def test_reference_units_build_from_the_checkout_and_load():
    lib = ref.build()
    assert lib and os.path.isfile(lib)
    L = ref.lib()
    for name in ("ref_crop", "ref_voxelize_within_cropping_volume", "ref_transform", "ref_carved_idxs", "ref_overlap", "ref_dense_fuse",
                 "ref_dense_carve_keys", "ref_undistort", "ref_voxel_idx"):
        assert hasattr(L, name)
    assert ref.lib().ref_icp_max_correspondence_distance(0.1) == 2.0 * 0.1 and ref.lib().ref_icp_max_correspondence_distance(0.0005) == 0.05
    assert ref.lib().ref_information_matrix_max_correspondence_distance(0.2) == 1.5 * 0.2


@pytest.mark.parametrize("invert", [False, True])
@pytest.mark.parametrize("rk,ok", KINDS)
def test_croppers_keep_the_same_points_as_the_reference(rk, ok, invert):
    """croppers.cpp:53-61,121-165 incl. boundary values, inversion, and no-return points (every comparison with NaN is false)"""
    rng = np.random.default_rng(5 + rk)
    n = 20000
    pts = rng.normal(size=(n, 3)) * [12, 12, 3]
    pts[::97] = np.nan
    pts[5::131, 1] = np.inf
    pts[7::173, 2] = -np.inf
    t = np.array([1.5, -2.0, 0.3])
    pts[1] = t + [10.0, 0, 0]  # exactly on the outer radius
    pts[2] = t + [0, 2.0, 0]  # exactly on the inner radius
    pts[3] = [t[0] + 3.0, t[1], 2.0]  # exactly on the cylinder's upper z (absolute z, croppers.cpp:161)
    pts[4] = [t[0] + 3.0, t[1], -1.0]
    nrm = rng.normal(size=(n, 3))
    col = rng.uniform(size=(n, 3))
    idx, p, nn, cc = ref.crop(pts, rk, rmin=2.0, rmax=10.0, zmin=-1.0, zmax=2.0, pose=_pose(t), invert=invert, nrm=nrm, col=col)
    c = po.make_crop(ok, center=t, rmin=2.0, rmax=10.0, zmin=-1.0, zmax=2.0, invert=invert)
    assert np.array_equal(po.crop_indices(pts, c), idx)
    assert np.array_equal(p, pts[idx], equal_nan=True) and np.array_equal(nn, nrm[idx]) and np.array_equal(cc, col[idx])
    if rk != ref.CROP_NONE and not invert:
        assert {1, 2}.issubset(set(idx.tolist())) or rk in (ref.CROP_CYLINDER,)  # the radii are inclusive
    if rk == ref.CROP_CYLINDER and not invert:
        assert {3, 4}.issubset(set(idx.tolist()))


@pytest.mark.parametrize("voxel", [0.1, 0.25])
@pytest.mark.parametrize("rk,ok", KINDS[1:])
def test_voxelize_within_cropping_volume_equals_the_reference_bit_for_bit(rk, ok, voxel):
    """helpers.cpp:115-183 + AccumulatedPoint (:30-75): pass-through points first and unchanged, per-voxel sums in input order, NaN
    normals skipped but counted, normals re-normalised, a voxel keeps the colour of its last valid-colour point.  Voxel part compared
    in key order (the reference's is its hash map's)."""
    rng = np.random.default_rng(7)
    n = 30000
    pts = rng.normal(size=(n, 3)) * [8, 8, 2]
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm[::53] = np.nan
    col = rng.uniform(size=(n, 3))
    t = np.array([0.5, -1.0, 0.1])
    c = po.make_crop(ok, center=t, rmin=1.0, rmax=9.0, zmin=-1.5, zmax=1.5)
    op, on, npass = po.voxelize_within_volume(pts, nrm, voxel, c)
    oc = po.voxelize_within_volume_colors(pts, col, voxel, c)
    rp, rn, rc = ref.voxelize_within_cropping_volume(pts, voxel, rk, rmin=1.0, rmax=9.0, zmin=-1.5, zmax=1.5, pose=_pose(t), nrm=nrm, col=col)
    assert len(op) == len(rp) and 0 < npass < len(op)
    for a, b in ((op, rp), (on, rn), (oc, rc)):
        assert np.array_equal(a[:npass], b[:npass], equal_nan=True)  # pass-through: same points, same (input) order
    oo, ro = _key_order(op[npass:], voxel), _key_order(rp[npass:], voxel)
    for a, b in ((op, rp), (on, rn), (oc, rc)):
        assert np.array_equal(a[npass:][oo], b[npass:][ro], equal_nan=True)


def test_is_valid_color_is_the_reference_s_always_true_predicate():
    """helpers.cpp:83-85 compares the BOOLEAN `c.array().all()` with 0.0 and 1.0: true for every colour, also out of range or NaN --
    which is why the merge lets every colour through (oracle: the last point of a voxel wins)"""
    for c in ([0.2, 0.3, 0.4], [1.5, -2.0, 0.1], [0.0, 0.0, 0.0], [np.nan, 0.1, 0.2]):
        assert ref.lib().ref_is_valid_color(np.array(c, dtype=np.float64).ctypes.data_as(ref._dp)) == 1
    rng = np.random.default_rng(1)
    pts = rng.uniform(-1, 1, size=(4000, 3))
    col = rng.uniform(-1, 2, size=(4000, 3))
    c = po.make_crop(po.CROP_MAX_RADIUS, rmax=5.0)
    op, _, npass = po.voxelize_within_volume(pts, None, 0.3, c)
    oc = po.voxelize_within_volume_colors(pts, col, 0.3, c)
    rp, _, rc = ref.voxelize_within_cropping_volume(pts, 0.3, ref.CROP_MAX_RADIUS, rmax=5.0, col=col)
    assert npass == 0 and np.array_equal(oc[_key_order(op, 0.3)], rc[_key_order(rp, 0.3)])


def test_transform_equals_the_reference_and_its_identity_quirk_is_documented():
    """helpers.cpp:273-305.  Away from the identity: p' = (T p).xyz / w, n' = R n, bit for bit.  Within 1e-4 of the identity the
    reference returns the input FOLLOWED by the transformed copies (2 n points, 2 n normals, n colours -- SURVEY.md B4), which this
    backend deliberately does not reproduce; the test shows what SURVEY claims for the case that occurs (the exact identity of the first
    scan): after voxelizeWithinCroppingVolume the duplicated insert and the plain one are the same map (same voxels, means equal to a few ulp)."""
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(5000, 3)) * [6, 6, 1.5]
    nrm = rng.normal(size=(5000, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    A = _pose([1.0, 2.0, 3.0])
    tp, tn, _ = ref.transform(A, pts, nrm)
    assert np.array_equal(tp, po.transform_points(pts, A)) and np.array_equal(tn, po.transform_normals(nrm, A))
    near = np.eye(4)
    near[0, 3] = 5e-5
    dp, dn, has_col = ref.transform(near, pts, nrm, col=rng.uniform(size=(5000, 3)))
    assert len(dp) == 2 * len(pts) and np.array_equal(dp[: len(pts)], pts) and np.array_equal(dp[len(pts):], po.transform_points(pts, near))
    assert not has_col
    c = po.make_crop(po.CROP_MAX_RADIUS, rmax=30.0)
    # (for a T that is near but not equal to the identity the two copies are up to 1e-4 apart and a few of them land in neighbouring
    # voxels: the reference's map then holds a handful of extra voxels; the case that does occur is the exact identity of the first scan)
    a_p, _, _ = po.voxelize_within_volume(dp, dn, 0.1, c)
    b_p, _, _ = po.voxelize_within_volume(po.transform_points(pts, near), po.transform_normals(nrm, near), 0.1, c)
    assert 0 <= len(a_p) - len(b_p) <= 0.002 * len(b_p)
    exact = np.eye(4)
    dp, dn, _ = ref.transform(exact, pts, nrm)
    a_p, a_n, _ = po.voxelize_within_volume(dp, dn, 0.1, c)
    b_p, b_n, _ = po.voxelize_within_volume(pts, nrm, 0.1, c)
    oa, ob = _key_order(a_p, 0.1), _key_order(b_p, 0.1)
    assert len(a_p) == len(b_p) and np.abs(a_p[oa] - b_p[ob]).max() <= 4e-15 and np.abs(a_n[oa] - b_n[ob]).max() <= 4e-16


def _carving_case():
    scene = syn.make_scene()
    poses = syn.figure_eight_poses(20, 0.1)
    mp, mn = syn.sample_map(scene, 60000, seed=3)
    T = poses[3]
    scan = syn.vlp16_scan(scene, T)[:6000]
    scan = scan @ T[:3, :3].T + T[:3, 3]
    sensor = T[:3, 3].copy()
    # things that are no longer there: map points strewn along the scan's rays, with normals facing and not facing the sensor
    rng = np.random.default_rng(9)
    pick = rng.choice(len(scan), 1500, replace=False)
    frac = rng.uniform(0.2, 0.9, size=(1500, 1))
    ghosts = sensor + (scan[pick] - sensor) * frac + rng.normal(scale=0.02, size=(1500, 3))
    gn = rng.normal(size=(1500, 3))
    gn /= np.linalg.norm(gn, axis=1, keepdims=True)
    gn[::40] = 0.0  # normalized() of a zero vector stays zero: |direction . n| = 0
    return scan, sensor, np.vstack([mp, ghosts]), np.vstack([mn, gn])


@pytest.mark.parametrize("min_dot,with_normals,subset", [(0.5, True, True), (0.0, True, True), (0.9, True, False), (0.5, False, True)])
def test_space_carving_removes_the_same_points_as_the_reference(min_dot, with_normals, subset):
    """getIdxsOfCarvedPoints (helpers.cpp:221-271) over VoxelMap (Voxel.cpp:121-146): ray samples every voxel, path limit
    max(voxel, min(length - truncation, max length)), a point goes if |direction . normalised normal| > min dot (strict)"""
    scan, sensor, mp, mn = _carving_case()
    sub = np.flatnonzero(np.linalg.norm(mp - sensor, axis=1) < 25.0) if subset else np.arange(len(mp))
    nr = mn if with_normals else None
    f = po.carve_flags(scan, sensor, mp, nr, sub, voxel=0.2, max_length=15.0, truncation=0.3, min_dot=min_dot)
    ids = ref.carved_idxs(scan, sensor, mp, nr, sub if subset else None, voxel=0.2, max_length=15.0, truncation=0.3, min_dot=min_dot)
    assert f.sum() > 100
    assert np.array_equal(np.flatnonzero(f), ids)


@pytest.mark.parametrize("voxel,min_points", [(0.5, 1), (1.0, 3), (0.3, 2)])
def test_overlap_indices_equal_the_reference(voxel, min_points):
    """computeIndicesOfOverlappingPoints (helpers.cpp:307-332); the source is placed by [O3D] PointCloud::Transform (restated in the stand-in)"""
    scene = syn.make_scene()
    poses = syn.figure_eight_poses(20, 0.1)
    mp, _ = syn.sample_map(scene, 60000, seed=3)
    src = syn.vlp16_scan(scene, poses[5])[:20000]
    a = po.overlap_indices(src, mp, poses[5], voxel, min_points)
    b = ref.overlap_indices(src, mp, poses[5], voxel, min_points)
    assert len(a[0]) > 100 and len(a[1]) > 100
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("voxel", [0.05, 0.2])
def test_dense_voxel_map_equals_the_reference_bit_for_bit(voxel):
    """VoxelizedPointCloud::insert x 3 scans + toPointCloud (Voxel.cpp:66-114): running sums in insertion order, mean = sum / count,
    the mean normal NOT re-normalised"""
    rng = np.random.default_rng(11)
    pts = rng.normal(size=(50000, 3)) * [3, 3, 1]
    nrm = rng.normal(size=(50000, 3))
    op, on, oc = po.dense_fuse(pts, nrm, voxel)
    rp, rn, rc, rk = ref.dense_fuse(pts, nrm, voxel, batches=3)
    assert len(op) == len(rp)
    ko = np.array([ref.voxel_idx(p, voxel) for p in op[:200]])  # the mean of a voxel lies in it: its key by the reference's own getVoxelIdx
    assert np.array_equal(ko, np.floor(op[:200] * (1.0 / voxel)).astype(np.int32))
    oo = _key_order(op, voxel)
    ro = np.lexsort((rk[:, 2], rk[:, 1], rk[:, 0]))
    assert np.array_equal(op[oo], rp[ro]) and np.array_equal(on[oo], rn[ro]) and np.array_equal(oc[oo], rc[ro])


def test_dense_map_carving_removes_the_same_voxels_as_the_reference():
    """Submap::carve for the dense map (Submap.cpp:126-136): removeDuplicatePointsWithinSameVoxels (Voxel.cpp:162-191), then
    getKeysOfCarvedPoints (helpers.cpp:347-377) with getVoxelsWithinPointNeighborhood (VoxelHashMap.cpp:13-44)"""
    scan, sensor, mp, _ = _carving_case()
    voxel = 0.1
    dm, _, _ = po.dense_fuse(mp, None, voxel)
    rem = po.dense_carve(scan, sensor, dm, voxel, radius=0.1, max_length=15.0, truncation=0.3)
    keys = ref.dense_carve_keys(scan, sensor, dm, voxel, radius=0.1, max_length=15.0, truncation=0.3, dedup_scan=True)
    a = set(map(tuple, np.floor(dm[rem] * (1.0 / voxel)).astype(np.int64)))
    b = set(map(tuple, keys.astype(np.int64)))
    assert len(a) > 100 and a == b


@pytest.mark.parametrize("clockwise", [False, True])
def test_constant_velocity_deskew_agrees_with_the_reference(clockwise):
    """ConstantVelocityMotionCompensation (MotionCompensation.cpp:27-139) with math.cpp's fromRPY / toRPY and Transform.cpp's makeTransform
    run from the reference; both sides are given the velocity the REFERENCE estimates from its pose buffer.  Tolerance, not bits: the
    motion goes through a quaternion (AngleAxis product, normalisation, rotation matrix) whose rounding is Eigen's -- the stand-in's and
    the oracle's sequences differ in the last place"""
    scene = syn.make_scene()
    p = syn.vlp16_scan(scene, syn.figure_eight_poses(20, 0.1)[2])[:5000]
    out, vel = ref.undistort(p, [0.12, -0.03, 0.01], [0.01, -0.02, 0.15], 0.1, 0.1, clockwise)
    assert np.allclose(vel[:3], np.array([0.12, -0.03, 0.01]) / (0.1 + 1e-6), rtol=0, atol=1e-9)  # linear velocity = translation / (dt + 1e-6)
    assert np.allclose(vel[3:], np.array([0.01, -0.02, 0.15]) / (0.1 + 1e-6), rtol=0, atol=1e-9)
    mine = po.undistort(p, vel[:3], vel[3:], 0.1, clockwise)
    assert np.abs(mine - out).max() <= 2e-14
    assert np.abs(out - p).max() > 1e-3  # it did move something


@pytest.mark.parametrize("carve_every", [3, 10])
def test_the_reference_s_own_frame_loop_agrees_with_the_oracle_loop(carve_every):
    """Row a10, the glue: LidarOdometry::addRangeScan (Odometry.cpp:32-79) and Mapper::addRangeMeasurement (Mapper.cpp:101-181) with
    ScanToMapIcp (ScanToMapRegistration.cpp:35-62), SubmapCollection::insertScan (SubmapCollection.cpp:172-207) and Submap::insertScan /
    carve (Submap.cpp:39-125) -- the reference's code, compiled unchanged and run scan by scan -- against oracle/pipeline.py, the loop
    the GPU stream is held to (tests/test_pipeline_gpu.py).  The Open3D calls underneath are served by the same oracle on both sides, so
    what is compared is the orchestration: which volume crops what, the odometry prior of the scan matcher, the map patch it registers
    against, the fitness gates, when a scan is inserted and when the map is carved (always asked for; every carveSpaceEveryNscans_-th
    insertion counted from the second).  Same map size after every frame, poses within 1e-10 m / rad (they differ by the rounding of a
    4x4 inverse and of the first insertion's duplicated points, SURVEY B4, amplified by the registrations), same final map."""
    import bench
    from oracle.pipeline import OracleLoop

    mp, op = bench.stream_parameters()
    mp.mapBuilder_.carving_.carveSpaceEveryNscans_ = carve_every
    scene = syn.make_scene()
    poses = syn.figure_eight_poses(200, 0.1)
    frames = 13 if carve_every == 10 else 8
    scans = [np.asarray(syn.os128_scan(scene, poses[k], frame=k), dtype=np.float64)[::8] for k in range(frames)]
    R = ref.ReferenceSlam(mp, op, carve_every_n_scans=carve_every)
    O = OracleLoop(po, mp, op)
    worst = 0.0
    for k, s in enumerate(scans):
        rc, odom, T, n_map, n_sub = R.add_scan(s, 0.1 * k)
        O.odometry(s, k)
        O.mapping(s, k)
        assert rc == 1 and n_sub == 1
        assert n_map == len(O.map_p), (k, n_map, len(O.map_p))
        worst = max(worst, *syn.se3_error(T, O.T), *syn.se3_error(odom, O.odom))
    assert worst < 1e-10, worst
    assert O.n_carved > 0  # the carving did act
    rp, rn = R.map()
    a, b = _key_order(rp, 0.1), _key_order(O.map_p, 0.1)
    assert np.abs(rp[a] - O.map_p[b]).max() < 1e-10
    well = np.abs(rn[a] - O.map_n[b]).max(axis=1) < 1e-9  # a voxel whose normals nearly cancel amplifies the last bits of the pose
    assert well.mean() > 0.995
    # the scan the reference's mapper matched last = crop + VoxelDownSample + normals + narrow crop of the oracle's chain, bit for bit
    mpts, mnrm = R.preprocessed_scan()
    icp = mp.scanMatcher_.icp_
    v, n = O._pre(scans[-1], mp.mapBuilder_.cropper_, mp.scanProcessing_.voxelSize_, icp)
    sc = mp.scanProcessing_.cropper_
    keep = po.crop_indices(v, po.make_crop(po.CROP_MIN_MAX_RADIUS, rmin=sc.croppingMinRadius_, rmax=sc.croppingMaxRadius_))
    assert np.array_equal(mpts, v[keep]) and np.array_equal(mnrm, n[keep])
    R.close()


def test_submap_dense_map_insertion_and_carving_follow_the_reference():
    """Submap::insertScanDenseMap (Submap.cpp:77-92) run from the reference: crop the raw scan in the SENSOR frame (the cropper is put at
    the identity), place it with o3d_slam::transform, fuse; carving asked for on every scan and performed when nScansInsertedDenseMap_ %
    carveSpaceEveryNscans_ == 1, with the RAW (sensor-frame) scan against the map-frame sensor position (Submap.cpp:88 -- as written).
    The oracle's dense_fuse / dense_carve, called in that order with those arguments (what tests/test_pipeline_gpu.py holds the device
    mirror to), give the same voxels and the same means."""
    scene = syn.make_scene()
    poses = syn.figure_eight_poses(200, 0.1)[10:14]
    scans = [np.asarray(syn.vlp16_scan(scene, T), dtype=np.float64)[::4] for T in poses]
    voxel, rmax, every = 0.1, 15.0, 2
    rp, rk, rc, sizes = ref.submap_dense(scans, poses, voxel=voxel, crop_rmax=rmax, carve_every=every)
    # the same sequence on the oracle: a dict voxel key -> (sum, count)
    acc = {}
    n_inserted = 0
    mine_sizes = []
    for raw, T in zip(scans, poses):
        inside = raw[np.linalg.norm(raw, axis=1) <= rmax]
        placed = po.transform_points(inside, T)
        for p in placed:
            k = tuple(np.floor(p * (1.0 / voxel)).astype(np.int64))
            s, c = acc.get(k, (np.zeros(3), 0))
            acc[k] = (s + p, c + 1)
        if acc and n_inserted % every == 1:
            keys = list(acc.keys())
            means = np.array([acc[k][0] / acc[k][1] for k in keys])
            gone = po.dense_carve(raw, T[:3, 3], means, voxel, radius=0.1, max_length=20.0, truncation=0.1)
            for k, g in zip(keys, gone):
                if g:
                    del acc[k]
        n_inserted += 1
        mine_sizes.append(len(acc))
    assert mine_sizes == sizes and sizes[1] < sizes[0] + len(scans[1])  # the carving did remove voxels at the second scan
    keys = sorted(acc.keys())
    assert np.array_equal(np.array(keys, dtype=np.int64), rk.astype(np.int64))
    means = np.array([acc[k][0] / acc[k][1] for k in keys])
    assert np.array_equal(means, rp) and np.array_equal(np.array([acc[k][1] for k in keys]), rc)


def test_the_mapper_s_gates_follow_the_reference():
    """Mapper.cpp:151-156 (a scan whose refinement fitness is below minRefinementFitness_ changes nothing: no pose, no insertion, and the next
    prior spans the odometry motion since the last ACCEPTED scan) and Mapper.cpp:170-176 (no insertion while the sensor has moved less than
    minMovementBetweenMappingSteps_ since the last one; the pose is still refined), plus Odometry.cpp:52-67 (a scan the odometry cannot
    register -- fitness <= 0.1 -- still replaces the scan to match against).  One stream with a corrupted scan in it, through the reference's
    own loop and through the oracle loop in non-strict mode: same verdict for every scan, same map sizes, same poses."""
    import bench
    from oracle.pipeline import OracleLoop

    mp, op = bench.stream_parameters()
    mp.minMovementBetweenMappingSteps_ = 0.25  # the sensor moves ~0.14 m per frame: every second scan is inserted
    mp.scanMatcher_.minRefinementFitness_ = 0.8
    scene = syn.make_scene()
    poses = syn.figure_eight_poses(200, 0.1)
    scans = [np.asarray(syn.os128_scan(scene, poses[k], frame=k), dtype=np.float64)[::8] for k in range(9)]
    bad = scans[4].copy()  # half of the scan lifted into the air: the odometry still finds enough of it (> 0.1), the mapper does not (< 0.8)
    bad[::2] += [0.0, 0.0, 6.0]
    scans[4] = bad
    rng = np.random.default_rng(2)
    scans[7] = rng.uniform(-15.0, 15.0, size=scans[7].shape) * [1, 1, 0.1] + [0, 0, 20.0]  # a slab 20 m up, nothing of the scene: the odometry refuses it
    R = ref.ReferenceSlam(mp, op, min_movement=mp.minMovementBetweenMappingSteps_)
    O = OracleLoop(po, mp, op)
    O.strict = False
    verdicts = []
    for k, s in enumerate(scans):
        rc, odom, T, n_map, n_sub = R.add_scan(s, 0.1 * k)
        ok_o = O.odometry(s, k)
        ok_m = O.mapping(s, k) if ok_o else None
        mine = 1 if (ok_o and ok_m) else (0 if not ok_o else -1)
        verdicts.append((rc, mine))
        assert rc == mine, (k, verdicts)
        assert n_map == len(O.map_p), (k, n_map, len(O.map_p))
        if k > 0:
            assert max(*syn.se3_error(T, O.T)) < 1e-9, (k, syn.se3_error(T, O.T))
    assert [v[0] for v in verdicts].count(-1) >= 1 and [v[0] for v in verdicts].count(0) >= 1, verdicts  # both rejections did happen
    assert O.n_not_inserted >= 2 and O.n_rejected >= 1
    R.close()


def test_the_eigen_stand_in_checks_itself(tmp_path):
    """tests/cpp/test_mini_eigen.cpp: the pieces of the stand-in whose silent failure would make a reference run meaningless (write-back of
    `T.matrix() *= M`, inverses, quaternions, the voxel-index expressions), compiled warning-free and run"""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "test_mini_eigen")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-Wall", "-Wextra", "-Werror", "-o", exe,
                        os.path.join(root, "tests", "cpp", "test_mini_eigen.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def _golden_with_flags(tmp_path, tag, flags):
    """the reference's units rebuilt with other compiler flags into a scratch directory, the golden generator run against that library"""
    import subprocess
    import sys

    out = tmp_path / tag
    out.mkdir()
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    r = subprocess.run(["make", "-j", "8", "-C", os.path.join(root, "oracle", "ref_build"), f"REF={ref.REFERENCE}", f"OUT={out}", f"CXXFLAGS={flags}",
                        f"{out}/libo3dslam_ref.so"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    npz = out / "units.npz"
    env = dict(os.environ, O3DS_REF_LIB=str(out / "libo3dslam_ref.so"), O3DS_REF_GOLDEN_OUT=str(npz),
               LD_LIBRARY_PATH=os.path.join(root, "oracle") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))  # (libo3d_oracle.so: the scratch library's rpath points elsewhere)
    r = subprocess.run([sys.executable, os.path.join(here, "golden", "make_ref_golden.py")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(npz)


@pytest.mark.skipif(not ref.sources_present(), reason="needs the reference checkout")
def test_reference_fixtures_do_not_depend_on_the_optimisation_level_and_what_fma_would_move(tmp_path):
    """The committed fixtures (tests/golden/ref_units.npz) come from the reference's units built -O2 -ffp-contract=off.  The reference
    itself builds -O3 without -march (open3d_slam/CMakeLists.txt:6): rebuilt that way, every array is IDENTICAL.  Rebuilt with fused
    multiply-add allowed (-O3 -mfma -ffp-contract=fast, which the reference's build does not do), what moves is reported: only
    floating-point values (voxel means, transformed points), by rounding; every index, count, key and kept / carved set stays put.  This
    bounds the caveat of the stand-in Eigen (oracle/ref_build/README.md): the bit-for-bit claims concern comparisons, index formation,
    ordering and NaN handling, and those do not move with the flags."""
    base = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_units.npz"))
    shim = "-std=c++17 -fPIC -w -Ishim -I" + os.path.join(ref.REFERENCE, "open3d_slam", "open3d_slam", "include")
    o3 = _golden_with_flags(tmp_path, "o3", "-O3 " + shim)
    assert sorted(o3.files) == sorted(base.files)
    for k in base.files:
        np.testing.assert_array_equal(o3[k], base[k], err_msg=k)
    fma = _golden_with_flags(tmp_path, "fma", "-O3 -mfma -ffp-contract=fast " + shim)
    moved = []
    for k in base.files:
        a, b = base[k], fma[k]
        assert a.shape == b.shape, k  # no point changes side of a volume, no voxel gains or loses a member
        if not np.array_equal(a, b, equal_nan=True):
            assert np.issubdtype(a.dtype, np.floating), k  # integers (indices, keys, counts) never move
            with np.errstate(invalid="ignore"):
                err = np.nanmax(np.abs(a - b) / np.maximum(np.abs(a), 1e-300))
            moved.append((k, float(err)))
            assert err < 1e-13, (k, err)
    print("arrays that move when fused multiply-add is allowed:", moved)
