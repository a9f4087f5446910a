"""Host mirror of the reference seams (no GPU needed): factories, parameter copy, predicates, error behaviour."""
import numpy as np
import pytest

from open3d_slam_amd import backend, croppers, parameters as P
from open3d_slam_amd.cloud_registration import (RegistrationIcpPointToPlane, cloudRegistrationFactory, createPointToPlaneIcp)
from open3d_slam_amd.scan_to_map_registration import ScanToMapIcp, scanToMapRegistrationFactory, toCloudRegistrationType


def test_parameter_defaults_match_reference_structs():
    icp = P.IcpParameters()  # Parameters.hpp:66-71
    assert (icp.maxNumIter_, icp.maxCorrespondenceDistance_, icp.knn_, icp.maxDistanceKnn_) == (50, 0.2, 5, 10.0)
    c = P.ScanCroppingParameters()  # Parameters.hpp:51-57
    assert (c.croppingMinZ_, c.croppingMaxZ_, c.croppingMinRadius_, c.croppingMaxRadius_, c.cropperName_) == (-10.0, 10.0, 0.0, 20.0, "MaxRadius")
    assert P.ScanProcessingParameters().voxelSize_ == 0.03 and P.MapBuilderParameters().mapVoxelSize_ == 0.03
    assert P.ScanToMapRegistrationParameters().minRefinementFitness_ == 0.7
    assert P.CloudRegistrationParameters().regType_ == P.CloudRegistrationType.PointToPlaneIcp
    lua = P.lua_default_mapper_parameters()
    assert lua.scanMatcher_.icp_.knn_ == 20 and lua.scanMatcher_.icp_.maxDistanceKnn_ == 3.0
    assert lua.scanProcessing_.cropper_.cropperName_ == "MinMaxRadius"


def test_dense_map_builder_parameters_default_like_the_reference():
    """MapperParameters::denseMapBuilder_ / isBuildDenseMap_ (Parameters.hpp:164-165) and SpaceCarvingParameters (:85-92)."""
    from open3d_slam_amd import parameters as P

    m = P.MapperParameters()
    assert m.isBuildDenseMap_ is True
    assert m.denseMapBuilder_.mapVoxelSize_ == 0.03 and m.denseMapBuilder_.cropper_.cropperName_ == "MaxRadius"
    c = m.denseMapBuilder_.carving_
    assert (c.voxelSize_, c.maxRaytracingLength_, c.truncationDistance_, c.carveSpaceEveryNscans_, c.minDotProductWithNormal_,
            c.neighborhoodRadiusDenseMap_) == (0.1, 20.0, 0.1, 10, 0.5, 0.1)
    assert m.denseMapBuilder_ is not m.mapBuilder_  # independent structs, as in the reference


def test_cloud_registration_factory_copies_parameters():
    p = P.CloudRegistrationParameters()
    p.icp_ = P.IcpParameters(maxNumIter_=17, maxCorrespondenceDistance_=0.7, knn_=9, maxDistanceKnn_=1.5)
    reg = cloudRegistrationFactory(p)
    assert isinstance(reg, RegistrationIcpPointToPlane)
    assert (reg.maxCorrespondenceDistance_, reg.knnNormalEstimation_, reg.maxRadiusNormalEstimation_) == (0.7, 9, 1.5)
    c = reg.icpConvergenceCriteria_
    assert (c.max_iteration_, c.relative_fitness_, c.relative_rmse_) == (17, 1e-6, 1e-6)  # only max_iteration_ is overridden
    assert createPointToPlaneIcp(p).maxCorrespondenceDistance_ == 0.7
    p.regType_ = P.CloudRegistrationType.GeneralizedIcp
    g = cloudRegistrationFactory(p)
    assert type(g).__name__ == "RegistrationIcpGeneralized" and g.maxCorrespondenceDistance_ == 0.7 and g.icpConvergenceCriteria_.max_iteration_ == 17
    p.regType_ = P.CloudRegistrationType.PointToPointIcp
    pp = cloudRegistrationFactory(p)  # CloudRegistration.cpp:76-81: only the distance and max_iteration_ are taken over
    assert type(pp).__name__ == "RegistrationIcpPointToPoint" and pp.maxCorrespondenceDistance_ == 0.7 and pp.icpConvergenceCriteria_.max_iteration_ == 17
    p.regType_ = 42
    with pytest.raises(RuntimeError, match="unknown type"):
        cloudRegistrationFactory(p)


def test_normal_estimation_parameter_asserts():
    reg = RegistrationIcpPointToPlane()
    reg.maxRadiusNormalEstimation_ = 0.0
    with pytest.raises(RuntimeError, match="maxRadiusNormalEstimation_"):
        reg.estimateNormalsOrCovariancesIfNeeded(None)
    reg.maxRadiusNormalEstimation_ = 1.0
    reg.knnNormalEstimation_ = 0
    with pytest.raises(RuntimeError, match="knnNormalEstimation_"):
        reg.estimateNormalsOrCovariancesIfNeeded(None)


def test_cropper_factory_and_predicates(oracle):
    sp = P.ScanCroppingParameters(croppingMinZ_=-1.0, croppingMaxZ_=5.0, croppingMinRadius_=2.0, croppingMaxRadius_=30.0)
    pts = np.array([[2.0, 0, 0], [1.9999999, 0, 0], [30.0, 0, 0], [30.0000001, 0, 0], [0, 0, 5.0], [1.0, 1.0, 6.0]])
    kinds = {"MaxRadius": oracle.CROP_MAX_RADIUS, "MinRadius": oracle.CROP_MIN_RADIUS, "MinMaxRadius": oracle.CROP_MIN_MAX_RADIUS,
             "Cylinder": oracle.CROP_CYLINDER}
    pose = np.eye(4)
    pose[:3, 3] = [0.5, -0.25, 0.1]
    for name, ok in kinds.items():
        sp.cropperName_ = name
        cr = croppers.croppingVolumeFactory(sp)
        cr.setPose(pose)
        for inv in (False, True):
            cr.setIsInvertVolume(inv)
            abi = cr.to_abi()
            assert abi.kind == ok and bool(abi.invert) == inv and list(abi.center) == [0.5, -0.25, 0.1]
            oc = oracle.make_crop(ok, center=pose[:3, 3], rmin=abi.rmin, rmax=abi.rmax, zmin=abi.zmin, zmax=abi.zmax, invert=inv)
            keep = set(oracle.crop_indices(pts, oc).tolist())
            assert {i for i, p in enumerate(pts) if cr.isWithinVolume(p)} == keep, (name, inv)
    sp.cropperName_ = "NoSuchCropper"
    with pytest.raises(RuntimeError):
        croppers.croppingVolumeFactory(sp)
    base = croppers.CroppingVolume()
    assert base.isWithinVolume([1e9, 0, 0]) and base.to_abi().kind == backend.CROP_NONE


def test_scan_to_map_factory_and_type_mapping():
    p = P.lua_default_mapper_parameters()
    s2m = scanToMapRegistrationFactory(p)
    assert isinstance(s2m, ScanToMapIcp)
    assert s2m.cloudRegistration.maxCorrespondenceDistance_ == 1.0 and s2m.cloudRegistration.knnNormalEstimation_ == 20
    assert s2m.mapBuilderCropper_.radiusMin_ == 2.0 and s2m.scanMatcherCropper_.radiusMax_ == 30.0
    cr = toCloudRegistrationType(p.scanMatcher_)
    assert cr.regType_ == P.CloudRegistrationType.PointToPlaneIcp and cr.icp_ is p.scanMatcher_.icp_
    p.scanMatcher_.scanToMapRegType_ = 42
    with pytest.raises(RuntimeError):
        scanToMapRegistrationFactory(p)


def test_split_exact_makes_f64_sums_order_independent():
    """The arithmetic behind the order-independent record sums (icp_kernels.hpp: split_exact), emulated in numpy float64: values split
    into hi + lo with hi a multiple of q and lo a multiple of q * 2^-41 add up EXACTLY, so any summation order gives the same bits,
    and hi + lo reproduces v to within q * 2^-42."""
    import numpy as np

    rng = np.random.default_rng(0)
    n = 4096
    v = rng.normal(size=n) * np.exp(rng.uniform(-20, 5, size=n))  # many magnitudes, both signs
    bound = np.abs(v).sum()
    q = 2.0 ** (int(np.ceil(np.log2(bound))) + 3 - 53)  # 2^53 q >= 8 * bound

    def split(x):
        c_hi = 6755399441055744.0 * q
        h = (x + c_hi) - c_hi
        r = x - h
        c_lo = c_hi * 2.0 ** -41
        return h, (r + c_lo) - c_lo

    hi, lo = split(v)
    assert np.all(np.abs(v - (hi + lo)) <= q * 2.0 ** -42)
    sums = set()
    for _ in range(20):
        perm = rng.permutation(n)
        sh = 0.0
        sl = 0.0
        for k in perm:  # strictly sequential f64 adds in a random order
            sh += hi[k]
            sl += lo[k]
        sums.add((sh, sl))
    assert len(sums) == 1  # bit-identical whatever the order
    sh, sl = next(iter(sums))
    import math

    assert abs((sh + sl) - math.fsum(v)) <= n * q * 2.0 ** -42 + 2.0 ** -52 * abs(math.fsum(v))
    # plain f64 summation of the same values is NOT order-independent (the reason for the split)
    plain = {float(np.add.reduce(v[rng.permutation(n)])) for _ in range(20)}
    assert len(plain) > 1


def test_pcd_writer_header_and_round_trip(tmp_path):
    """saveToFile (output.cpp:39-47) on the host side alone: a stand-in backend that packs the rows with numpy (the statement of
    what o3ds_cloud_download_f32 must produce) -> header known answer, '.pcd' suffix rule, readPcd round trip."""
    import numpy as np
    from open3d_slam_amd import output

    rng = np.random.default_rng(5)
    pts, nrm = rng.normal(size=(37, 3)) * 20, rng.normal(size=(37, 3))

    class _Be:
        def __init__(self, normals):
            self.normals = normals

        def size(self, cid):
            return len(pts), self.normals is not None

        def has_colors(self, cid):
            return False

        def download_f32(self, cid, step, ox, oy, oz, on, orgb=None, rounding=0):
            rows = np.zeros((len(pts), step), np.uint8)
            for col, off in enumerate((ox, oy, oz)):
                rows[:, off:off + 4] = pts[:, col].astype("<f4").reshape(-1, 1).view(np.uint8)
            if on is not None:
                for col in range(3):
                    rows[:, on + 4 * col:on + 4 * col + 4] = self.normals[:, col].astype("<f4").reshape(-1, 1).view(np.uint8)
            return rows

    class _Cloud:
        def __init__(self, be):
            self.be, self.id = be, 1

        def HasNormals(self):
            return self.be.normals is not None

        def HasColors(self):
            return False

    assert output._pcd_header(5, False) == (b"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n"
                                            b"COUNT 1 1 1\nWIDTH 5\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 5\nDATA binary\n")
    assert output.saveToFile(str(tmp_path / "a"), _Cloud(_Be(nrm)))
    p, q, _ = output.readPcd(str(tmp_path / "a.pcd"))
    np.testing.assert_array_equal(p, pts.astype(np.float32))
    np.testing.assert_array_equal(q, nrm.astype(np.float32))
    assert (tmp_path / "a.pcd").stat().st_size == len(output._pcd_header(37, True)) + 37 * 24
    assert output.saveToFile(str(tmp_path / "b.pcd"), _Cloud(_Be(None)))
    p, q, _ = output.readPcd(str(tmp_path / "b.pcd"))
    assert q is None and (tmp_path / "b.pcd").stat().st_size == len(output._pcd_header(37, False)) + 37 * 12
    np.testing.assert_array_equal(p, pts.astype(np.float32))
    assert not output.saveToFile(str(tmp_path / "no_such_dir" / "c"), _Cloud(_Be(None)))  # false, not an exception (WritePointCloudToPCD)


def test_cropper_containment_is_what_makes_the_narrow_crop_a_no_op(oracle):
    """crop(crop(x, V1), V2) = crop(x, V1) when V2 contains V1: ScanToMapIcp::processForScanMatchingAndMerging then hands out ONE cloud
    (ScanToMapRegistration.cpp:46-48; the shipped configurations use the same volume twice).  `contains` must only say yes when that holds
    -- checked on random points against the oracle's croppers, both directions."""
    import numpy as np

    from open3d_slam_amd import croppers as C

    rng = np.random.default_rng(4)
    pts = rng.uniform(-40, 40, size=(20000, 3))
    vols = [C.MaxRadiusCroppingVolume(20.0), C.MaxRadiusCroppingVolume(30.0), C.MinRadiusCroppingVolume(2.0), C.MinRadiusCroppingVolume(5.0),
            C.MinMaxRadiusCroppingVolume(2.0, 30.0), C.MinMaxRadiusCroppingVolume(3.0, 25.0), C.CylinderCroppingVolume(15.0, -1.0, 3.0),
            C.CylinderCroppingVolume(20.0, -2.0, 3.0), C.CroppingVolume()]
    shifted = C.MaxRadiusCroppingVolume(30.0)
    shifted.setPose(np.array([[1, 0, 0, 0.5], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]]))
    inverted = C.MaxRadiusCroppingVolume(30.0)
    inverted.setIsInvertVolume(True)
    vols += [shifted, inverted]

    def keep(v):
        a = v.to_abi()
        return set(oracle.crop_indices(pts, oracle.make_crop(a.kind, center=tuple(a.center), rmin=a.rmin, rmax=a.rmax, zmin=a.zmin, zmax=a.zmax,
                                                             invert=bool(a.invert))).tolist())

    kept = [keep(v) for v in vols]
    said_yes = 0
    for i, outer in enumerate(vols):
        for j, inner in enumerate(vols):
            if outer.contains(inner):
                said_yes += 1
                assert kept[j] <= kept[i], (i, j)  # never a false yes
    assert said_yes >= 12  # every volume contains itself (but the inverted one), the nested pairs, everything inside the base volume
    assert vols[4].contains(vols[4]) and vols[4].contains(vols[5]) and not vols[5].contains(vols[4])
    assert vols[8].contains(vols[0]) and not vols[0].contains(vols[8]) and not shifted.contains(vols[0]) and not inverted.contains(inverted)


def test_a_cropping_volume_s_abi_struct_follows_every_change_of_what_enters_it():
    """to_abi() keeps its struct between calls (a frame asks for the same volumes again and again); a new pose, a new radius or the
    inversion flag must give a new one, and the same state the same one."""
    import numpy as np

    from open3d_slam_amd import croppers as C

    v = C.MinMaxRadiusCroppingVolume(2.0, 30.0)
    a = v.to_abi()
    assert v.to_abi() is a and tuple(a.center) == (0.0, 0.0, 0.0) and (a.rmin, a.rmax, a.invert) == (2.0, 30.0, 0)
    T = np.eye(4)
    T[:3, 3] = [1.5, -2.0, 0.25]
    v.setPose(T)
    b = v.to_abi()
    assert b is not a and tuple(b.center) == (1.5, -2.0, 0.25) and v.to_abi() is b
    v.setPose(T.copy())  # the same translation again: nothing that enters the struct changed
    assert v.to_abi() is b
    v.setParameters(2.0, 25.0)
    c = v.to_abi()
    assert c is not b and c.rmax == 25.0 and tuple(c.center) == (1.5, -2.0, 0.25)
    v.setIsInvertVolume(True)
    d = v.to_abi()
    assert d is not c and d.invert == 1
    v.pose_[0, 3] = 9.0  # (even a write into the pose array is seen: the key is read from it)
    assert tuple(v.to_abi().center) == (9.0, -2.0, 0.25)
    cyl = C.CylinderCroppingVolume(10.0, -1.0, 2.0)
    assert (cyl.to_abi().rmax, cyl.to_abi().zmin, cyl.to_abi().zmax) == (10.0, -1.0, 2.0)
    cyl.setParameters(10.0, -1.5, 2.0)
    assert cyl.to_abi().zmin == -1.5


def test_a_cloud_seen_to_hold_points_is_not_asked_again_until_points_are_taken_from_it():
    """PointCloud.IsEmpty() remembers a non-empty answer (the mirror's clouds are made once and the map only grows); Submap.carve, the one
    place where points leave a cloud in place, forgets it.  An empty answer is never remembered (a size still in flight may resolve)."""
    from open3d_slam_amd.pointcloud import PointCloud

    class Be:
        h = None

        def __init__(self):
            self.bounds, self.n, self.asked = (0, 100), 0, 0

        def size_bound(self, cid):
            self.asked += 1
            return self.bounds

        def size(self, cid):
            self.asked += 1
            return self.n, False

    be = Be()
    c = PointCloud(be, 7, owns=False)
    assert c.IsEmpty() and be.asked == 2  # bounds say "perhaps": the exact size is asked for, and it is 0
    assert c.IsEmpty() and be.asked == 4  # ... and asked again the next time
    be.n = 5
    assert not c.IsEmpty() and be.asked == 6
    assert not c.IsEmpty() and not c.IsEmpty() and be.asked == 6  # remembered
    be.bounds, be.n = (0, 0), 0
    assert not c.IsEmpty()  # (nobody told the mirror that points were taken away)
    c.forget_size()
    assert c.IsEmpty() and be.asked == 7  # upper bound 0: empty without asking for the exact size
    be.bounds = (3, 50)
    assert not c.IsEmpty() and be.asked == 8 and not c.IsEmpty() and be.asked == 8


def test_random_down_sample_in_the_mirror_asks_the_device_to_draw_and_never_for_the_size(monkeypatch):
    """pointcloud.random_down_sample ([O3D] RandomDownSample at Odometry.cpp:29 / ScanToMapRegistration.cpp:39): under the default reading
    of SelectByIndex one 64-bit seed per call goes to o3ds_random_down_sample -- the cloud's size is not asked for (it may be in flight), no
    index list is made; the seeds are the caller's generator's, so two mirrors with the same seed name the same subsets, and the oracle
    loop's _down draws the same seed from the same generator.  ratio >= 1 keeps the cloud itself; the shuffled-order reading still goes
    through the host's permutation and o3ds_select_by_index."""
    from open3d_slam_amd import pointcloud
    from open3d_slam_amd.pointcloud import PointCloud, random_down_sample
    from oracle.pipeline import OracleLoop

    class Be:
        h = None

        def __init__(self):
            self.calls = []

        def random_down_sample(self, cid, ratio, seed):
            self.calls.append(("draw", cid, ratio, seed))
            return 100 + len(self.calls)

        def select_by_index(self, cid, idx):
            self.calls.append(("select", cid, len(idx)))
            return 200 + len(self.calls)

        def size(self, cid):
            self.calls.append(("size", cid))
            return 1000, False

        def size_bound(self, cid):
            self.calls.append(("size_bound", cid))
            return 0, 1000

        def free(self, cid):
            self.calls.append(("free", cid))

    monkeypatch.setattr(pointcloud, "SELECT_BY_INDEX_KEEPS_CLOUD_ORDER", True)
    be = Be()
    rng = np.random.default_rng(71)
    c = PointCloud(be, 7)
    assert random_down_sample(c, 1.0, rng) is c and be.calls == []
    out = random_down_sample(c, 0.3, rng)
    assert [x[0] for x in be.calls] == ["draw", "free"] and be.calls[0][1:3] == (7, 0.3) and out.id == 101
    seeds = [be.calls[0][3]]
    c2 = PointCloud(be, 8)
    random_down_sample(c2, 0.3, rng)
    seeds.append(be.calls[2][3])
    want = np.random.default_rng(71)
    assert seeds == [int(want.integers(0, 2**64, dtype=np.uint64)) for _ in range(2)] and seeds[0] != seeds[1]
    # the oracle loop draws the same seeds from the same generator (and applies them to whatever cloud it is handed)
    loop = OracleLoop.__new__(OracleLoop)
    loop.select_by_index_keeps_cloud_order, loop.shuffle_at_full_ratio = True, False
    from oracle.pipeline import draw_keep

    v = np.arange(3000.0).reshape(1000, 3)
    g = np.random.default_rng(71)
    got_v, _ = loop._down(v, -v, 0.3, g)
    np.testing.assert_array_equal(got_v, v[draw_keep(seeds[0], 1000, 0.3)])
    # the other reading of SelectByIndex: the shuffled list itself, made on the host from the exact size
    monkeypatch.setattr(pointcloud, "SELECT_BY_INDEX_KEEPS_CLOUD_ORDER", False)
    be.calls.clear()
    random_down_sample(PointCloud(be, 9), 0.3, np.random.default_rng(5))
    assert [x[0] for x in be.calls] == ["size", "select", "free"] and be.calls[1][2] == 300
