"""TEST INFRASTRUCTURE (checker, never shipped or measured as the product): the orchestration of one lidar frame -- odometry
(Odometry.cpp:25-79) and mapping (Mapper.cpp:101-181, ScanToMapRegistration.cpp:35-62, Submap.cpp:39-75) -- played on the CPU oracle.
Used by tests/test_pipeline_gpu.py as the reference loop and by bench.py as the cpu_baseline leg of the scans/s metric."""
import numpy as np


def draw_keys(seed: int, n: int) -> np.ndarray:
    """The 64-bit key of every index under o3ds_random_down_sample's counter-based generator (include/o3ds_backend.h): splitmix64's finaliser of
    seed + (i + 1) * 0x9E3779B97F4A7C15, all arithmetic modulo 2^64.  Distinct arguments, a bijective mix: distinct keys."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def draw_keep(seed: int, n: int, ratio: float) -> np.ndarray:
    """[O3D] RandomDownSample as this repository draws it: the k = int(ratio * n) indices with the smallest keys, ascending (the cloud order
    [O3D] SelectByIndex's mask walk emits).  Open3D itself shuffles with an mt19937 seeded from std::random_device: any uniformly drawn
    k-subset is a faithful outcome, none is reproducible; this one is a function of the seed."""
    k = int(ratio * n)
    if k <= 0:
        return np.zeros(0, dtype=np.int64)
    keys = draw_keys(seed, n)
    return np.sort(np.argpartition(keys, k - 1)[:k]) if k < n else np.arange(n)


class OracleLoop:
    """The same orchestration (Odometry.cpp:25-79, Mapper.cpp:101-181, ScanToMapRegistration.cpp:35-62, Submap.cpp:39-75) on
    the CPU oracle."""

    def __init__(self, oracle, mp, op):
        self.o, self.mp, self.op = oracle, mp, op
        self.map_p = np.zeros((0, 3))
        self.map_n = np.zeros((0, 3))
        self.T = np.eye(4)
        self.Tprev = np.eye(4)
        self.odom = np.eye(4)
        self.odom_at = {}
        self.prev = None
        self.last_t = None
        self.rng_odo = self.rng_map = None  # RandomDownSample keep lists (see set_down_sample_seeds)
        self.carving = True  # as the reference: SubmapCollection::insertScan passes isPerformCarving = true
        self.n_inserted, self.n_carved = 0, 0
        self.strict = True  # a failed fitness gate is an error (the streams of this repository never fail one); False: follow the reference's rejection paths
        self.n_rejected, self.n_not_inserted = 0, 0
        self.T_inserted = np.eye(4)  # pose of the last insertion = the map builder cropper's pose until the next one
        self.shuffle_at_full_ratio = False

    def set_down_sample_seeds(self, odo_seed, map_seed, shuffle_at_full_ratio=False):
        """[O3D] RandomDownSample (Odometry.cpp:29, ScanToMapRegistration.cpp:39) shuffles with a generator seeded from
        std::random_device; the test hands both sides the same numpy generators, advanced once per scan, so both keep the same
        index list: permutation(n)[: int(ratio * n)] (see select_by_index_keeps_cloud_order for the order of the kept points)."""
        self.rng_odo = np.random.default_rng(odo_seed)
        self.rng_map = np.random.default_rng(map_seed)
        self.shuffle_at_full_ratio = shuffle_at_full_ratio

    # [O3D] SelectByIndex walks the cloud through a mask of the listed indices: the kept points come out in CLOUD order (v0.15.1
    # PointCloud.cpp, restated; unpinned).  False: in the order of the shuffled list (SURVEY A.7's reading).  The device loop has the same switch.
    select_by_index_keeps_cloud_order = True

    def _down(self, v, n, ratio, rng):
        cloud_order = self.select_by_index_keeps_cloud_order
        if ratio >= 1.0 and (cloud_order or not self.shuffle_at_full_ratio):
            return v, n
        if cloud_order:  # the draw the device makes (o3ds_random_down_sample): one seed per call, whatever the size of the cloud
            seed = int((rng or np.random.default_rng()).integers(0, 2**64, dtype=np.uint64))
            keep = draw_keep(seed, len(v), min(ratio, 1.0))
            return v[keep], n[keep]
        if len(v) == 0:
            return v, n
        keep = (rng or np.random.default_rng()).permutation(len(v))[: int(min(ratio, 1.0) * len(v))]
        return v[keep], n[keep]

    def _pre(self, raw, crop_p, voxel, icp):
        o = self.o
        keep = o.crop_indices(raw, o.make_crop(o.CROP_MIN_MAX_RADIUS, rmin=crop_p.croppingMinRadius_, rmax=crop_p.croppingMaxRadius_))
        v = o.voxel_down_sample(raw[keep], voxel)
        return v, o.estimate_normals(v, icp.maxDistanceKnn_, icp.knn_)

    def odometry(self, raw, t):
        icp = self.op.scanMatcher_.icp_
        v, n = self._pre(raw, self.op.scanProcessing_.cropper_, self.op.scanProcessing_.voxelSize_, icp)
        v, n = self._down(v, n, self.op.scanProcessing_.downSamplingRatio_, self.rng_odo)
        if self.prev is not None:
            r = self.o.icp_point_to_plane(self.prev, v, n, icp.maxCorrespondenceDistance_, max_iter=icp.maxNumIter_)
            if not r["fitness"] > 0.1:  # Odometry.cpp:52-67: the new scan still becomes the one to match against, nothing else moves
                assert not self.strict, ("odometry fitness", r["fitness"])
                if len(v):
                    self.prev = v
                return False
            self.odom = self.odom @ np.linalg.inv(r["transformation"])
        self.prev = v
        self.odom_at[t] = self.odom.copy()
        return True

    def mapping(self, raw, t):
        o, mp = self.o, self.mp
        icp = mp.scanMatcher_.icp_
        v, n = self._pre(raw, mp.mapBuilder_.cropper_, mp.scanProcessing_.voxelSize_, icp)
        v, n = self._down(v, n, mp.scanProcessing_.downSamplingRatio_, self.rng_map)
        sc = mp.scanProcessing_.cropper_
        keep = o.crop_indices(v, o.make_crop(o.CROP_MIN_MAX_RADIUS, rmin=sc.croppingMinRadius_, rmax=sc.croppingMaxRadius_))
        match = v[keep]
        if len(self.map_p) == 0:
            T_ins = np.eye(4)
        else:
            est = self.Tprev @ (np.linalg.inv(self.odom_at[self.last_t]) @ self.odom_at[t])
            patch = o.crop_indices(self.map_p, o.make_crop(o.CROP_MIN_MAX_RADIUS, center=self.T[:3, 3], rmin=sc.croppingMinRadius_,
                                                            rmax=sc.croppingMaxRadius_))
            r = o.icp_point_to_plane(match, self.map_p[patch], self.map_n[patch], icp.maxCorrespondenceDistance_, init=est,
                                     max_iter=icp.maxNumIter_)
            if not mp.isIgnoreMinRefinementFitness_ and r["fitness"] < mp.scanMatcher_.minRefinementFitness_:
                # Mapper.cpp:151-156: the refinement is skipped -- no pose update, no insertion, and lastMeasurementTimestamp_ /
                # mapToRangeSensorPrev_ stay where they were, so the next prior spans the odometry motion since the last ACCEPTED scan
                assert not self.strict, ("scan-to-map fitness", r["fitness"])
                self.n_rejected += 1
                return False
            self.T = r["transformation"]
            T_ins = self.T
            # Mapper.cpp:170-176: a sensor that has not moved minMovementBetweenMappingSteps_ since the last insertion does not insert
            motion = np.linalg.inv(self.T_inserted) @ self.T
            if np.linalg.norm(motion[:3, 3]) < mp.minMovementBetweenMappingSteps_:
                self.n_not_inserted += 1
                self.last_t = t
                self.Tprev = self.T.copy()
                return True
        tp, tn = o.transform_points(v, T_ins), o.transform_normals(n, T_ins)
        mc = mp.mapBuilder_.cropper_
        # Submap::insertScan (Submap.cpp:54-72): SubmapCollection::insertScan always asks for carving (SubmapCollection.cpp:178,189,203);
        # Submap::carve (Submap.cpp:109-125) acts when the map is not empty and nScansInsertedMap_ % carveSpaceEveryNscans_ == 1, with
        # the RAW scan placed by the new pose, on the map points inside the map builder's volume -- whose pose is still that of the
        # PREVIOUS insertion (setPose follows, Submap.cpp:71)
        cv = mp.mapBuilder_.carving_
        if self.carving and len(self.map_p) and self.n_inserted % cv.carveSpaceEveryNscans_ == 1:
            inside = o.crop_indices(self.map_p, o.make_crop(o.CROP_MIN_MAX_RADIUS, center=self.T_inserted[:3, 3], rmin=mc.croppingMinRadius_,
                                                             rmax=mc.croppingMaxRadius_))
            gone = o.carve_flags(o.transform_points(np.asarray(raw, dtype=np.float64), T_ins), T_ins[:3, 3], self.map_p, self.map_n, inside,
                                 voxel=cv.voxelSize_, max_length=cv.maxRaytracingLength_, truncation=cv.truncationDistance_,
                                 min_dot=cv.minDotProductWithNormal_)
            self.map_p, self.map_n = self.map_p[~gone], self.map_n[~gone]
            self.n_carved += int(gone.sum())
        self.n_inserted += 1
        self.T_inserted = T_ins.copy()
        crop = o.make_crop(o.CROP_MIN_MAX_RADIUS, center=T_ins[:3, 3], rmin=mc.croppingMinRadius_, rmax=mc.croppingMaxRadius_)
        self.map_p, self.map_n, _ = o.voxelize_within_volume(np.vstack([self.map_p, tp]), np.vstack([self.map_n, tn]),
                                                             mp.mapBuilder_.mapVoxelSize_, crop)
        self.last_t = t
        self.Tprev = self.T.copy()
        return True
