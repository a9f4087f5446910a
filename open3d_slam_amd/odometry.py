"""LidarOdometry (include/open3d_slam/Odometry.hpp:27, src/Odometry.cpp:25-79): a CALLER of the hot path, kept as a thin
host harness with the reference's method names so that config 3 (full odometry + mapper loop) can be driven end to end.
All arithmetic goes through the CloudRegistration seam on the device."""
from __future__ import annotations

import numpy as np

from .cloud_registration import cloudRegistrationFactory
from .croppers import CroppingVolume, croppingVolumeFactory
from .parameters import OdometryParameters
from .pointcloud import PointCloud, random_down_sample, shared_preprocess

_IDENTITY = np.eye(4)
_IDENTITY.setflags(write=False)  # (the initial guess of every scan-to-scan registration: read, never written)


class LidarOdometry:
    def __init__(self, be):
        self.be = be
        self.params_ = OdometryParameters()
        self.cropper_ = CroppingVolume()  # Odometry.cpp:20-23: base volume = everything, until setParameters
        self.cloudRegistration_ = cloudRegistrationFactory(self.params_.scanMatcher_)
        self.cloudPrev_: PointCloud | None = None
        self.odomToRangeSensorCumulative_ = np.eye(4)
        self.odomToRangeSensorBuffer_: list[tuple[float, np.ndarray]] = []
        self.lastMeasurementTimestamp_ = None
        self._downsample_rng = None
        self._shuffle_at_full_ratio = False

    def setParameters(self, p: OdometryParameters):  # Odometry.cpp:96-100
        self.params_ = p
        self.cropper_ = croppingVolumeFactory(p.scanProcessing_.cropper_)
        self.cloudRegistration_ = cloudRegistrationFactory(p.scanMatcher_)

    def preprocess(self, cloud: PointCloud) -> PointCloud:  # Odometry.cpp:25-30
        # cropper_->crop(in) then voxelize(voxelSize_, cropped) (Odometry.cpp:26-27) as one call, same result bit for bit
        vox = shared_preprocess(cloud, self.cropper_.to_abi(), self.params_.scanProcessing_.voxelSize_, self.cloudRegistration_)
        # RandomDownSample(ratio) (Odometry.cpp:29); setDownSampleSeed pins the kept-index lists
        return random_down_sample(vox, self.params_.scanProcessing_.downSamplingRatio_, self._downsample_rng, self._shuffle_at_full_ratio)

    def preprocessAhead(self, cloud: PointCloud) -> None:
        """crop -> voxelize -> normals of a raw scan that addRangeScan will be given next, queued now: the result waits on the raw scan
        (pointcloud.shared_preprocess) for preprocess() -- which still draws its own RandomDownSample -- and for the mapper.  What
        open3d_slam's odometry worker does on its own thread while the mapper is busy with the previous scan (SlamWrapper.cpp:228-229);
        a single-threaded driver calls it from o3ds_icp_overlap_next (Backend.overlap_next), behind a registration's launches."""
        shared_preprocess(cloud, self.cropper_.to_abi(), self.params_.scanProcessing_.voxelSize_, self.cloudRegistration_).release()

    def setDownSampleSeed(self, seed: int | None, shuffle_at_full_ratio: bool = False):
        self._downsample_rng = None if seed is None else np.random.default_rng(seed)
        self._shuffle_at_full_ratio = bool(shuffle_at_full_ratio)

    def addRangeScan(self, cloud: PointCloud, timestamp: float) -> bool:  # Odometry.cpp:32-79
        if self.cloudPrev_ is None or self.cloudPrev_.IsEmpty():
            self.cloudPrev_ = self.preprocess(cloud)
            self.odomToRangeSensorBuffer_.append((timestamp, self.odomToRangeSensorCumulative_.copy()))
            self.lastMeasurementTimestamp_ = timestamp
            return True
        if timestamp < self.lastMeasurementTimestamp_:
            return False  # measurements came out of order
        pre = self.preprocess(cloud)
        # B3: registers PREVIOUS -> CURRENT and integrates the inverse; the target normals are the current scan's
        # (the grid the normal estimation just built for `pre` is kept with the cloud and serves as the registration's target index)
        result = self.cloudRegistration_.registerClouds(self.cloudPrev_, pre, _IDENTITY)
        ok = result.fitness_ > 0.1  # Odometry.cpp:51 ("todo magic")
        if not ok:
            if not pre.IsEmpty():
                self.cloudPrev_.release()
                self.cloudPrev_ = pre
            else:
                pre.release()  # nothing to keep: give the (empty) device cloud back
            return False
        self.odomToRangeSensorCumulative_ = self.odomToRangeSensorCumulative_ @ np.linalg.inv(result.transformation_)
        self.cloudPrev_.release()
        self.cloudPrev_ = pre
        self.odomToRangeSensorBuffer_.append((timestamp, self.odomToRangeSensorCumulative_.copy()))
        self.lastMeasurementTimestamp_ = timestamp
        return True

    def getOdomToRangeSensor(self, t: float) -> np.ndarray:
        """exact-stamp lookup (the reference interpolates in TransformInterpolationBuffer; the harness feeds exact stamps).  The stamps
        asked for are the newest ones: looked for from the back (a scan of the whole buffer per call grew with the stream)."""
        for ts, T in reversed(self.odomToRangeSensorBuffer_):
            if ts == t:
                return T
            if ts < t:
                break  # stamps ascend
        raise RuntimeError("odomToRangeSensorBuffer_ does not have the desired transform")

    def hasTransform(self, t: float) -> bool:
        for ts, _ in reversed(self.odomToRangeSensorBuffer_):
            if ts == t:
                return True
            if ts < t:
                break
        return False
