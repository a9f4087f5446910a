"""Mapper::addRangeMeasurement (include/open3d_slam/Mapper.hpp:47, src/Mapper.cpp:101-181) for ONE active submap:
the caller of the scan-to-map hot path, kept as a thin host harness with the reference's names.  Submap switching,
loop closure and the dense map are out of scope."""
from __future__ import annotations

import numpy as np

from .odometry import LidarOdometry
from .parameters import MapperParameters
from .pointcloud import PointCloud
from .scan_to_map_registration import scanToMapRegistrationFactory
from .submap import Submap


class Mapper:
    def __init__(self, be, odometry: LidarOdometry | None = None):
        self.be = be
        self.odometry_ = odometry
        self.params_ = MapperParameters()
        self.submap_ = Submap(be)
        self.mapToRangeSensor_ = np.eye(4)
        self.mapToRangeSensorPrev_ = np.eye(4)
        self.mapToRangeSensorLastScanInsertion_ = np.eye(4)
        self.lastMeasurementTimestamp_ = None
        self.mapToRangeSensorBuffer_: list[tuple[float, np.ndarray]] = []
        self.lastResult_ = None
        self.update(self.params_)

    def setParameters(self, p: MapperParameters):  # Mapper.cpp:35-38
        self.params_ = p
        self.update(p)

    def update(self, p: MapperParameters):  # Mapper.cpp:53-56
        self.scan2MapReg_ = scanToMapRegistrationFactory(p)
        self.submap_.setParameters(p)

    def getActiveSubmap(self) -> Submap:
        return self.submap_

    def getAssembledMapPointCloud(self) -> PointCloud:  # Mapper.cpp:183-208 (this harness holds one submap)
        from .output import assembleMapPointCloud
        return assembleMapPointCloud(self.be, [self.submap_])

    def getMapToRangeSensor(self) -> np.ndarray:
        return self.mapToRangeSensor_

    def addRangeMeasurement(self, rawScan: PointCloud, timestamp: float) -> bool:
        p = self.params_
        if self.submap_.isEmpty():  # Mapper.cpp:105-114: insert the first scan at identity
            processed = self.scan2MapReg_.processForScanMatchingAndMerging(rawScan, self.mapToRangeSensor_)
            self.submap_.insertScan(rawScan, processed.merge_, np.eye(4), timestamp, isPerformCarving=True)
            self.mapToRangeSensorBuffer_.append((timestamp, self.mapToRangeSensor_.copy()))
            # (the reference leaves lastMeasurementTimestamp_ at the epoch here and lets TransformInterpolationBuffer clamp the lookup of
            # the next frame to the earliest odometry sample; this harness looks stamps up exactly, so it records the first stamp)
            self.lastMeasurementTimestamp_ = timestamp
            self._release(processed)
            return True
        if timestamp < self.lastMeasurementTimestamp_:
            return False
        estimate = self.mapToRangeSensorPrev_.copy()
        if self.odometry_ is not None and self.odometry_.hasTransform(timestamp) and self.odometry_.hasTransform(self.lastMeasurementTimestamp_):
            # Mapper.cpp:130-137: predict with the odometry motion since the last mapped scan
            odomNow = self.odometry_.getOdomToRangeSensor(timestamp)
            odomPrev = self.odometry_.getOdomToRangeSensor(self.lastMeasurementTimestamp_)
            estimate = self.mapToRangeSensorPrev_ @ (np.linalg.inv(odomPrev) @ odomNow)
        processed = self.scan2MapReg_.processForScanMatchingAndMerging(rawScan, self.mapToRangeSensor_)
        result = self.scan2MapReg_.scanToMapRegistration(processed.match_, self.submap_, self.mapToRangeSensor_, estimate)
        self.lastResult_ = result
        if not p.isIgnoreMinRefinementFitness_ and result.fitness_ < p.scanMatcher_.minRefinementFitness_:
            self._release(processed)
            return False  # Mapper.cpp:151-156: pose not updated, scan not inserted
        self.mapToRangeSensor_ = np.array(result.transformation_)
        self.mapToRangeSensorBuffer_.append((timestamp, self.mapToRangeSensor_.copy()))
        motion = np.linalg.inv(self.mapToRangeSensorLastScanInsertion_) @ self.mapToRangeSensor_
        if not (np.linalg.norm(motion[:3, 3]) < p.minMovementBetweenMappingSteps_):  # Mapper.cpp:170-176
            # SubmapCollection::insertScan (SubmapCollection.cpp:178,189,203) always asks the submap to carve; Submap::carve applies the
            # every-N-scans gate itself (Submap.cpp:111)
            self.submap_.insertScan(rawScan, processed.merge_, self.mapToRangeSensor_, timestamp, isPerformCarving=True)
            self.mapToRangeSensorLastScanInsertion_ = self.mapToRangeSensor_.copy()
        self.lastMeasurementTimestamp_ = timestamp
        self.mapToRangeSensorPrev_ = self.mapToRangeSensor_.copy()
        self._release(processed)
        return True

    @staticmethod
    def _release(processed):
        if processed.match_ is not processed.merge_:
            processed.match_.release()
        processed.merge_.release()
