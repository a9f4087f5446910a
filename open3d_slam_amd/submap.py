"""Submap map-fusion core (include/open3d_slam/Submap.hpp:38-39,71-90, src/Submap.cpp:39-75,138-144) on a device map."""
from __future__ import annotations

import numpy as np

from .croppers import croppingVolumeFactory
from .parameters import MapperParameters
from .pointcloud import PointCloud


class Submap:
    def __init__(self, be, id_: int = 0, parentId: int = 0):
        self.be = be
        self.id_, self.parentId_ = id_, parentId
        self.params_ = MapperParameters()
        self.mapCloud_ = PointCloud.from_numpy(be, np.zeros((0, 3)))
        self.mapToRangeSensor_ = np.eye(4)
        self.nScansInsertedMap_ = 0
        self.update(self.params_)

    def setParameters(self, p: MapperParameters):  # Submap.cpp:146-149
        self.params_ = p
        self.update(p)

    def update(self, p: MapperParameters):
        self.mapBuilderCropper_ = croppingVolumeFactory(p.mapBuilder_.cropper_)

    def getMapPointCloud(self) -> PointCloud:
        return self.mapCloud_

    def isEmpty(self) -> bool:
        return self.mapCloud_.IsEmpty()

    def getMapToRangeSensor(self):
        return self.mapToRangeSensor_

    def insertScan(self, rawScan, preProcessedScan: PointCloud, mapToRangeSensor, time=None, isPerformCarving: bool = False) -> bool:
        """Submap.cpp:39-75: [carve the map with the raw scan,] map += T * scan; re-voxelize inside the map-builder volume centred
        on the sensor; rebuild the NN index."""
        if preProcessedScan.IsEmpty():
            return True
        self.mapToRangeSensor_ = np.array(mapToRangeSensor, dtype=np.float64)
        icp = self.params_.scanMatcher_.icp_
        if self.params_.isUseInitialMap_ and self.mapCloud_.IsEmpty():  # Submap.cpp:47-52
            self.be.cloud_append(self.mapCloud_.id, preProcessedScan.id)
            v = self.be.voxel_down_sample(self.mapCloud_.id, self.params_.mapBuilder_.mapVoxelSize_)
            self.mapCloud_.release()
            self.mapCloud_ = PointCloud(self.be, v)
            self.be.build_index(self.mapCloud_.id, icp.maxCorrespondenceDistance_)
            return True
        if isPerformCarving:  # Submap.cpp:56-60 (the cropper still holds the pose of the previous insertion, as in the reference)
            self.carve(rawScan, self.mapToRangeSensor_)
        self.mapBuilderCropper_.setPose(self.mapToRangeSensor_)
        self.be.map_insert_scan(self.mapCloud_.id, preProcessedScan.id, self.mapToRangeSensor_, self.params_.mapBuilder_.mapVoxelSize_,
                                self.mapBuilderCropper_.to_abi(), max_corr_hint=icp.maxCorrespondenceDistance_)
        self.nScansInsertedMap_ += 1
        return True

    def carve(self, rawScan: PointCloud, mapToRangeSensor) -> int:
        """Submap::carve (Submap.cpp:109-125): only every carveSpaceEveryNscans_-th insertion, never on an empty map."""
        c = self.params_.mapBuilder_.carving_
        if self.mapCloud_.IsEmpty() or not (self.nScansInsertedMap_ % c.carveSpaceEveryNscans_ == 1):
            return 0
        return self.be.map_carve(self.mapCloud_.id, rawScan.id, mapToRangeSensor, self.mapBuilderCropper_.to_abi(), voxel=c.voxelSize_,
                                 max_length=c.maxRaytracingLength_, truncation=c.truncationDistance_, min_dot=c.minDotProductWithNormal_)
