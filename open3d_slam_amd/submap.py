"""Submap map-fusion core (include/open3d_slam/Submap.hpp:38-39,71-90, src/Submap.cpp:39-149) on a device map: the sparse map
(insertScan, carve, transform) and the dense voxel map (insertScanDenseMap and its carving).  The C++ twin of this class is
open3d_slam_amd/host/o3ds_mapping.hpp."""
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
        self.nScansInsertedDenseMap_ = 0
        self._denseMap = None  # o3ds_dense_map id, made on first use (Submap::update move-assigns a fresh VoxelizedPointCloud, Submap.cpp:211)
        self.update(self.params_)

    def setParameters(self, p: MapperParameters):  # Submap.cpp:146-149
        self.params_ = p
        self.update(p)

    def update(self, p: MapperParameters):  # Submap.cpp:208-216
        self.mapBuilderCropper_ = croppingVolumeFactory(p.mapBuilder_.cropper_)
        self.denseMapCropper_ = croppingVolumeFactory(p.denseMapBuilder_.cropper_)
        if self._denseMap is not None:
            self.be.dense_map_free(self._denseMap)
            self._denseMap = None

    def _dense(self) -> int:
        if self._denseMap is None:
            self._denseMap = self.be.dense_map_create(self.params_.denseMapBuilder_.mapVoxelSize_)
        return self._denseMap

    def getDenseMapSize(self) -> int:
        return 0 if self._denseMap is None else self.be.dense_map_size(self._denseMap)

    def getDenseMapPointCloud(self) -> PointCloud:
        """getDenseMap().toPointCloud() (Voxel.cpp:18-36): voxel means, ascending key order."""
        return PointCloud(self.be, self.be.dense_map_to_cloud(self._dense()))

    def insertScanDenseMap(self, rawScan: PointCloud, mapToRangeSensor, time=None, isPerformCarving: bool = False) -> bool:
        """Submap.cpp:77-92: crop the raw scan with the dense-map volume (sensor frame), drop colours outside [0, 1]^3 (ColorRangeCropper,
        default bounds), insert T * scan into the voxel map; every carveSpaceEveryNscans_-th call carve with the RAW scan (sensor
        frame, as the reference passes it) from the map-frame sensor position (Submap.cpp:88,126-136)."""
        be = self.be
        T = np.array(mapToRangeSensor, dtype=np.float64)
        self.denseMapCropper_.setPose(np.eye(4))
        cropped = self.denseMapCropper_.crop(rawScan)
        if cropped.HasColors():
            col = cropped.colors_
            ok = np.flatnonzero(np.all((col >= 0.0) & (col <= 1.0), axis=1)).astype(np.uint32)
            if len(ok) != len(col):
                kept = PointCloud(be, be.select_by_index(cropped.id, ok))
                cropped.release()
                cropped = kept
        if not cropped.IsEmpty():
            be.dense_map_insert(self._dense(), cropped.id, T)
        cropped.release()
        c = self.params_.denseMapBuilder_.carving_
        if isPerformCarving and self.getDenseMapSize() > 0 and self.nScansInsertedDenseMap_ % c.carveSpaceEveryNscans_ == 1:
            be.dense_map_carve(self._dense(), rawScan.id, T[:3, 3], None, radius=c.neighborhoodRadiusDenseMap_,
                               max_length=c.maxRaytracingLength_, truncation=c.truncationDistance_)
        self.nScansInsertedDenseMap_ += 1
        return True

    def transform(self, T):
        """Submap::transform (Submap.cpp:94-107): the sparse map (its NN index is rebuilt), the dense map as VoxelizedPointCloud::transform
        is written, mapToRangeSensor_ = mapToRangeSensor_ * T."""
        T = np.array(T, dtype=np.float64)
        if not self.mapCloud_.IsEmpty():
            moved = PointCloud(self.be, self.be.transform_cloud(self.mapCloud_.id, T))
            self.mapCloud_.release()
            self.mapCloud_ = moved
            self.be.build_index(self.mapCloud_.id, self.params_.scanMatcher_.icp_.maxCorrespondenceDistance_)
        if self._denseMap is not None:
            self.be.dense_map_transform(self._denseMap, T)
        self.mapToRangeSensor_ = self.mapToRangeSensor_ @ T

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
            self.carve(rawScan, self.mapToRangeSensor_, want_count=False)
        self.mapBuilderCropper_.setPose(self.mapToRangeSensor_)
        self.be.map_insert_scan(self.mapCloud_.id, preProcessedScan.id, self.mapToRangeSensor_, self.params_.mapBuilder_.mapVoxelSize_,
                                self.mapBuilderCropper_.to_abi(), max_corr_hint=icp.maxCorrespondenceDistance_)
        self.nScansInsertedMap_ += 1
        return True

    def carve(self, rawScan: PointCloud, mapToRangeSensor, want_count: bool = True):
        """Submap::carve (Submap.cpp:109-125): only every carveSpaceEveryNscans_-th insertion, never on an empty map.  Returns the number of
        removed points (the reference returns nothing; want_count=False does not wait for it)."""
        c = self.params_.mapBuilder_.carving_
        if self.mapCloud_.IsEmpty() or not (self.nScansInsertedMap_ % c.carveSpaceEveryNscans_ == 1):
            return 0
        self.mapCloud_.forget_size()  # (the one place where points leave a cloud of the mirror in place)
        return self.be.map_carve(self.mapCloud_.id, rawScan.id, mapToRangeSensor, self.mapBuilderCropper_.to_abi(), voxel=c.voxelSize_,
                                 max_length=c.maxRaytracingLength_, truncation=c.truncationDistance_, min_dot=c.minDotProductWithNormal_,
                                 want_count=want_count)
