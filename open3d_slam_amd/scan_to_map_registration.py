"""ScanToMapRegistration seam (include/open3d_slam/ScanToMapRegistration.hpp:24-59, src/ScanToMapRegistration.cpp)."""
from __future__ import annotations

import dataclasses

import numpy as np

from .cloud_registration import RegistrationResult, cloudRegistrationFactory
from .croppers import croppingVolumeFactory
from .parameters import (CloudRegistrationParameters, CloudRegistrationType, MapperParameters, ScanToMapRegistrationParameters,
                         ScanToMapRegistrationType)
from .pointcloud import PointCloud, random_down_sample, shared_preprocess

_IDENTITY = np.eye(4)
_IDENTITY.setflags(write=False)


@dataclasses.dataclass
class ProcessedScans:  # ScanToMapRegistration.hpp:24-27
    merge_: PointCloud
    match_: PointCloud


def toCloudRegistrationType(p: ScanToMapRegistrationParameters) -> CloudRegistrationParameters:  # ScanToMapRegistration.cpp:104-129
    ret = CloudRegistrationParameters()
    ret.icp_ = p.icp_
    m = {ScanToMapRegistrationType.PointToPlaneIcp: CloudRegistrationType.PointToPlaneIcp,
         ScanToMapRegistrationType.PointToPointIcp: CloudRegistrationType.PointToPointIcp,
         ScanToMapRegistrationType.GeneralizedIcp: CloudRegistrationType.GeneralizedIcp}
    if p.scanToMapRegType_ not in m:
        raise RuntimeError("Conversion not possible from ScanToMapRegistrationParameters to CloudRegistrationParameters, "
                           "for this particular scan to map reg type")
    ret.regType_ = m[p.scanToMapRegType_]
    return ret


class ScanToMapRegistration:  # ScanToMapRegistration.hpp:29-38
    def processForScanMatchingAndMerging(self, cloud: PointCloud, mapToRangeSensor) -> ProcessedScans:
        raise NotImplementedError

    def scanToMapRegistration(self, scan: PointCloud, activeSubmap, mapToRangeSensor, initialGuess) -> RegistrationResult:
        raise NotImplementedError

    def isMergeScanValid(self, cloud: PointCloud) -> bool:
        raise NotImplementedError

    def prepareInitialMap(self, map_: PointCloud) -> None:
        raise NotImplementedError


class ScanToMapIcp(ScanToMapRegistration):  # ScanToMapRegistration.hpp:40-59
    def __init__(self):
        self.params_ = MapperParameters()
        self._downsample_rng = None  # [O3D] RandomDownSample seeds from random_device; see setDownSampleSeed
        self.update(self.params_)

    def setParameters(self, p: MapperParameters):  # ScanToMapRegistration.cpp:24-27
        self.params_ = p
        self.update(p)

    def update(self, p: MapperParameters):  # ScanToMapRegistration.cpp:28-32
        self.mapBuilderCropper_ = croppingVolumeFactory(self.params_.mapBuilder_.cropper_)
        self.scanMatcherCropper_ = croppingVolumeFactory(self.params_.scanProcessing_.cropper_)
        self.cloudRegistration = cloudRegistrationFactory(toCloudRegistrationType(p.scanMatcher_))

    def setDownSampleSeed(self, seed: int | None, shuffle_at_full_ratio: bool = False):
        """The reference's RandomDownSample is seeded from std::random_device (non-reproducible); a seed makes
        the kept-index lists reproducible (one generator, advanced once per scan)."""
        self._downsample_rng = None if seed is None else np.random.default_rng(seed)
        self._shuffle_at_full_ratio = bool(shuffle_at_full_ratio)

    def _random_down_sample(self, cloud: PointCloud, ratio: float) -> PointCloud:
        return random_down_sample(cloud, ratio, self._downsample_rng, getattr(self, "_shuffle_at_full_ratio", False))

    def preprocess(self, cloud: PointCloud) -> PointCloud:  # ScanToMapRegistration.cpp:35-40
        # mapBuilderCropper_->crop(in), voxelize(voxelSize_, cropped), estimateNormalsOrCovariancesIfNeeded (.cpp:36-38): the cloud the
        # odometry has just computed from the same raw scan when the two are configured alike (pointcloud.shared_preprocess)
        voxelized = shared_preprocess(cloud, self.mapBuilderCropper_.to_abi(), self.params_.scanProcessing_.voxelSize_, self.cloudRegistration)
        return self._random_down_sample(voxelized, self.params_.scanProcessing_.downSamplingRatio_)

    def processForScanMatchingAndMerging(self, cloud: PointCloud, mapToRangeSensor) -> ProcessedScans:  # .cpp:42-54
        wide = self.preprocess(cloud)
        self.scanMatcherCropper_.setPose(_IDENTITY)
        # the scan-matcher volume of the shipped configurations is the map-builder volume (or contains it): cropping what the latter kept
        # returns it unchanged, so the second cloud is the first (no kernels, no size read-back)
        narrow = wide if self.scanMatcherCropper_.contains(self.mapBuilderCropper_) else self.scanMatcherCropper_.crop(wide)
        if narrow.IsEmpty():  # (assert_gt(size, 0): decided without waiting for a size that is still in flight on the device)
            raise RuntimeError("ScanToMapIcp::narrow cropped size is zero")
        if wide.IsEmpty():
            raise RuntimeError("ScanToMapIcp::wideCropped cropped size is zero")
        return ProcessedScans(merge_=wide, match_=narrow)

    def scanToMapRegistration(self, scan: PointCloud, activeSubmap, mapToRangeSensor, initialGuess) -> RegistrationResult:
        """ScanToMapRegistration.cpp:55-62.  The reference copies the map patch inside the scan-matcher volume around
        the PREVIOUS pose (B1) and rebuilds a KD-tree on it; here the volume is a predicate fused into the search over
        the submap's resident index -- same correspondences, no O(N) copy or rebuild per scan."""
        mapCloud = activeSubmap.getMapPointCloud()
        self.scanMatcherCropper_.setPose(mapToRangeSensor)
        if mapCloud.IsEmpty():
            raise RuntimeError("map patch size is zero")
        r = self.cloudRegistration.registerClouds(scan, mapCloud, initialGuess, target_crop=self.scanMatcherCropper_.to_abi())
        return r

    def isMergeScanValid(self, cloud: PointCloud) -> bool:  # ScanToMapRegistration.cpp:64-80
        t = self.params_.scanMatcher_.scanToMapRegType_
        if t == ScanToMapRegistrationType.PointToPlaneIcp:
            return cloud.HasNormals()
        if t == ScanToMapRegistrationType.PointToPointIcp:
            return True
        if t == ScanToMapRegistrationType.GeneralizedIcp:
            return cloud.HasNormals()
        raise RuntimeError("cannot check whether merge scan is valid for this registration type")

    def prepareInitialMap(self, map_: PointCloud) -> None:  # ScanToMapRegistration.cpp:81-84
        self.cloudRegistration.estimateNormalsOrCovariancesIfNeeded(map_)


def createScanToMapIcp(p: MapperParameters) -> ScanToMapIcp:
    ret = ScanToMapIcp()
    ret.setParameters(p)
    return ret


def scanToMapRegistrationFactory(p: MapperParameters) -> ScanToMapRegistration:  # ScanToMapRegistration.cpp:91-102
    if p.scanMatcher_.scanToMapRegType_ in tuple(ScanToMapRegistrationType):
        return createScanToMapIcp(p)
    raise RuntimeError("scanToMapRegistrationFactory: unknown type of registration scan to map")
