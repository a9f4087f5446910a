"""Parameter structs of the hot path, same names and defaults as the reference's
open3d_slam/include/open3d_slam/Parameters.hpp (line numbers cited per struct)."""
from __future__ import annotations

import dataclasses
import enum

import numpy as np


class CloudRegistrationType(enum.IntEnum):  # Parameters.hpp:37
    PointToPlaneIcp = 0
    PointToPointIcp = 1
    GeneralizedIcp = 2


CloudRegistrationStringToEnumMap = {t.name: t for t in CloudRegistrationType}  # Parameters.hpp:39-42


class ScanToMapRegistrationType(enum.IntEnum):  # Parameters.hpp:44
    PointToPlaneIcp = 0
    PointToPointIcp = 1
    GeneralizedIcp = 2


ScanToMapRegistrationStringToEnumMap = {t.name: t for t in ScanToMapRegistrationType}  # Parameters.hpp:46-49


@dataclasses.dataclass
class ScanCroppingParameters:  # Parameters.hpp:51-57
    croppingMinZ_: float = -10.0
    croppingMaxZ_: float = 10.0
    croppingMinRadius_: float = 0.0
    croppingMaxRadius_: float = 20.0
    cropperName_: str = "MaxRadius"


@dataclasses.dataclass
class ScanProcessingParameters:  # Parameters.hpp:59-64
    downSamplingRatio_: float = 1.0
    voxelSize_: float = 0.03
    pointCloudBufferSize_: int = 1
    cropper_: ScanCroppingParameters = dataclasses.field(default_factory=ScanCroppingParameters)


@dataclasses.dataclass
class IcpParameters:  # Parameters.hpp:66-71
    maxNumIter_: int = 50
    maxCorrespondenceDistance_: float = 0.2
    knn_: int = 5
    maxDistanceKnn_: float = 10.0


@dataclasses.dataclass
class CloudRegistrationParameters:  # Parameters.hpp:73-76
    regType_: CloudRegistrationType = CloudRegistrationType.PointToPlaneIcp
    icp_: IcpParameters = dataclasses.field(default_factory=IcpParameters)


@dataclasses.dataclass
class OdometryParameters:  # Parameters.hpp:78-83
    scanMatcher_: CloudRegistrationParameters = dataclasses.field(default_factory=CloudRegistrationParameters)
    scanProcessing_: ScanProcessingParameters = dataclasses.field(default_factory=ScanProcessingParameters)
    isPublishOdometryMsgs_: bool = False
    odometryBufferSize_: int = 1


@dataclasses.dataclass
class SpaceCarvingParameters:  # Parameters.hpp:85-92
    voxelSize_: float = 0.1
    maxRaytracingLength_: float = 20.0
    truncationDistance_: float = 0.1
    carveSpaceEveryNscans_: int = 10
    minDotProductWithNormal_: float = 0.5
    neighborhoodRadiusDenseMap_: float = 0.1


@dataclasses.dataclass
class MapBuilderParameters:  # Parameters.hpp:94-98
    mapVoxelSize_: float = 0.03
    cropper_: ScanCroppingParameters = dataclasses.field(default_factory=ScanCroppingParameters)
    carving_: SpaceCarvingParameters = dataclasses.field(default_factory=SpaceCarvingParameters)


@dataclasses.dataclass
class ScanToMapRegistrationParameters:  # Parameters.hpp:145-149
    scanToMapRegType_: ScanToMapRegistrationType = ScanToMapRegistrationType.PointToPlaneIcp
    minRefinementFitness_: float = 0.7
    icp_: IcpParameters = dataclasses.field(default_factory=IcpParameters)


@dataclasses.dataclass
class MapperParameters:  # Parameters.hpp:158-178 (hot-path subset)
    scanMatcher_: ScanToMapRegistrationParameters = dataclasses.field(default_factory=ScanToMapRegistrationParameters)
    scanProcessing_: ScanProcessingParameters = dataclasses.field(default_factory=ScanProcessingParameters)
    minMovementBetweenMappingSteps_: float = 0.0
    isIgnoreMinRefinementFitness_: bool = False
    mapBuilder_: MapBuilderParameters = dataclasses.field(default_factory=MapBuilderParameters)
    denseMapBuilder_: MapBuilderParameters = dataclasses.field(default_factory=MapBuilderParameters)
    isBuildDenseMap_: bool = True
    isUseInitialMap_: bool = False
    isMergeScansIntoMap_: bool = True


def lua_default_mapper_parameters() -> MapperParameters:
    """The shipped Lua defaults for the hot-path knobs
    (ros/open3d_slam_ros/param/default/parameter_structure_definitions.lua:49-72,94-118), with the registration type
    set to point-to-plane (the shipped files select GeneralizedIcp, a 'next' row -- SURVEY.md 0.4)."""
    p = MapperParameters()
    p.scanMatcher_.icp_ = IcpParameters(maxNumIter_=50, maxCorrespondenceDistance_=1.0, knn_=20, maxDistanceKnn_=3.0)
    p.scanMatcher_.minRefinementFitness_ = 0.7
    p.scanProcessing_.voxelSize_ = 0.1
    p.scanProcessing_.downSamplingRatio_ = 1.0
    p.scanProcessing_.cropper_ = ScanCroppingParameters(croppingMinRadius_=2.0, croppingMaxRadius_=30.0, cropperName_="MinMaxRadius")
    p.mapBuilder_.mapVoxelSize_ = 0.1
    p.mapBuilder_.cropper_ = ScanCroppingParameters(croppingMinRadius_=2.0, croppingMaxRadius_=30.0, cropperName_="MinMaxRadius")
    return p


def identity() -> np.ndarray:
    return np.eye(4)
