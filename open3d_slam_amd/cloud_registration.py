"""CloudRegistration seam of the reference (include/open3d_slam/CloudRegistration.hpp:19-42,
src/CloudRegistration.cpp:44-65,85-100) over the HIP backend: same class names, members, factory and error behaviour."""
from __future__ import annotations

import dataclasses

import numpy as np

from . import backend as _b
from .parameters import CloudRegistrationParameters, CloudRegistrationType
from .pointcloud import PointCloud


@dataclasses.dataclass
class RegistrationResult:
    """open3d::pipelines::registration::RegistrationResult fields open3d_slam reads
    (Odometry.cpp:51-72, Mapper.cpp:151-159); correspondence_set_ is never consumed and is not materialised."""
    transformation_: np.ndarray
    fitness_: float
    inlier_rmse_: float
    iterations_: int = 0
    converged_: bool = False


@dataclasses.dataclass
class ICPConvergenceCriteria:
    """[O3D] defaults; the reference only overrides max_iteration_ (CloudRegistration.cpp:63)."""
    relative_fitness_: float = 1e-6
    relative_rmse_: float = 1e-6
    max_iteration_: int = 30


class CloudRegistration:  # CloudRegistration.hpp:19-27
    def registerClouds(self, source: PointCloud, target: PointCloud, init) -> RegistrationResult:
        raise NotImplementedError

    def estimateNormalsOrCovariancesIfNeeded(self, cloud: PointCloud) -> None:
        return None


class RegistrationIcpPointToPlane(CloudRegistration):  # CloudRegistration.hpp:29-42
    def __init__(self):
        self.maxCorrespondenceDistance_ = 1.0
        self.knnNormalEstimation_ = 10
        self.maxRadiusNormalEstimation_ = 2.0
        self.icpConvergenceCriteria_ = ICPConvergenceCriteria()

    def registerClouds(self, source: PointCloud, target: PointCloud, init, target_crop=None) -> RegistrationResult:
        """CloudRegistration.cpp:44-48 -> [O3D] RegistrationICP(source, target, maxCorrespondenceDistance_, init,
        TransformationEstimationPointToPlane, icpConvergenceCriteria_).  Raises (like Open3D's LogError -> runtime_error)
        on max_correspondence_distance <= 0 or a target without normals."""
        c = self.icpConvergenceCriteria_
        try:
            r = source.be.icp_point_to_plane_dev(source.id, target.id, self.maxCorrespondenceDistance_, init=init,
                                                 max_iter=c.max_iteration_, rel_fitness=c.relative_fitness_,
                                                 rel_rmse=c.relative_rmse_, target_crop=target_crop)
        except _b.BackendError as e:
            raise RuntimeError(str(e)) from e
        return RegistrationResult(r["transformation"], r["fitness"], r["inlier_rmse"], r["iterations"], r["converged"])

    def estimateNormalsOrCovariancesIfNeeded(self, cloud: PointCloud) -> None:
        """CloudRegistration.cpp:49-56: assert_gt on both parameters, then EstimateNormals(Hybrid) + NormalizeNormals +
        OrientNormalsTowardsCameraLocation."""
        if not self.maxRadiusNormalEstimation_ > 0.0:
            raise RuntimeError("maxRadiusNormalEstimation_")  # assert_gt (assert.hpp)
        if not self.knnNormalEstimation_ > 0:
            raise RuntimeError("knnNormalEstimation_")
        cloud.be.estimate_normals(cloud.id, self.maxRadiusNormalEstimation_, self.knnNormalEstimation_)


class RegistrationIcpGeneralized(CloudRegistration):  # CloudRegistration.hpp:56-69
    def __init__(self):
        self.maxCorrespondenceDistance_ = 1.0
        self.knnNormalEstimation_ = 10
        self.maxRadiusNormalEstimation_ = 2.0
        self.icpConvergenceCriteria_ = ICPConvergenceCriteria()

    def registerClouds(self, source: PointCloud, target: PointCloud, init, target_crop=None) -> RegistrationResult:
        """CloudRegistration.cpp:16-21 -> [O3D] RegistrationGeneralizedICP(source, target, maxCorrespondenceDistance_, init,
        TransformationEstimationForGeneralizedICP(), icpConvergenceCriteria_); covariances are built from the clouds' normals."""
        c = self.icpConvergenceCriteria_
        try:
            r = source.be.icp_generalized_dev(source.id, target.id, self.maxCorrespondenceDistance_, init=init, max_iter=c.max_iteration_,
                                              rel_fitness=c.relative_fitness_, rel_rmse=c.relative_rmse_, target_crop=target_crop)
        except _b.BackendError as e:
            raise RuntimeError(str(e)) from e
        return RegistrationResult(r["transformation"], r["fitness"], r["inlier_rmse"], r["iterations"], r["converged"])

    def estimateNormalsOrCovariancesIfNeeded(self, cloud: PointCloud) -> None:  # CloudRegistration.cpp:22-30
        if not self.maxRadiusNormalEstimation_ > 0.0:
            raise RuntimeError("maxRadiusNormalEstimation_")
        if not self.knnNormalEstimation_ > 0:
            raise RuntimeError("knnNormalEstimation_")
        cloud.be.estimate_normals(cloud.id, self.maxRadiusNormalEstimation_, self.knnNormalEstimation_)


class RegistrationIcpPointToPoint(CloudRegistration):  # CloudRegistration.hpp:30-41
    def __init__(self):
        self.maxCorrespondenceDistance_ = 1.0
        self.icpConvergenceCriteria_ = ICPConvergenceCriteria()

    def registerClouds(self, source: PointCloud, target: PointCloud, init, target_crop=None) -> RegistrationResult:
        """CloudRegistration.cpp:69-74 -> [O3D] RegistrationICP(source, target, maxCorrespondenceDistance_, init,
        TransformationEstimationPointToPoint(), icpConvergenceCriteria_): closed-form Umeyama update per iteration."""
        c = self.icpConvergenceCriteria_
        try:
            r = source.be.icp_point_to_point_dev(source.id, target.id, self.maxCorrespondenceDistance_, init=init, max_iter=c.max_iteration_,
                                                 rel_fitness=c.relative_fitness_, rel_rmse=c.relative_rmse_, target_crop=target_crop)
        except _b.BackendError as e:
            raise RuntimeError(str(e)) from e
        return RegistrationResult(r["transformation"], r["fitness"], r["inlier_rmse"], r["iterations"], r["converged"])

    def estimateNormalsOrCovariancesIfNeeded(self, cloud: PointCloud) -> None:  # base-class no-op (CloudRegistration.hpp:26)
        return None


def createPointToPointIcp(p: CloudRegistrationParameters) -> RegistrationIcpPointToPoint:  # CloudRegistration.cpp:76-81
    ret = RegistrationIcpPointToPoint()
    ret.maxCorrespondenceDistance_ = p.icp_.maxCorrespondenceDistance_
    ret.icpConvergenceCriteria_.max_iteration_ = p.icp_.maxNumIter_
    return ret


def createGeneralizedIcp(p: CloudRegistrationParameters) -> RegistrationIcpGeneralized:  # CloudRegistration.cpp:32-39
    ret = RegistrationIcpGeneralized()
    ret.maxCorrespondenceDistance_ = p.icp_.maxCorrespondenceDistance_
    ret.knnNormalEstimation_ = p.icp_.knn_
    ret.maxRadiusNormalEstimation_ = p.icp_.maxDistanceKnn_
    ret.icpConvergenceCriteria_.max_iteration_ = p.icp_.maxNumIter_
    return ret


def createPointToPlaneIcp(p: CloudRegistrationParameters) -> RegistrationIcpPointToPlane:  # CloudRegistration.cpp:58-65
    ret = RegistrationIcpPointToPlane()
    ret.maxCorrespondenceDistance_ = p.icp_.maxCorrespondenceDistance_
    ret.knnNormalEstimation_ = p.icp_.knn_
    ret.maxRadiusNormalEstimation_ = p.icp_.maxDistanceKnn_
    ret.icpConvergenceCriteria_.max_iteration_ = p.icp_.maxNumIter_
    return ret


def cloudRegistrationFactory(p: CloudRegistrationParameters) -> CloudRegistration:  # CloudRegistration.cpp:85-100
    if p.regType_ == CloudRegistrationType.PointToPlaneIcp:
        return createPointToPlaneIcp(p)
    if p.regType_ == CloudRegistrationType.GeneralizedIcp:
        return createGeneralizedIcp(p)
    if p.regType_ == CloudRegistrationType.PointToPointIcp:
        return createPointToPointIcp(p)
    raise RuntimeError("cloud: unknown type of cloud registration")
