// NOT Open3D.  open3d::pipelines::registration (v0.15.1) as open3d_slam's sources spell it: the plain data types are written out, the
// ALGORITHMS (RegistrationICP, RegistrationGeneralizedICP, EvaluateRegistration, GetInformationMatrixFromPointClouds) are declared here and,
// in open3d_standin.cpp, SERVED BY THE CPU ORACLE's restatement (oracle/o3d_oracle.c) -- so that the reference's own glue around them
// (Odometry.cpp, Mapper.cpp, ScanToMapRegistration.cpp, Submap.cpp ...) can be run; nothing executed through them says anything about
// Open3D itself.
#pragma once
#include <memory>
#include <vector>

#include "../../geometry/PointCloud.h"
namespace open3d {
namespace pipelines {
namespace registration {
typedef std::vector<Eigen::Vector2i> CorrespondenceSet;
class RegistrationResult {
 public:
  RegistrationResult(const Eigen::Matrix4d& T = Eigen::Matrix4d::Identity()) : transformation_(T) {}
  Eigen::Matrix4d transformation_;
  CorrespondenceSet correspondence_set_;
  double fitness_ = 0.0, inlier_rmse_ = 0.0;
};
class ICPConvergenceCriteria {
 public:
  ICPConvergenceCriteria(double relative_fitness = 1e-6, double relative_rmse = 1e-6, int max_iteration = 30)
      : relative_fitness_(relative_fitness), relative_rmse_(relative_rmse), max_iteration_(max_iteration) {}
  double relative_fitness_, relative_rmse_;
  int max_iteration_;
};
class RobustKernel;
class TransformationEstimation {
 public:
  virtual ~TransformationEstimation() = default;
};
class TransformationEstimationPointToPoint : public TransformationEstimation {
 public:
  explicit TransformationEstimationPointToPoint(bool with_scaling = false) : with_scaling_(with_scaling) {}
  bool with_scaling_;
};
class TransformationEstimationPointToPlane : public TransformationEstimation {
 public:
  TransformationEstimationPointToPlane() {}
  explicit TransformationEstimationPointToPlane(std::shared_ptr<RobustKernel>) {}
};
RegistrationResult RegistrationICP(const geometry::PointCloud& source, const geometry::PointCloud& target, double max_correspondence_distance,
                                   const Eigen::Matrix4d& init = Eigen::Matrix4d::Identity(),
                                   const TransformationEstimation& estimation = TransformationEstimationPointToPoint(false),
                                   const ICPConvergenceCriteria& criteria = ICPConvergenceCriteria());
RegistrationResult EvaluateRegistration(const geometry::PointCloud& source, const geometry::PointCloud& target, double max_correspondence_distance,
                                        const Eigen::Matrix4d& transformation = Eigen::Matrix4d::Identity());
Eigen::Matrix<double, 6, 6> GetInformationMatrixFromPointClouds(const geometry::PointCloud& source, const geometry::PointCloud& target,
                                                                double max_correspondence_distance, const Eigen::Matrix4d& transformation);
}  // namespace registration
}  // namespace pipelines
}  // namespace open3d
