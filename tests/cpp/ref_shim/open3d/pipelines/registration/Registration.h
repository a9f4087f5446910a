// NOT Open3D: open3d::pipelines::registration (v0.15.1) as open3d_slam's mapping sources spell it (see ../../../Eigen/eigen_shim.hpp)
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
  RegistrationResult(const Eigen::Matrix4d& T = Eigen::Matrix4d::Identity());
  Eigen::Matrix4d transformation_;
  CorrespondenceSet correspondence_set_;
  double fitness_ = 0.0, inlier_rmse_ = 0.0;
};
class ICPConvergenceCriteria {
 public:
  ICPConvergenceCriteria(double relative_fitness = 1e-6, double relative_rmse = 1e-6, int max_iteration = 30);
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
  explicit TransformationEstimationPointToPoint(bool with_scaling = false);
};
class TransformationEstimationPointToPlane : public TransformationEstimation {
 public:
  TransformationEstimationPointToPlane();
  explicit TransformationEstimationPointToPlane(std::shared_ptr<RobustKernel>);
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
