// NOT Open3D: RegistrationResult / ICPConvergenceCriteria as open3d_slam reads them (see ../../../Eigen/Dense for why this exists)
#pragma once
#include "../../../Eigen/Dense"
namespace open3d {
namespace pipelines {
namespace registration {
class RegistrationResult {
 public:
  RegistrationResult() = default;
  RegistrationResult(const Eigen::Matrix4d& T) : transformation_(T) {}
  Eigen::Matrix4d transformation_;
  double fitness_ = 0.0, inlier_rmse_ = 0.0;
};
class ICPConvergenceCriteria {
 public:
  double relative_fitness_ = 1e-6, relative_rmse_ = 1e-6;
  int max_iteration_ = 30;
};
}  // namespace registration
}  // namespace pipelines
}  // namespace open3d
