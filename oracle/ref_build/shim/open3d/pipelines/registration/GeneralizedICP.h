// NOT Open3D (see Registration.h)
#pragma once
#include "Registration.h"
namespace open3d {
namespace pipelines {
namespace registration {
class TransformationEstimationForGeneralizedICP : public TransformationEstimation {
 public:
  explicit TransformationEstimationForGeneralizedICP(double epsilon = 1e-3) : epsilon_(epsilon) {}
  double epsilon_ = 1e-3;
};
RegistrationResult RegistrationGeneralizedICP(const geometry::PointCloud& source, const geometry::PointCloud& target, double max_correspondence_distance,
                                              const Eigen::Matrix4d& init = Eigen::Matrix4d::Identity(),
                                              const TransformationEstimationForGeneralizedICP& estimation = TransformationEstimationForGeneralizedICP(),
                                              const ICPConvergenceCriteria& criteria = ICPConvergenceCriteria());
}  // namespace registration
}  // namespace pipelines
}  // namespace open3d
