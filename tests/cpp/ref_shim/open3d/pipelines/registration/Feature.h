// NOT Open3D (see ../../../Eigen/eigen_shim.hpp)
#pragma once
#include <memory>

#include "../../geometry/PointCloud.h"
namespace open3d {
namespace pipelines {
namespace registration {
class Feature {
 public:
  Eigen::MatrixXd data_;
  void Resize(int dim, int n);
  size_t Dimension() const;
  size_t Num() const;
};
std::shared_ptr<Feature> ComputeFPFHFeature(const geometry::PointCloud& input, const geometry::KDTreeSearchParam& p = geometry::KDTreeSearchParamKNN());
}  // namespace registration
}  // namespace pipelines
}  // namespace open3d
