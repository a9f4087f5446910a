// NOT Open3D: the FPFH feature type Submap.hpp names; ComputeFPFHFeature (place recognition, out of scope) aborts when called
#pragma once
#include <memory>

#include "../../geometry/PointCloud.h"
namespace open3d {
namespace pipelines {
namespace registration {
class Feature {
 public:
  std::vector<double> data_;
  int dim_ = 0, num_ = 0;
  void Resize(int dim, int n) {
    dim_ = dim, num_ = n;
    data_.assign((size_t)dim * n, 0.0);
  }
  size_t Dimension() const { return (size_t)dim_; }
  size_t Num() const { return (size_t)num_; }
};
std::shared_ptr<Feature> ComputeFPFHFeature(const geometry::PointCloud& input, const geometry::KDTreeSearchParam& p = geometry::KDTreeSearchParamKNN());
}  // namespace registration
}  // namespace pipelines
}  // namespace open3d
