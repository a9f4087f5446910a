// NOT Open3D: the members of open3d::geometry::PointCloud that cross the hot-path seams (see ../../Eigen/Dense for why this exists)
#pragma once
#include <vector>

#include "../../Eigen/Dense"
namespace open3d {
namespace geometry {
class PointCloud {
 public:
  virtual ~PointCloud() = default;  // open3d::geometry::Geometry is polymorphic
  std::vector<Eigen::Vector3d> points_, normals_, colors_;
  std::vector<Eigen::Matrix3d> covariances_;
  bool HasPoints() const { return !points_.empty(); }
  bool HasNormals() const { return !points_.empty() && normals_.size() == points_.size(); }
  bool HasColors() const { return !points_.empty() && colors_.size() == points_.size(); }
  bool IsEmpty() const { return !HasPoints(); }
};
}  // namespace geometry
}  // namespace open3d
