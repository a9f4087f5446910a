// o3ds_standalone_types.hpp -- minimal stand-ins for the Open3D / Eigen types that cross the reference's hot-path seams,
// so that o3ds_adapter.hpp compiles and is testable in an image without Open3D/Eigen.  Inside open3d_slam define
// O3DS_USE_OPEN3D and the adapter uses the real open3d::geometry::PointCloud / Eigen::Isometry3d instead
// (identical memory layout: std::vector<Eigen::Vector3d> is a contiguous double[3n]; Matrix4d is column-major).
#pragma once
#include <array>
#include <cstddef>
#include <vector>

namespace open3d {
namespace geometry {
struct PointCloud {  // open3d::geometry::PointCloud members used on this path (typedefs.hpp:24)
  std::vector<std::array<double, 3>> points_;
  std::vector<std::array<double, 3>> normals_;
  std::vector<std::array<double, 3>> colors_;
  bool HasPoints() const { return !points_.empty(); }
  bool HasNormals() const { return !points_.empty() && normals_.size() == points_.size(); }
  bool HasColors() const { return !points_.empty() && colors_.size() == points_.size(); }
  bool IsEmpty() const { return points_.empty(); }
};
}  // namespace geometry
namespace pipelines {
namespace registration {
struct RegistrationResult {  // fields open3d_slam reads (Odometry.cpp:51-72, Mapper.cpp:151-159)
  std::array<double, 16> transformation_{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};  // column-major
  double fitness_ = 0.0;
  double inlier_rmse_ = 0.0;
};
struct ICPConvergenceCriteria {
  double relative_fitness_ = 1e-6;
  double relative_rmse_ = 1e-6;
  int max_iteration_ = 30;
};
}  // namespace registration
}  // namespace pipelines
}  // namespace open3d

namespace o3d_slam {
struct Transform {  // stand-in for Eigen::Isometry3d (Transform.hpp:15): column-major 4x4
  std::array<double, 16> m{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
  static Transform Identity() { return Transform(); }
  const double* data() const { return m.data(); }
  double tx() const { return m[12]; }
  double ty() const { return m[13]; }
  double tz() const { return m[14]; }
};
}  // namespace o3d_slam
