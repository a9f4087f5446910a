// o3ds_standalone_types.hpp -- minimal stand-ins for the Open3D / Eigen types that cross the reference's hot-path seams,
// so that o3ds_adapter.hpp compiles and is testable in an image without Open3D/Eigen.  Inside open3d_slam define
// O3DS_USE_OPEN3D and the adapter uses the real open3d::geometry::PointCloud / Eigen::Isometry3d instead
// (identical memory layout: std::vector<Eigen::Vector3d> is a contiguous double[3n]; Matrix4d is column-major).
#pragma once
#include <array>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <ratio>
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
  std::array<double, 3> translation() const { return {{m[12], m[13], m[14]}}; }
  Transform operator*(const Transform& o) const {  // Isometry3d composition: (*this) * o
    Transform r;
    for (int c = 0; c < 4; ++c)
      for (int row = 0; row < 4; ++row) {
        double s = 0.0;
        for (int k = 0; k < 4; ++k) s += m[k * 4 + row] * o.m[c * 4 + k];
        r.m[c * 4 + row] = s;
      }
    return r;
  }
};
// time.hpp:40-49: the 100-ns universal time scale the reference stamps scans with
struct UniversalTimeScaleClock {
  using rep = int64_t;
  using period = std::ratio<1, 10000000>;
  using duration = std::chrono::duration<rep, period>;
  using time_point = std::chrono::time_point<UniversalTimeScaleClock>;
  static constexpr bool is_steady = true;
};
using Time = UniversalTimeScaleClock::time_point;
}  // namespace o3d_slam
