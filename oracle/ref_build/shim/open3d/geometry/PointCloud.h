// NOT Open3D.  open3d::geometry::PointCloud (v0.15.1) as a plain container with the members open3d_slam's own sources touch, so that
// those sources can be compiled and run (see ../../Eigen/mini_eigen.hpp, ../../../README.md).  The trivial members are written out
// ([O3D] marks what restates Open3D behaviour from its documentation / SURVEY.md App. A); the ALGORITHMS (VoxelDownSample,
// EstimateNormals, registration) are declared here and SERVED BY THE CPU ORACLE's restatement (open3d_served_by_oracle.cpp), so that
// the reference's glue around them can run; KD-tree searches and a RandomDownSample that actually drops points abort.  They are not part
// of /root/reference, so nothing executed from this build can say anything about Open3D itself.
#pragma once
#include <memory>
#include <tuple>
#include <vector>

#include "../../Eigen/mini_eigen.hpp"
namespace open3d {
namespace geometry {
class KDTreeSearchParam {
 public:
  virtual ~KDTreeSearchParam() = default;
};
class KDTreeSearchParamKNN : public KDTreeSearchParam {
 public:
  explicit KDTreeSearchParamKNN(int knn = 30) : knn_(knn) {}
  int knn_;
};
class KDTreeSearchParamRadius : public KDTreeSearchParam {
 public:
  explicit KDTreeSearchParamRadius(double radius) : radius_(radius) {}
  double radius_;
};
class KDTreeSearchParamHybrid : public KDTreeSearchParam {
 public:
  KDTreeSearchParamHybrid(double radius, int max_nn) : radius_(radius), max_nn_(max_nn) {}
  double radius_;
  int max_nn_;
};
class PointCloud {
 public:
  PointCloud() {}
  explicit PointCloud(const std::vector<Eigen::Vector3d>& points) : points_(points) {}
  virtual ~PointCloud() = default;
  std::vector<Eigen::Vector3d> points_, normals_, colors_;
  std::vector<Eigen::Matrix3d> covariances_;
  // set by this stand-in's EstimateNormals: the oracle's routine it delegates to already normalises and orients (it restates the three
  // calls the reference always makes together, CloudRegistration.cpp:25-27,53-55), so the two follow-up calls then leave the normals alone
  bool normals_final_ = false;
  bool HasPoints() const { return points_.size() > 0; }
  bool HasNormals() const { return points_.size() > 0 && normals_.size() == points_.size(); }
  bool HasColors() const { return points_.size() > 0 && colors_.size() == points_.size(); }
  bool HasCovariances() const { return !points_.empty() && covariances_.size() == points_.size(); }
  bool IsEmpty() const { return !HasPoints(); }
  PointCloud& Clear() {
    points_.clear(), normals_.clear(), colors_.clear(), covariances_.clear();
    return *this;
  }
  PointCloud& Transform(const Eigen::Matrix4d& T);                                   // [O3D] TransformPoints / Normals / Covariances
  PointCloud& operator+=(const PointCloud& cloud);                                   // [O3D] PointCloud::operator+=
  PointCloud& RemoveNonFinitePoints(bool remove_nan = true, bool remove_infinite = true);  // [O3D]
  std::shared_ptr<PointCloud> SelectByIndex(const std::vector<size_t>& indices, bool invert = false) const;  // [O3D]
  Eigen::Vector3d GetMinBound() const;
  Eigen::Vector3d GetMaxBound() const;
  Eigen::Vector3d GetCenter() const;  // [O3D] ComputeCenter: mean of the points (zero for an empty cloud)
  // algorithms: NOT available (abort)
  std::shared_ptr<PointCloud> VoxelDownSample(double voxel_size) const;
  std::shared_ptr<PointCloud> RandomDownSample(double ratio) const;
  void EstimateNormals(const KDTreeSearchParam& p = KDTreeSearchParamKNN(), bool fast = true);
  void EstimateCovariances(const KDTreeSearchParam& p = KDTreeSearchParamKNN());
  void OrientNormalsTowardsCameraLocation(const Eigen::Vector3d& camera = Eigen::Vector3d::Zero());
  PointCloud& NormalizeNormals();
};
}  // namespace geometry
}  // namespace open3d
