// NOT Open3D: declarations with the spelling of open3d::geometry::PointCloud (v0.15.1) as open3d_slam's mapping sources use it
// (see ../../Eigen/eigen_shim.hpp for why this exists)
#pragma once
#include <memory>
#include <tuple>
#include <vector>

#include "../../Eigen/eigen_shim.hpp"
namespace open3d {
namespace geometry {
class KDTreeSearchParam {
 public:
  virtual ~KDTreeSearchParam() = default;
};
class KDTreeSearchParamKNN : public KDTreeSearchParam {
 public:
  explicit KDTreeSearchParamKNN(int knn = 30);
};
class KDTreeSearchParamRadius : public KDTreeSearchParam {
 public:
  explicit KDTreeSearchParamRadius(double radius);
};
class KDTreeSearchParamHybrid : public KDTreeSearchParam {
 public:
  KDTreeSearchParamHybrid(double radius, int max_nn);
};
class AxisAlignedBoundingBox {
 public:
  AxisAlignedBoundingBox();
  AxisAlignedBoundingBox(const Eigen::Vector3d& mn, const Eigen::Vector3d& mx);
  Eigen::Vector3d min_bound_, max_bound_;
  Eigen::Vector3d GetCenter() const;
  Eigen::Vector3d GetExtent() const;
};
class PointCloud {
 public:
  PointCloud();
  explicit PointCloud(const std::vector<Eigen::Vector3d>& points);
  virtual ~PointCloud() = default;  // open3d::geometry::Geometry is polymorphic (virtual destructor, Clear, IsEmpty, ...)
  std::vector<Eigen::Vector3d> points_, normals_, colors_;
  std::vector<Eigen::Matrix3d> covariances_;
  bool HasPoints() const;
  bool HasNormals() const;
  bool HasColors() const;
  bool HasCovariances() const;
  bool IsEmpty() const;
  PointCloud& Clear();
  PointCloud& Transform(const Eigen::Matrix4d&);
  PointCloud& Translate(const Eigen::Vector3d&, bool relative = true);
  PointCloud& operator+=(const PointCloud&);
  PointCloud operator+(const PointCloud&) const;
  PointCloud& NormalizeNormals();
  PointCloud& PaintUniformColor(const Eigen::Vector3d&);
  PointCloud& RemoveNonFinitePoints(bool remove_nan = true, bool remove_infinite = true);
  std::shared_ptr<PointCloud> SelectByIndex(const std::vector<size_t>&, bool invert = false) const;
  std::shared_ptr<PointCloud> VoxelDownSample(double voxel_size) const;
  std::shared_ptr<PointCloud> RandomDownSample(double ratio) const;
  std::shared_ptr<PointCloud> UniformDownSample(size_t every_k) const;
  std::shared_ptr<PointCloud> Crop(const AxisAlignedBoundingBox&) const;
  std::tuple<std::shared_ptr<PointCloud>, std::vector<size_t>> RemoveStatisticalOutliers(size_t, double) const;
  std::tuple<std::shared_ptr<PointCloud>, std::vector<size_t>> RemoveRadiusOutliers(size_t, double) const;
  void EstimateNormals(const KDTreeSearchParam& p = KDTreeSearchParamKNN(), bool fast = true);
  void EstimateCovariances(const KDTreeSearchParam& p = KDTreeSearchParamKNN());
  void OrientNormalsTowardsCameraLocation(const Eigen::Vector3d& camera = Eigen::Vector3d::Zero());
  Eigen::Vector3d GetMinBound() const;
  Eigen::Vector3d GetMaxBound() const;
  Eigen::Vector3d GetCenter() const;
  AxisAlignedBoundingBox GetAxisAlignedBoundingBox() const;
};
}  // namespace geometry
}  // namespace open3d
