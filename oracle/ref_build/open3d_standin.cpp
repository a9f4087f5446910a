// Definitions behind oracle/ref_build/shim/open3d/*: the container members of open3d::geometry::PointCloud that open3d_slam's own
// sources call, restated from Open3D v0.15.1's documented behaviour ([O3D]), and ABORTING stubs for the KD-tree searches (the
// algorithms the reference's glue calls are in open3d_served_by_oracle.cpp).  Test infrastructure only.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "shim/open3d/Open3D.h"

namespace open3d {
namespace geometry {
namespace {
[[noreturn]] void not_here(const char* what) {
  std::fprintf(stderr, "oracle/_ref: %s is an Open3D algorithm; Open3D is not part of /root/reference and is not in this build\n", what);
  std::abort();
}
}  // namespace

// [O3D] Geometry3D::TransformPoints / TransformNormals / TransformCovariances
PointCloud& PointCloud::Transform(const Eigen::Matrix4d& T) {
  for (auto& p : points_) {
    const Eigen::Vector4d q = T * Eigen::Vector4d(p(0), p(1), p(2), 1.0);
    p = q.head<3>() / q(3);
  }
  for (auto& n : normals_) {
    const Eigen::Vector4d q = T * Eigen::Vector4d(n(0), n(1), n(2), 0.0);
    n = q.head<3>();
  }
  const Eigen::Matrix3d R = T.block<3, 3>(0, 0);
  for (auto& c : covariances_) c = R * c * R.transpose();
  return *this;
}

// [O3D] PointCloud::operator+=: an attribute survives only if both sides carry it (or this side was empty)
PointCloud& PointCloud::operator+=(const PointCloud& cloud) {
  if (cloud.IsEmpty()) return *this;
  const size_t old_n = points_.size(), add_n = cloud.points_.size(), new_n = old_n + add_n;
  if ((!HasPoints() || HasNormals()) && cloud.HasNormals()) {
    normals_.resize(new_n);
    for (size_t i = 0; i < add_n; ++i) normals_[old_n + i] = cloud.normals_[i];
  } else {
    normals_.clear();
  }
  if ((!HasPoints() || HasColors()) && cloud.HasColors()) {
    colors_.resize(new_n);
    for (size_t i = 0; i < add_n; ++i) colors_[old_n + i] = cloud.colors_[i];
  } else {
    colors_.clear();
  }
  if ((!HasPoints() || HasCovariances()) && cloud.HasCovariances()) {
    covariances_.resize(new_n);
    for (size_t i = 0; i < add_n; ++i) covariances_[old_n + i] = cloud.covariances_[i];
  } else {
    covariances_.clear();
  }
  points_.resize(new_n);
  for (size_t i = 0; i < add_n; ++i) points_[old_n + i] = cloud.points_[i];
  return *this;
}

// [O3D] PointCloud::RemoveNonFinitePoints
PointCloud& PointCloud::RemoveNonFinitePoints(bool remove_nan, bool remove_infinite) {
  const bool hn = HasNormals(), hc = HasColors(), hv = HasCovariances();
  const size_t old_n = points_.size();
  size_t k = 0;
  for (size_t i = 0; i < old_n; ++i) {
    const bool is_nan = remove_nan && (std::isnan(points_[i](0)) || std::isnan(points_[i](1)) || std::isnan(points_[i](2)));
    const bool is_inf = remove_infinite && (std::isinf(points_[i](0)) || std::isinf(points_[i](1)) || std::isinf(points_[i](2)));
    if (!is_nan && !is_inf) {
      points_[k] = points_[i];
      if (hn) normals_[k] = normals_[i];
      if (hc) colors_[k] = colors_[i];
      if (hv) covariances_[k] = covariances_[i];
      ++k;
    }
  }
  points_.resize(k);
  if (hn) normals_.resize(k);
  if (hc) colors_.resize(k);
  if (hv) covariances_.resize(k);
  return *this;
}

// [O3D] PointCloud::SelectByIndex
std::shared_ptr<PointCloud> PointCloud::SelectByIndex(const std::vector<size_t>& indices, bool invert) const {
  auto out = std::make_shared<PointCloud>();
  const bool hn = HasNormals(), hc = HasColors(), hv = HasCovariances();
  std::vector<bool> mask(points_.size(), invert);
  for (size_t i : indices) mask[i] = !invert;
  for (size_t i = 0; i < points_.size(); ++i)
    if (mask[i]) {
      out->points_.push_back(points_[i]);
      if (hn) out->normals_.push_back(normals_[i]);
      if (hc) out->colors_.push_back(colors_[i]);
      if (hv) out->covariances_.push_back(covariances_[i]);
    }
  return out;
}

Eigen::Vector3d PointCloud::GetMinBound() const {
  if (points_.empty()) return Eigen::Vector3d(0.0, 0.0, 0.0);
  Eigen::Vector3d m = points_[0];
  for (const auto& p : points_)
    for (int a = 0; a < 3; ++a) m(a) = std::min(m(a), p(a));
  return m;
}
Eigen::Vector3d PointCloud::GetMaxBound() const {
  if (points_.empty()) return Eigen::Vector3d(0.0, 0.0, 0.0);
  Eigen::Vector3d m = points_[0];
  for (const auto& p : points_)
    for (int a = 0; a < 3; ++a) m(a) = std::max(m(a), p(a));
  return m;
}

Eigen::Vector3d PointCloud::GetCenter() const {
  Eigen::Vector3d c(0.0, 0.0, 0.0);
  if (points_.empty()) return c;
  for (const auto& p : points_) c += p;
  return c / double(points_.size());
}

bool KDTreeFlann::SetGeometry(const PointCloud&) { not_here("KDTreeFlann::SetGeometry"); }
int KDTreeFlann::SearchKNN(const Eigen::Vector3d&, int, std::vector<int>&, std::vector<double>&) const { not_here("KDTreeFlann::SearchKNN"); }
int KDTreeFlann::SearchRadius(const Eigen::Vector3d&, double, std::vector<int>&, std::vector<double>&) const { not_here("KDTreeFlann::SearchRadius"); }
int KDTreeFlann::SearchHybrid(const Eigen::Vector3d&, double, int, std::vector<int>&, std::vector<double>&) const {
  not_here("KDTreeFlann::SearchHybrid");
}
}  // namespace geometry
}  // namespace open3d
