// The Open3D ALGORITHMS that open3d_slam's glue calls -- VoxelDownSample, EstimateNormals (+ NormalizeNormals, OrientNormalsTowardsCamera-
// Location), RandomDownSample, RegistrationICP, RegistrationGeneralizedICP, GetInformationMatrixFromPointClouds -- SERVED BY THE CPU ORACLE's
// restatement (oracle/o3d_oracle.c, linked as libo3d_oracle.so), plus no-op stand-ins for the loop-closure classes (place recognition,
// FPFH), which are out of scope.  With these in place the reference's own Odometry.cpp / Mapper.cpp / ScanToMapRegistration.cpp /
// Submap.cpp / SubmapCollection.cpp run unchanged: what such a run pins is THEIR logic (what is cropped with which volume, which prior
// the scan matcher gets, when a scan is inserted, which gates apply) -- the algorithms underneath are the oracle's on both sides of any
// comparison and say nothing about Open3D.  Test infrastructure only.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../o3d_oracle.h"
#include "open3d_slam/PlaceRecognition.hpp"
#include "shim/open3d/Open3D.h"
#include "shim/open3d/pipelines/registration/Feature.h"
#include "shim/open3d/pipelines/registration/GeneralizedICP.h"
#include "shim/open3d/pipelines/registration/Registration.h"

namespace {
std::vector<double> flat(const std::vector<Eigen::Vector3d>& v) {
  std::vector<double> f(3 * v.size());
  for (size_t i = 0; i < v.size(); ++i)
    for (int a = 0; a < 3; ++a) f[3 * i + a] = v[i](a);
  return f;
}
void unflat(const std::vector<double>& f, size_t n, std::vector<Eigen::Vector3d>* v) {
  v->resize(n);
  for (size_t i = 0; i < n; ++i) (*v)[i] = Eigen::Vector3d(f[3 * i], f[3 * i + 1], f[3 * i + 2]);
}
void colmajor(const Eigen::Matrix4d& M, double T[16]) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) T[c * 4 + r] = M(r, c);
}
open3d::pipelines::registration::RegistrationResult result_of(const orc_icp_result& o) {
  open3d::pipelines::registration::RegistrationResult r;
  for (int c = 0; c < 4; ++c)
    for (int rr = 0; rr < 4; ++rr) r.transformation_(rr, c) = o.transformation[c * 4 + rr];
  r.fitness_ = o.fitness;
  r.inlier_rmse_ = o.inlier_rmse;
  return r;  // correspondence_set_ stays empty: no caller in open3d_slam reads it
}
[[noreturn]] void refuse(const char* what) {
  std::fprintf(stderr, "oracle/_ref: %s\n", what);
  std::abort();
}
}  // namespace

namespace open3d {
namespace geometry {
std::shared_ptr<PointCloud> PointCloud::VoxelDownSample(double voxel_size) const {
  auto out = std::make_shared<PointCloud>();
  const size_t n = points_.size();
  if (n == 0) return out;
  const std::vector<double> p = flat(points_);
  std::vector<double> op(3 * n), oa(3 * n);
  if (HasNormals()) {
    const std::vector<double> a = flat(normals_);
    const size_t m = orc_voxel_down_sample(p.data(), a.data(), n, voxel_size, op.data(), oa.data());
    unflat(op, m, &out->points_);
    unflat(oa, m, &out->normals_);
  } else {
    const size_t m = orc_voxel_down_sample(p.data(), nullptr, n, voxel_size, op.data(), nullptr);
    unflat(op, m, &out->points_);
  }
  if (HasColors()) {  // colours are averaged exactly as normals are (o3d_oracle.h)
    const std::vector<double> c = flat(colors_);
    const size_t m = orc_voxel_down_sample(p.data(), c.data(), n, voxel_size, op.data(), oa.data());
    unflat(oa, m, &out->colors_);
  }
  return out;
}
std::shared_ptr<PointCloud> PointCloud::RandomDownSample(double ratio) const {
  if (ratio < 1.0) refuse("RandomDownSample(ratio < 1) draws from std::random_device in Open3D: not reproducible, not served");
  std::vector<size_t> all(points_.size());
  for (size_t i = 0; i < all.size(); ++i) all[i] = i;
  return SelectByIndex(all);  // [O3D]: ratio 1 selects every index; SelectByIndex keeps cloud order
}
void PointCloud::EstimateNormals(const KDTreeSearchParam& p, bool) {
  const size_t n = points_.size();
  const std::vector<double> pts = flat(points_);
  std::vector<double> out(3 * n);
  if (const auto* h = dynamic_cast<const KDTreeSearchParamHybrid*>(&p)) {
    orc_estimate_normals(pts.data(), n, h->radius_, h->max_nn_, out.data());
  } else {
    refuse("EstimateNormals with a search parameter other than Hybrid is not on the path (CloudRegistration.cpp:24,52 use Hybrid)");
  }
  unflat(out, n, &normals_);
  normals_final_ = true;
}
PointCloud& PointCloud::NormalizeNormals() {
  if (normals_final_) return *this;
  for (auto& nn : normals_) {  // [O3D]: n.normalize(); NaN -> (0, 0, 1)
    nn.normalize();
    if (std::isnan(nn(0))) nn = Eigen::Vector3d(0.0, 0.0, 1.0);
  }
  return *this;
}
void PointCloud::OrientNormalsTowardsCameraLocation(const Eigen::Vector3d& camera) {
  if (normals_final_) return;
  for (size_t i = 0; i < points_.size() && i < normals_.size(); ++i) {  // [O3D]
    const Eigen::Vector3d to_cam = camera - points_[i];
    auto& nn = normals_[i];
    if (nn.norm() == 0.0) {
      nn = to_cam;
      if (nn.norm() == 0.0)
        nn = Eigen::Vector3d(0.0, 0.0, 1.0);
      else
        nn.normalize();
    } else if (nn.dot(to_cam) < 0.0) {
      nn = nn * -1.0;
    }
  }
}
void PointCloud::EstimateCovariances(const KDTreeSearchParam&) { refuse("EstimateCovariances is not on the path"); }
}  // namespace geometry

namespace pipelines {
namespace registration {
RegistrationResult RegistrationICP(const geometry::PointCloud& source, const geometry::PointCloud& target, double max_corr, const Eigen::Matrix4d& init,
                                   const TransformationEstimation& estimation, const ICPConvergenceCriteria& c) {
  const std::vector<double> s = flat(source.points_), t = flat(target.points_);
  double T[16];
  colmajor(init, T);
  orc_icp_result o;
  if (dynamic_cast<const TransformationEstimationPointToPlane*>(&estimation)) {
    if (!target.HasNormals()) refuse("RegistrationICP point-to-plane: the target has no normals ([O3D] raises an error)");
    const std::vector<double> tn = flat(target.normals_);
    orc_icp_point_to_plane(s.data(), source.points_.size(), t.data(), tn.data(), target.points_.size(), nullptr, max_corr, T, c.max_iteration_,
                           c.relative_fitness_, c.relative_rmse_, &o);
  } else if (dynamic_cast<const TransformationEstimationPointToPoint*>(&estimation)) {
    orc_icp_point_to_point(s.data(), source.points_.size(), t.data(), target.points_.size(), nullptr, max_corr, T, c.max_iteration_, c.relative_fitness_,
                           c.relative_rmse_, &o);
  } else {
    refuse("RegistrationICP: unknown estimation");
  }
  return result_of(o);
}
RegistrationResult RegistrationGeneralizedICP(const geometry::PointCloud& source, const geometry::PointCloud& target, double max_corr,
                                              const Eigen::Matrix4d& init, const TransformationEstimationForGeneralizedICP& e,
                                              const ICPConvergenceCriteria& c) {
  if (!source.HasNormals() || !target.HasNormals()) refuse("RegistrationGeneralizedICP: both clouds carry normals whenever open3d_slam reaches it");
  const std::vector<double> s = flat(source.points_), sn = flat(source.normals_), t = flat(target.points_), tn = flat(target.normals_);
  double T[16];
  colmajor(init, T);
  orc_icp_result o;
  orc_icp_generalized(s.data(), sn.data(), source.points_.size(), t.data(), tn.data(), target.points_.size(), nullptr, max_corr, T, c.max_iteration_,
                      c.relative_fitness_, c.relative_rmse_, e.epsilon_, &o);
  return result_of(o);
}
RegistrationResult EvaluateRegistration(const geometry::PointCloud&, const geometry::PointCloud&, double, const Eigen::Matrix4d&) {
  refuse("EvaluateRegistration is not on the path");
}
Eigen::Matrix<double, 6, 6> GetInformationMatrixFromPointClouds(const geometry::PointCloud& source, const geometry::PointCloud& target, double max_corr,
                                                                const Eigen::Matrix4d& transformation) {
  const std::vector<double> s = flat(source.points_), t = flat(target.points_);
  double T[16], out[36];
  colmajor(transformation, T);
  orc_information_matrix(s.data(), source.points_.size(), t.data(), target.points_.size(), nullptr, max_corr, T, out);
  Eigen::Matrix<double, 6, 6> M;
  for (int r = 0; r < 6; ++r)
    for (int cc = 0; cc < 6; ++cc) M(r, cc) = out[r * 6 + cc];
  return M;
}
std::shared_ptr<Feature> ComputeFPFHFeature(const geometry::PointCloud&, const geometry::KDTreeSearchParam&) {
  refuse("ComputeFPFHFeature (place recognition) is out of scope and not served");
}
}  // namespace registration
}  // namespace pipelines
namespace utility {
std::vector<std::string> SplitString(const std::string& s, const std::string& delimiters, bool trim_empty_str) {
  std::vector<std::string> out;
  size_t pos = 0;
  while (pos <= s.size()) {
    const size_t e = s.find_first_of(delimiters, pos);
    const std::string tok = s.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
    if (!tok.empty() || !trim_empty_str) out.push_back(tok);
    if (e == std::string::npos) break;
    pos = e + 1;
  }
  return out;
}
}  // namespace utility
}  // namespace open3d

// loop closure is out of scope: PlaceRecognition.cpp (FPFH + RANSAC from Open3D) is not compiled; the mapper may hold one
namespace o3d_slam {
PlaceRecognition::PlaceRecognition() {}
void PlaceRecognition::setParameters(const MapperParameters& p) { params_ = p; }
void PlaceRecognition::setFolderPath(const std::string& folderPath) { folderPath_ = folderPath; }
Constraints PlaceRecognition::buildLoopClosureConstraints(const Transform&, const SubmapCollection&, const AdjacencyMatrix&, size_t, size_t,
                                                          const Time&) const {
  return Constraints();
}
}  // namespace o3d_slam
