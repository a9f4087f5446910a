// o3ds_adapter.hpp -- C++ host side of the drop-in: the reference's own hot-path classes, re-implemented over the
// C-ABI of include/o3ds_backend.h.  Same names, members, argument meaning and error behaviour
// (std::runtime_error where the reference's assert_* / Open3D LogError throw) as:
//   CloudRegistration / RegistrationIcpPointToPlane / RegistrationIcpPointToPoint / RegistrationIcpGeneralized / cloudRegistrationFactory
//        include/open3d_slam/CloudRegistration.hpp:19-42, src/CloudRegistration.cpp:44-65,85-100
//   CroppingVolume family / croppingVolumeFactory      include/open3d_slam/croppers.hpp:26-47, src/croppers.cpp
//   voxelize / voxelizeWithinCroppingVolume / transform include/open3d_slam/helpers.hpp:20-25, src/helpers.cpp:107-183,273-305
//   DeviceSubmap: the submap resident in HBM            src/ScanToMapRegistration.cpp:55-62, src/Submap.cpp:39-75,94-125
//   (ScanToMapIcp, Submap, VoxelizedPointCloud and the remaining helpers over it: o3ds_mapping.hpp)
//   saveToFile                                          include/open3d_slam/output.hpp, src/output.cpp:39-47
// Header-only; link with -lo3ds_backend.  One backend handle per calling thread (thread_local) for the stateless calls, because the
// reference calls registerClouds concurrently from odometry / mapping / loop-closure threads (SlamWrapper.cpp:258-347,406-448);
// objects that keep device state (DeviceSubmap, VoxelizedPointCloud) own a handle of their own.
#pragma once
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/o3ds_backend.h"
#ifdef O3DS_USE_OPEN3D
#include <open3d/geometry/PointCloud.h>
#include <open3d/pipelines/registration/Registration.h>
#include "open3d_slam/Transform.hpp"
#include "open3d_slam/time.hpp"
#else
#include "o3ds_standalone_types.hpp"
#endif

namespace o3d_slam {

using PointCloud = open3d::geometry::PointCloud;
using PointCloudPtr = std::shared_ptr<PointCloud>;
using RegistrationResult = open3d::pipelines::registration::RegistrationResult;

// ---- Parameters (include/open3d_slam/Parameters.hpp:37-76,94-98; same names and defaults) -------------------------
enum class CloudRegistrationType : int { PointToPlaneIcp, PointToPointIcp, GeneralizedIcp };
struct ScanCroppingParameters {
  double croppingMinZ_ = -10.0, croppingMaxZ_ = 10.0, croppingMinRadius_ = 0.0, croppingMaxRadius_ = 20.0;
  std::string cropperName_ = "MaxRadius";
};
struct IcpParameters {
  int maxNumIter_ = 50;
  double maxCorrespondenceDistance_ = 0.2;
  int knn_ = 5;
  double maxDistanceKnn_ = 10.0;
};
struct CloudRegistrationParameters {
  CloudRegistrationType regType_ = CloudRegistrationType::PointToPlaneIcp;
  IcpParameters icp_;
};

namespace o3ds_detail {

inline const double* xyz(const std::vector<decltype(PointCloud::points_)::value_type>& v) {
  return v.empty() ? nullptr : reinterpret_cast<const double*>(v.data());
}
inline const double* pose_data(const Transform& T) {
#ifdef O3DS_USE_OPEN3D
  return T.matrix().data();
#else
  return T.data();
#endif
}

// RAII handle, one per thread.  Errors become std::runtime_error like the reference's assert_* (assert.hpp:13-64).
class Handle {
 public:
  static o3ds_handle get() {
    thread_local Handle h;
    return h.h_;
  }
  // o3ds_last_error(NULL) is the last error of the calling thread, whichever handle raised it
  static void check(int rc) {
    if (rc != O3DS_OK) throw std::runtime_error(std::string("o3ds: ") + o3ds_last_error(nullptr));
  }

 private:
  Handle() {
    const int rc = o3ds_create(0, &h_);
    if (rc != O3DS_OK) throw std::runtime_error(std::string("o3ds_create: ") + o3ds_last_error(nullptr));
  }
  ~Handle() {
    if (h_) o3ds_destroy(h_);
  }
  o3ds_handle h_ = nullptr;
};

// A handle that belongs to an OBJECT instead of a thread: device clouds belong to the handle that made them, and the reference's
// Submap is built on one thread, filled by mappingWorker, read by the dense-map and visualisation threads
// (SlamWrapper.cpp:290-347,363-386; Submap.cpp:69,187-190) -- so a device-resident map owns its handle (one stream + scratch) and
// the mutex the reference guards mapCloud_ with.  Created lazily: constructing the owner needs no device.
class OwnedHandle {
 public:
  OwnedHandle() = default;
  OwnedHandle(const OwnedHandle&) = delete;
  OwnedHandle& operator=(const OwnedHandle&) = delete;
  ~OwnedHandle() {
    if (h_) o3ds_destroy(h_);
  }
  o3ds_handle get() const {
    if (!h_) {
      const int rc = o3ds_create(0, &h_);
      if (rc != O3DS_OK) throw std::runtime_error(std::string("o3ds_create: ") + o3ds_last_error(nullptr));
    }
    return h_;
  }
  bool created() const { return h_ != nullptr; }

 private:
  mutable o3ds_handle h_ = nullptr;
};

// scoped device cloud (on the calling thread's handle unless another one is named)
class DevCloud {
 public:
  DevCloud() = default;
  explicit DevCloud(const PointCloud& c, o3ds_handle h = nullptr) : h_(h ? h : Handle::get()) {
    Handle::check(o3ds_cloud_upload(h_, xyz(c.points_), c.HasNormals() ? xyz(c.normals_) : nullptr, c.points_.size(), &id_));
    if (c.HasColors()) Handle::check(o3ds_cloud_set_colors(h_, id_, xyz(c.colors_)));  // ride along (never read by registration)
  }
  explicit DevCloud(o3ds_cloud id, o3ds_handle h = nullptr) : h_(h ? h : Handle::get()), id_(id) {}
  DevCloud(const DevCloud&) = delete;
  DevCloud& operator=(const DevCloud&) = delete;
  DevCloud(DevCloud&& o) noexcept : h_(o.h_), id_(o.id_) { o.id_ = 0; }
  DevCloud& operator=(DevCloud&& o) noexcept {
    reset();
    h_ = o.h_;
    id_ = o.id_;
    o.id_ = 0;
    return *this;
  }
  ~DevCloud() { reset(); }
  void reset() {
    if (id_) o3ds_cloud_free(h_, id_);
    id_ = 0;
  }
  o3ds_cloud id() const { return id_; }
  o3ds_handle handle() const { return h_; }
  o3ds_cloud release() {  // forget the id without freeing it (borrowed clouds)
    const o3ds_cloud i = id_;
    id_ = 0;
    return i;
  }
  size_t size() const {
    size_t n = 0;
    Handle::check(o3ds_cloud_size(h_, id_, &n, nullptr));
    return n;
  }
  bool hasNormals() const {
    size_t n = 0;
    int hn = 0;
    Handle::check(o3ds_cloud_size(h_, id_, &n, &hn));
    return hn != 0;
  }
  void download(PointCloud* out) const {
    size_t n = 0;
    int hn = 0;
    Handle::check(o3ds_cloud_size(h_, id_, &n, &hn));
    out->points_.resize(n);
    out->normals_.resize(hn ? n : 0);
    if (n)
      Handle::check(o3ds_cloud_download(h_, id_, reinterpret_cast<double*>(out->points_.data()),
                                        hn ? reinterpret_cast<double*>(out->normals_.data()) : nullptr, n));
    int hc = 0;
    Handle::check(o3ds_cloud_has_colors(h_, id_, &hc));
    out->colors_.resize(hc ? n : 0);
    if (n && hc) Handle::check(o3ds_cloud_get_colors(h_, id_, reinterpret_cast<double*>(out->colors_.data()), n));
  }

 private:
  o3ds_handle h_ = nullptr;
  o3ds_cloud id_ = 0;
};

}  // namespace o3ds_detail

// ---- croppers (croppers.hpp:26-47) ---------------------------------------------------------------------------------
enum class CroppingVolumeEnum : int { MaxRadius, MinRadius, Cylinder, MinMaxRadius };
static const std::map<std::string, CroppingVolumeEnum> cropperNames{{"MaxRadius", CroppingVolumeEnum::MaxRadius},
                                                                    {"MinRadius", CroppingVolumeEnum::MinRadius},
                                                                    {"Cylinder", CroppingVolumeEnum::Cylinder},
                                                                    {"MinMaxRadius", CroppingVolumeEnum::MinMaxRadius}};

class CroppingVolume {
 public:
  virtual ~CroppingVolume() = default;
  virtual void setScaling(double) {}
  void setIsInvertVolume(bool v) { isInvertVolume_ = v; }
  void setPose(const Transform& pose) { pose_ = pose; }
  // o3ds_crop of this volume: only pose_.translation() enters the predicate (croppers.cpp:121-165)
  o3ds_crop toAbi() const {
    o3ds_crop c{};
    fill(&c);
    c.invert = isInvertVolume_ ? 1 : 0;
    const double* m = o3ds_detail::pose_data(pose_);
    c.center[0] = m[12];
    c.center[1] = m[13];
    c.center[2] = m[14];
    return c;
  }
  // CroppingVolume::crop (croppers.cpp:76-106)
  std::shared_ptr<PointCloud> crop(const PointCloud& cloud) const {
    auto out = std::make_shared<PointCloud>();
    if (cloud.points_.empty()) return out;
    o3ds_detail::DevCloud in(cloud);
    const o3ds_crop c = toAbi();
    o3ds_cloud o = 0;
    o3ds_detail::Handle::check(o3ds_crop_cloud(o3ds_detail::Handle::get(), in.id(), &c, &o));
    o3ds_detail::DevCloud(o).download(out.get());
    return out;
  }
  void crop(PointCloud* cloud) const { *cloud = std::move(*crop(*cloud)); }

 protected:
  virtual void fill(o3ds_crop* c) const { c->kind = O3DS_CROP_NONE; }
  Transform pose_ = Transform::Identity();
  bool isInvertVolume_ = false;
};
class MinMaxRadiusCroppingVolume : public CroppingVolume {
 public:
  MinMaxRadiusCroppingVolume() = default;
  MinMaxRadiusCroppingVolume(double radiusMin, double radiusMax) : radiusMin_(radiusMin), radiusMax_(radiusMax) {}
  void setParameters(double radiusMin, double radiusMax) { radiusMin_ = radiusMin, radiusMax_ = radiusMax; }

 private:
  void fill(o3ds_crop* c) const final { c->kind = O3DS_CROP_MIN_MAX_RADIUS, c->rmin = radiusMin_, c->rmax = radiusMax_; }
  double radiusMin_ = 0.0, radiusMax_ = 1e4;
};
class MaxRadiusCroppingVolume : public CroppingVolume {
 public:
  MaxRadiusCroppingVolume() = default;
  explicit MaxRadiusCroppingVolume(double radius) : radius_(radius) {}
  void setParameters(double radius) { radius_ = radius; }

 private:
  void fill(o3ds_crop* c) const final { c->kind = O3DS_CROP_MAX_RADIUS, c->rmax = radius_; }
  double radius_ = 1e4;
};
class MinRadiusCroppingVolume : public CroppingVolume {
 public:
  MinRadiusCroppingVolume() = default;
  explicit MinRadiusCroppingVolume(double radius) : radius_(radius) {}
  void setParameters(double radius) { radius_ = radius; }

 private:
  void fill(o3ds_crop* c) const final { c->kind = O3DS_CROP_MIN_RADIUS, c->rmin = radius_; }
  double radius_ = 0.0;
};
class CylinderCroppingVolume : public CroppingVolume {
 public:
  CylinderCroppingVolume() = default;
  CylinderCroppingVolume(double radius, double minZ, double maxZ) : radius_(radius), minZ_(minZ), maxZ_(maxZ) {}
  void setParameters(double radius, double minZ, double maxZ) { radius_ = radius, minZ_ = minZ, maxZ_ = maxZ; }

 private:
  void fill(o3ds_crop* c) const final { c->kind = O3DS_CROP_CYLINDER, c->rmax = radius_, c->zmin = minZ_, c->zmax = maxZ_; }
  double radius_ = 1e4, minZ_ = -1e4, maxZ_ = 1e4;
};

inline std::unique_ptr<CroppingVolume> croppingVolumeFactory(CroppingVolumeEnum type, const ScanCroppingParameters& p) {  // croppers.cpp:23-47
  switch (type) {
    case CroppingVolumeEnum::Cylinder:
      return std::make_unique<CylinderCroppingVolume>(p.croppingMaxRadius_, p.croppingMinZ_, p.croppingMaxZ_);
    case CroppingVolumeEnum::MinRadius:
      return std::make_unique<MinRadiusCroppingVolume>(p.croppingMinRadius_);
    case CroppingVolumeEnum::MaxRadius:
      return std::make_unique<MaxRadiusCroppingVolume>(p.croppingMaxRadius_);
    case CroppingVolumeEnum::MinMaxRadius:
      return std::make_unique<MinMaxRadiusCroppingVolume>(p.croppingMinRadius_, p.croppingMaxRadius_);
    default:
      throw std::runtime_error("Unknown cropper type");
  }
}
inline std::unique_ptr<CroppingVolume> croppingVolumeFactory(const ScanCroppingParameters& p) {  // croppers.cpp:20-22
  return croppingVolumeFactory(cropperNames.at(p.cropperName_), p);
}

// ---- CloudRegistration seam (CloudRegistration.hpp:19-42) ----------------------------------------------------------
class CloudRegistration {
 public:
  using RegistrationResult = open3d::pipelines::registration::RegistrationResult;
  CloudRegistration() = default;
  virtual ~CloudRegistration() = default;
  virtual RegistrationResult registerClouds(const PointCloud& source, const PointCloud& target, const Transform& init) const = 0;
  virtual void estimateNormalsOrCovariancesIfNeeded(PointCloud* /*cloud*/) const {}
  // -- two hooks for the device-resident scan-to-map path (o3ds_mapping.hpp); not part of the reference interface.  A registration
  // the backend does not know returns false and the caller falls back to the two virtuals above on host clouds.
  virtual bool toAbi(o3ds_icp_params*) const { return false; }  // parameters + estimator for o3ds_icp_register_dev
  virtual bool estimateNormalsOrCovariancesIfNeededDev(o3ds_handle, o3ds_cloud) const { return false; }  // the same, in place on a device cloud
};

namespace o3ds_detail {
inline o3ds_icp_params icpParams(int method, double maxCorrespondenceDistance, const open3d::pipelines::registration::ICPConvergenceCriteria& c) {
  o3ds_icp_params p{};
  p.max_correspondence_distance = maxCorrespondenceDistance;
  p.max_iteration = c.max_iteration_;
  p.method = method;
  p.relative_fitness = c.relative_fitness_;
  p.relative_rmse = c.relative_rmse_;
  return p;
}
inline void checkNormalEstimationParameters(double maxRadius, int knn) {  // assert_gt x2, CloudRegistration.cpp:23-24,50-51
  if (!(maxRadius > 0.0)) throw std::runtime_error("maxRadiusNormalEstimation_");
  if (!(knn > 0)) throw std::runtime_error("knnNormalEstimation_");
}
}  // namespace o3ds_detail

class RegistrationIcpPointToPlane : public CloudRegistration {
 public:
  // CloudRegistration.cpp:44-48: RegistrationICP(source, target, maxCorrespondenceDistance_, init, pointToPlane_, criteria)
  RegistrationResult registerClouds(const PointCloud& source, const PointCloud& target, const Transform& init) const final {
    const o3ds_icp_params p = o3ds_detail::icpParams(O3DS_ICP_POINT_TO_PLANE, maxCorrespondenceDistance_, icpConvergenceCriteria_);
    o3ds_icp_result r{};
    o3ds_detail::Handle::check(o3ds_icp_point_to_plane(o3ds_detail::Handle::get(), o3ds_detail::xyz(source.points_), source.points_.size(),
                                                       o3ds_detail::xyz(target.points_),
                                                       target.HasNormals() ? o3ds_detail::xyz(target.normals_) : nullptr,
                                                       target.points_.size(), o3ds_detail::pose_data(init), &p, &r));
    return toResult(r);
  }
  // CloudRegistration.cpp:49-56
  void estimateNormalsOrCovariancesIfNeeded(PointCloud* cloud) const final {
    o3ds_detail::checkNormalEstimationParameters(maxRadiusNormalEstimation_, knnNormalEstimation_);
    if (cloud->points_.empty()) return;
    o3ds_detail::DevCloud d(*cloud);  // colours ride along; existing normals are overwritten, as [O3D] EstimateNormals does
    o3ds_detail::Handle::check(o3ds_estimate_normals(d.handle(), d.id(), maxRadiusNormalEstimation_, knnNormalEstimation_));
    d.download(cloud);
  }
  bool toAbi(o3ds_icp_params* p) const final {
    *p = o3ds_detail::icpParams(O3DS_ICP_POINT_TO_PLANE, maxCorrespondenceDistance_, icpConvergenceCriteria_);
    return true;
  }
  bool estimateNormalsOrCovariancesIfNeededDev(o3ds_handle h, o3ds_cloud c) const final {
    o3ds_detail::checkNormalEstimationParameters(maxRadiusNormalEstimation_, knnNormalEstimation_);
    o3ds_detail::Handle::check(o3ds_estimate_normals(h, c, maxRadiusNormalEstimation_, knnNormalEstimation_));
    return true;
  }
  static RegistrationResult toResult(const o3ds_icp_result& r) {
    RegistrationResult out;
#ifdef O3DS_USE_OPEN3D
    out.transformation_ = Eigen::Map<const Eigen::Matrix4d>(r.transformation);
#else
    for (int i = 0; i < 16; ++i) out.transformation_[i] = r.transformation[i];
#endif
    out.fitness_ = r.fitness;
    out.inlier_rmse_ = r.inlier_rmse;
    return out;
  }

  double maxCorrespondenceDistance_ = 1.0;
  int knnNormalEstimation_ = 10;
  double maxRadiusNormalEstimation_ = 2.0;
  open3d::pipelines::registration::ICPConvergenceCriteria icpConvergenceCriteria_;
};

// RegistrationIcpGeneralized (CloudRegistration.hpp:56-69, CloudRegistration.cpp:16-39): the estimator the shipped Lua configs select
class RegistrationIcpGeneralized : public CloudRegistration {
 public:
  RegistrationResult registerClouds(const PointCloud& source, const PointCloud& target, const Transform& init) const final {
    const o3ds_icp_params p = o3ds_detail::icpParams(O3DS_ICP_GENERALIZED, maxCorrespondenceDistance_, icpConvergenceCriteria_);
    o3ds_icp_result r{};
    o3ds_detail::Handle::check(o3ds_icp_generalized(o3ds_detail::Handle::get(), o3ds_detail::xyz(source.points_),
                                                    source.HasNormals() ? o3ds_detail::xyz(source.normals_) : nullptr, source.points_.size(),
                                                    o3ds_detail::xyz(target.points_),
                                                    target.HasNormals() ? o3ds_detail::xyz(target.normals_) : nullptr, target.points_.size(),
                                                    o3ds_detail::pose_data(init), &p, &r));
    return RegistrationIcpPointToPlane::toResult(r);
  }
  void estimateNormalsOrCovariancesIfNeeded(PointCloud* cloud) const final {  // CloudRegistration.cpp:22-30
    RegistrationIcpPointToPlane tmp;
    tmp.knnNormalEstimation_ = knnNormalEstimation_;
    tmp.maxRadiusNormalEstimation_ = maxRadiusNormalEstimation_;
    tmp.estimateNormalsOrCovariancesIfNeeded(cloud);
  }
  bool toAbi(o3ds_icp_params* p) const final {
    *p = o3ds_detail::icpParams(O3DS_ICP_GENERALIZED, maxCorrespondenceDistance_, icpConvergenceCriteria_);
    return true;
  }
  bool estimateNormalsOrCovariancesIfNeededDev(o3ds_handle h, o3ds_cloud c) const final {
    o3ds_detail::checkNormalEstimationParameters(maxRadiusNormalEstimation_, knnNormalEstimation_);
    o3ds_detail::Handle::check(o3ds_estimate_normals(h, c, maxRadiusNormalEstimation_, knnNormalEstimation_));
    return true;
  }
  double maxCorrespondenceDistance_ = 1.0;
  int knnNormalEstimation_ = 10;
  double maxRadiusNormalEstimation_ = 2.0;
  open3d::pipelines::registration::ICPConvergenceCriteria icpConvergenceCriteria_;
};

// RegistrationIcpPointToPoint (CloudRegistration.hpp:30-41, CloudRegistration.cpp:69-81)
class RegistrationIcpPointToPoint : public CloudRegistration {
 public:
  RegistrationResult registerClouds(const PointCloud& source, const PointCloud& target, const Transform& init) const final {
    const o3ds_icp_params p = o3ds_detail::icpParams(O3DS_ICP_POINT_TO_POINT, maxCorrespondenceDistance_, icpConvergenceCriteria_);
    o3ds_icp_result r{};
    o3ds_detail::Handle::check(o3ds_icp_point_to_point(o3ds_detail::Handle::get(), o3ds_detail::xyz(source.points_), source.points_.size(),
                                                       o3ds_detail::xyz(target.points_), target.points_.size(), o3ds_detail::pose_data(init),
                                                       &p, &r));
    return RegistrationIcpPointToPlane::toResult(r);
  }
  void estimateNormalsOrCovariancesIfNeeded(PointCloud*) const final {}  // nothing to prepare (CloudRegistration.hpp:26 default)
  bool toAbi(o3ds_icp_params* p) const final {
    *p = o3ds_detail::icpParams(O3DS_ICP_POINT_TO_POINT, maxCorrespondenceDistance_, icpConvergenceCriteria_);
    return true;
  }
  bool estimateNormalsOrCovariancesIfNeededDev(o3ds_handle, o3ds_cloud) const final { return true; }
  double maxCorrespondenceDistance_ = 1.0;
  open3d::pipelines::registration::ICPConvergenceCriteria icpConvergenceCriteria_;
};

inline std::unique_ptr<RegistrationIcpPointToPoint> createPointToPointIcp(const CloudRegistrationParameters& p) {  // CloudRegistration.cpp:76-81
  auto ret = std::make_unique<RegistrationIcpPointToPoint>();
  ret->maxCorrespondenceDistance_ = p.icp_.maxCorrespondenceDistance_;
  ret->icpConvergenceCriteria_.max_iteration_ = p.icp_.maxNumIter_;
  return ret;
}

inline std::unique_ptr<RegistrationIcpGeneralized> createGeneralizedIcp(const CloudRegistrationParameters& p) {  // CloudRegistration.cpp:32-39
  auto ret = std::make_unique<RegistrationIcpGeneralized>();
  ret->maxCorrespondenceDistance_ = p.icp_.maxCorrespondenceDistance_;
  ret->knnNormalEstimation_ = p.icp_.knn_;
  ret->maxRadiusNormalEstimation_ = p.icp_.maxDistanceKnn_;
  ret->icpConvergenceCriteria_.max_iteration_ = p.icp_.maxNumIter_;
  return ret;
}

inline std::unique_ptr<RegistrationIcpPointToPlane> createPointToPlaneIcp(const CloudRegistrationParameters& p) {  // CloudRegistration.cpp:58-65
  auto ret = std::make_unique<RegistrationIcpPointToPlane>();
  ret->maxCorrespondenceDistance_ = p.icp_.maxCorrespondenceDistance_;
  ret->knnNormalEstimation_ = p.icp_.knn_;
  ret->maxRadiusNormalEstimation_ = p.icp_.maxDistanceKnn_;
  ret->icpConvergenceCriteria_.max_iteration_ = p.icp_.maxNumIter_;
  return ret;
}

inline std::unique_ptr<CloudRegistration> cloudRegistrationFactory(const CloudRegistrationParameters& p) {  // CloudRegistration.cpp:85-100
  switch (p.regType_) {
    case CloudRegistrationType::PointToPlaneIcp:
      return createPointToPlaneIcp(p);
    case CloudRegistrationType::GeneralizedIcp:
      return createGeneralizedIcp(p);
    case CloudRegistrationType::PointToPointIcp:
      return createPointToPointIcp(p);
    default:
      throw std::runtime_error("cloud: unknown type of cloud registration");
  }
}

// ---- helpers (helpers.hpp:20-25) -----------------------------------------------------------------------------------
inline void voxelize(double voxelSize, PointCloud* pcl) {  // helpers.cpp:107-113
  if (voxelSize <= 0 || pcl->points_.empty()) return;
  o3ds_detail::DevCloud in(*pcl);
  o3ds_cloud o = 0;
  o3ds_detail::Handle::check(o3ds_voxel_down_sample(o3ds_detail::Handle::get(), in.id(), voxelSize, &o));
  o3ds_detail::DevCloud(o).download(pcl);
}
inline std::shared_ptr<PointCloud> voxelizeWithinCroppingVolume(double voxel_size, const CroppingVolume& croppingVolume,
                                                                const PointCloud& cloud) {  // helpers.cpp:115-183
  auto out = std::make_shared<PointCloud>();
  *out = cloud;
  if (voxel_size <= 0.0 || cloud.points_.empty()) return out;
  o3ds_detail::DevCloud m(cloud);
  const o3ds_crop c = croppingVolume.toAbi();
  o3ds_detail::Handle::check(o3ds_voxelize_within_volume(o3ds_detail::Handle::get(), m.id(), voxel_size, &c));
  m.download(out.get());
  return out;
}
inline std::shared_ptr<PointCloud> transform(const Transform& T, const PointCloud& cloud) {  // helpers.cpp:273-305 (Matrix4d in the reference)
  auto out = std::make_shared<PointCloud>();
  if (cloud.points_.empty()) return out;
  o3ds_detail::DevCloud in(cloud);
  o3ds_cloud o = 0;
  o3ds_detail::Handle::check(o3ds_transform_cloud(o3ds_detail::Handle::get(), in.id(), o3ds_detail::pose_data(T), &o));
  o3ds_detail::DevCloud(o).download(out.get());
  return out;
}

// ---- device-resident scan-to-map (ScanToMapRegistration.cpp:55-62 + Submap.cpp:39-75) ------------------------------
// The reference keeps mapCloud_ on the host, re-crops it and rebuilds a KD-tree for every scan; this keeps the submap and
// its index in HBM across scans, so scanToMapRegistration moves only the scan (<= 1.5 MB) over PCIe.  The map owns its backend
// handle and a mutex (the reference's mapPointCloudMutex_, Submap.cpp:69), so it may be built, filled and read from different
// threads.  o3d_slam::Submap (o3ds_mapping.hpp) wraps this in the reference's own class.
class DeviceSubmap {
 public:
  DeviceSubmap() = default;
  ~DeviceSubmap() {
    if (map_) o3ds_cloud_free(h_.get(), map_);
  }
  DeviceSubmap(const DeviceSubmap&) = delete;
  DeviceSubmap& operator=(const DeviceSubmap&) = delete;
  // Submap::carve for the sparse map (Submap.cpp:109-125): the caller applies the every-N-scans gate; returns #removed points
  size_t carve(const PointCloud& rawScan, const Transform& mapToRangeSensor, const CroppingVolume& mapBuilderCropper, double voxelSize,
               double maxRaytracingLength, double truncationDistance, double minDotProductWithNormal) {
    std::lock_guard<std::mutex> lck(mutex_);
    if (rawScan.IsEmpty() || sizeLocked() == 0) return 0;
    o3ds_detail::DevCloud s(rawScan, h_.get());
    const o3ds_crop c = mapBuilderCropper.toAbi();
    const o3ds_carving_params cp{voxelSize, maxRaytracingLength, truncationDistance, minDotProductWithNormal};
    size_t removed = 0;
    o3ds_detail::Handle::check(o3ds_map_carve(h_.get(), id(), s.id(), o3ds_detail::pose_data(mapToRangeSensor), &c, &cp, &removed));
    ++version_;
    return removed;
  }
  // Submap::insertScan core (Submap.cpp:54,70-72)
  bool insertScan(const PointCloud& preProcessedScan, const Transform& mapToRangeSensor, double mapVoxelSize,
                  CroppingVolume* mapBuilderCropper, double maxCorrespondenceDistance) {
    if (preProcessedScan.IsEmpty()) return true;
    std::lock_guard<std::mutex> lck(mutex_);
    o3ds_detail::DevCloud s(preProcessedScan, h_.get());
    mapBuilderCropper->setPose(mapToRangeSensor);
    const o3ds_crop c = mapBuilderCropper->toAbi();
    o3ds_detail::Handle::check(o3ds_map_insert_scan(h_.get(), id(), s.id(), o3ds_detail::pose_data(mapToRangeSensor), mapVoxelSize, &c,
                                                    maxCorrespondenceDistance));
    ++version_;
    return true;
  }
  // the isUseInitialMap_ branch of Submap::insertScan (Submap.cpp:47-52): map = voxelize(scan), no transform, no crop volume
  void setInitialMap(const PointCloud& preProcessedScan, double mapVoxelSize, double maxCorrespondenceDistance) {
    std::lock_guard<std::mutex> lck(mutex_);
    o3ds_detail::DevCloud s(preProcessedScan, h_.get());
    o3ds_cloud v = 0;
    o3ds_detail::Handle::check(o3ds_voxel_down_sample(h_.get(), s.id(), mapVoxelSize, &v));  // voxel <= 0 returns a copy (helpers.cpp:108-110)
    adopt(v, maxCorrespondenceDistance);
  }
  // mapCloud_.Transform(T) (Submap::transform, Submap.cpp:94-107); the NN index is rebuilt for the moved points
  void transform(const Transform& T, double maxCorrespondenceDistance) {
    std::lock_guard<std::mutex> lck(mutex_);
    if (sizeLocked() == 0) return;
    o3ds_cloud moved = 0;
    o3ds_detail::Handle::check(o3ds_transform_cloud(h_.get(), id(), o3ds_detail::pose_data(T), &moved));
    adopt(moved, maxCorrespondenceDistance);
  }
  // cloudRegistration->registerClouds(scan, crop(map), initialGuess) with the crop volume fused into the search
  // (ScanToMapIcp::scanToMapRegistration, ScanToMapRegistration.cpp:55-62); params.method selects the estimator
  RegistrationResult registerScan(const PointCloud& scan, const o3ds_crop& scanMatcherCrop, const Transform& initialGuess,
                                  const o3ds_icp_params& params) const {
    std::lock_guard<std::mutex> lck(mutex_);
    if (sizeLocked() == 0) throw std::runtime_error("map patch size is zero");  // assert_gt, ScanToMapRegistration.cpp:60
    o3ds_detail::DevCloud s(scan, h_.get());
    o3ds_icp_result r{};
    o3ds_detail::Handle::check(o3ds_icp_register_dev(h_.get(), s.id(), map_, &scanMatcherCrop, o3ds_detail::pose_data(initialGuess), &params, &r));
    return RegistrationIcpPointToPlane::toResult(r);
  }
  RegistrationResult scanToMapRegistration(const PointCloud& scan, CroppingVolume* scanMatcherCropper, const Transform& mapToRangeSensor,
                                           const Transform& initialGuess, const CloudRegistration& reg) const {
    o3ds_icp_params p{};
    if (!reg.toAbi(&p)) throw std::runtime_error("scanToMapRegistration: this CloudRegistration is not known to the backend");
    scanMatcherCropper->setPose(mapToRangeSensor);
    return registerScan(scan, scanMatcherCropper->toAbi(), initialGuess, p);
  }
  void getMapPointCloud(PointCloud* out) const {
    std::lock_guard<std::mutex> lck(mutex_);
    if (!map_) {
      *out = PointCloud();
      return;
    }
    o3ds_detail::DevCloud borrowed(map_, h_.get());
    try {
      borrowed.download(out);
    } catch (...) {
      borrowed.release();
      throw;
    }
    borrowed.release();
  }
  size_t size() const {
    std::lock_guard<std::mutex> lck(mutex_);
    return sizeLocked();
  }
  // bumped by every mutation: lets a caller keep a host mirror and refresh it only when the device map changed
  uint64_t version() const {
    std::lock_guard<std::mutex> lck(mutex_);
    return version_;
  }

  // saveToFile (output.cpp:39-47) of the device-resident map, rows narrowed on the device
  bool saveToFile(const std::string& filename) const;

 private:
  o3ds_cloud id() {  // the (initially empty) device cloud, made on first use so that constructing a map needs no device
    if (!map_) o3ds_detail::Handle::check(o3ds_cloud_upload(h_.get(), nullptr, nullptr, 0, &map_));
    return map_;
  }
  size_t sizeLocked() const {
    size_t n = 0;
    if (map_) o3ds_detail::Handle::check(o3ds_cloud_size(h_.get(), map_, &n, nullptr));
    return n;
  }
  void adopt(o3ds_cloud fresh, double maxCorrespondenceDistance) {
    if (map_) o3ds_cloud_free(h_.get(), map_);
    map_ = fresh;
    if (maxCorrespondenceDistance > 0.0) o3ds_detail::Handle::check(o3ds_cloud_build_index(h_.get(), map_, maxCorrespondenceDistance, 0.0));
    ++version_;
  }
  o3ds_detail::OwnedHandle h_;
  mutable std::mutex mutex_;
  o3ds_cloud map_ = 0;
  uint64_t version_ = 0;
};

// ---- egress: saveToFile (src/output.cpp:39-47) -----------------------------------------------------------------------
// The reference copies the cloud and calls [O3D] io::WritePointCloudToPCD with default options: binary, uncompressed PCD v0.7,
// float32 x y z (+ normal_x normal_y normal_z, + the packed rgb field, when present).  The device writes those
// rows itself (o3ds_cloud_download_f32), so a map that lives in HBM is saved without the fp64 host copy.  Returns false when the
// file cannot be written, as WritePointCloudToPCD does.
inline bool saveDeviceCloudToFile(const std::string& filename, o3ds_cloud cloud, o3ds_handle h = nullptr) {
  if (!h) h = o3ds_detail::Handle::get();
  const std::string name = filename.find(".pcd") == std::string::npos ? filename + ".pcd" : filename;
  size_t n = 0;
  int hn = 0, hc = 0;
  o3ds_detail::Handle::check(o3ds_cloud_size(h, cloud, &n, &hn));
  o3ds_detail::Handle::check(o3ds_cloud_has_colors(h, cloud, &hc));
  const size_t step = 12 + (hn ? 12 : 0) + (hc ? 4 : 0);
  std::vector<unsigned char> rows(n * step);
  o3ds_detail::Handle::check(o3ds_cloud_download_f32(h, cloud, rows.data(), n, step, 0, 4, 8, hn ? (size_t)12 : O3DS_NO_FIELD,
                                                     hc ? step - 4 : O3DS_NO_FIELD, 1));
  std::FILE* f = std::fopen(name.c_str(), "wb");
  if (!f) return false;
  std::string fields = "x y z", fours = "4 4 4", types = "F F F", ones = "1 1 1";
  for (int k = 0; k < (hn ? 3 : 0) + (hc ? 1 : 0); ++k) {
    static const char* const extra[] = {" normal_x", " normal_y", " normal_z"};
    fields += (k < (hn ? 3 : 0)) ? extra[k] : " rgb";
    fours += " 4", types += " F", ones += " 1";
  }
  bool ok = std::fprintf(f,
                         "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH %zu\nHEIGHT 1\n"
                         "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary\n",
                         fields.c_str(), fours.c_str(), types.c_str(), ones.c_str(), n, n) > 0;
  ok = ok && std::fwrite(rows.data(), 1, rows.size(), f) == rows.size();
  return std::fclose(f) == 0 && ok;
}
inline bool saveToFile(const std::string& filename, const PointCloud& cloud) {
  o3ds_detail::DevCloud d(cloud);
  return saveDeviceCloudToFile(filename, d.id(), d.handle());
}
inline bool DeviceSubmap::saveToFile(const std::string& filename) const {
  std::lock_guard<std::mutex> lck(mutex_);
  if (!map_) return o3d_slam::saveToFile(filename, PointCloud());
  return saveDeviceCloudToFile(filename, map_, h_.get());
}

}  // namespace o3d_slam
