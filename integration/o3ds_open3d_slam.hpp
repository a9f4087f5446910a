// o3ds_open3d_slam.hpp -- what integration/open3d_slam_o3ds.patch makes open3d_slam's own sources call.
//
// The patch keeps every class, header and signature of open3d_slam and replaces only the BODIES on the scan-matching / map-fusion
// path (CloudRegistration.cpp, ScanToMapRegistration.cpp, Submap.cpp) with calls into libo3ds_backend.so (include/o3ds_backend.h,
// a C ABI).  This header is the thin C++ between the two: it speaks open3d::geometry::PointCloud / Eigen::Isometry3d on one side
// and plain pointers on the other, and defines nothing in namespace o3d_slam (no clash with the reference's declarations).
//   Seam 1  CloudRegistration::registerClouds / estimateNormalsOrCovariancesIfNeeded  -> o3ds::registerClouds, o3ds::estimateNormals
//   Seam 2  ScanToMapIcp::preprocess / processForScanMatchingAndMerging,
//           LidarOdometry::preprocess                                                 -> o3ds::preprocessScan, o3ds::cropScan
//           ScanToMapIcp::scanToMapRegistration                                       -> o3ds::DeviceSubmap::registerScan
//   Seam 3  Submap::insertScan / carve / transform / copy                             -> o3ds::DeviceSubmap
// A scan is uploaded ONCE per worker: preprocessScan returns the reference's PointCloudPtr pointing at a ScanOnDevice -- an ordinary
// open3d PointCloud (host points and normals filled in, every caller may read it) that also remembers its copy in HBM.  The calls that
// follow in the reference's flow (registerClouds in LidarOdometry::addRangeScan, scanToMapRegistration and Submap::insertScan in
// Mapper::addRangeMeasurement) recognise it and use the device copy instead of uploading the host one again.
// Threads: a backend handle is one HIP stream plus scratch and is not re-entrant.  The stateless calls use one handle per calling
// thread; a DeviceSubmap owns a handle and a mutex (the reference guards mapCloud_ with mapPointCloudMutex_ in the same places).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include <open3d/geometry/PointCloud.h>
#include <open3d/pipelines/registration/Registration.h>
#include <Eigen/Dense>

#include "o3ds_backend.h"
#include "open3d_slam/Parameters.hpp"

namespace o3ds {

using PointCloud = open3d::geometry::PointCloud;
using RegistrationResult = open3d::pipelines::registration::RegistrationResult;
using ICPConvergenceCriteria = open3d::pipelines::registration::ICPConvergenceCriteria;

inline void check(o3ds_handle h, int rc) {
  if (rc != O3DS_OK) throw std::runtime_error(std::string("o3ds backend: ") + o3ds_last_error(h));
}

// one backend handle, destroyed with its owner
class OwnedHandle {
 public:
  OwnedHandle() = default;
  OwnedHandle(const OwnedHandle&) = delete;
  OwnedHandle& operator=(const OwnedHandle&) = delete;
  ~OwnedHandle() {
    if (h_) o3ds_destroy(h_);
  }
  o3ds_handle get() const {  // created on first use, so constructing the owner needs no device
    if (!h_) {
      int device = 0;
      if (const char* e = std::getenv("O3DS_DEVICE")) device = std::atoi(e);
      check(nullptr, o3ds_create(device, &h_));
    }
    return h_;
  }

 private:
  mutable o3ds_handle h_ = nullptr;
};
// a handle and the lock that serialises its users: the stateless calls take their calling thread's box, a DeviceSubmap owns one
struct HandleBox {
  OwnedHandle h;
  std::recursive_mutex m;  // recursive: the last reference to a device scan may be dropped inside a call that holds the lock
};
inline std::shared_ptr<HandleBox> threadBox() {  // one per calling thread (odometry, mapping, loop-closure workers), created on first use
  static thread_local std::shared_ptr<HandleBox> box = std::make_shared<HandleBox>();
  return box;
}

// CroppingVolume in the ABI's form.  croppingVolumeFactory (croppers.cpp:23-47) picks which parameters a named volume uses; only the
// translation of the pose enters the predicates (croppers.cpp:121-165).
inline o3ds_crop makeCrop(const o3d_slam::ScanCroppingParameters& p, const Eigen::Isometry3d& pose) {
  o3ds_crop c{};
  if (p.cropperName_ == "MaxRadius")
    c.kind = O3DS_CROP_MAX_RADIUS;
  else if (p.cropperName_ == "MinRadius")
    c.kind = O3DS_CROP_MIN_RADIUS;
  else if (p.cropperName_ == "MinMaxRadius")
    c.kind = O3DS_CROP_MIN_MAX_RADIUS;
  else if (p.cropperName_ == "Cylinder")
    c.kind = O3DS_CROP_CYLINDER;
  else
    throw std::runtime_error("Unknown cropper type");  // croppers.cpp:45
  c.invert = 0;
  for (int a = 0; a < 3; ++a) c.center[a] = pose.translation()(a);
  c.rmin = p.croppingMinRadius_;
  c.rmax = p.croppingMaxRadius_;
  c.zmin = p.croppingMinZ_;
  c.zmax = p.croppingMaxZ_;
  return c;
}

// every point `inner` keeps is kept by `outer` too (same kind, same centre, neither inverted, radii nested): cropping inner's output with
// outer returns it unchanged -- crop(crop(x, V1), V2) = crop(x, V1) for V1 inside V2
inline bool cropContains(const o3ds_crop& outer, const o3ds_crop& inner) {
  if (outer.kind == O3DS_CROP_NONE && !outer.invert) return true;
  if (outer.kind != inner.kind || outer.invert || inner.invert) return false;
  for (int a = 0; a < 3; ++a)
    if (outer.center[a] != inner.center[a]) return false;
  switch (outer.kind) {
    case O3DS_CROP_MAX_RADIUS:
      return outer.rmax >= inner.rmax;
    case O3DS_CROP_MIN_RADIUS:
      return outer.rmin <= inner.rmin;
    case O3DS_CROP_MIN_MAX_RADIUS:
      return outer.rmin <= inner.rmin && outer.rmax >= inner.rmax;
    case O3DS_CROP_CYLINDER:
      return outer.rmax >= inner.rmax && outer.zmin <= inner.zmin && outer.zmax >= inner.zmax;
    default:
      return false;
  }
}

inline o3ds_crop noCrop() {  // the base CroppingVolume (croppers.cpp:49-51): everything is inside
  o3ds_crop c{};
  c.kind = O3DS_CROP_NONE;
  return c;
}

// A device cloud whose last reference drops on a thread that does not own its handle is not freed there and then if the owner is inside a
// call (its lock is taken): waiting for a second handle while holding one is how two workers deadlock.  It goes to a short list that
// whoever next holds that handle's lock empties (drainDeferredFrees, called where the seams take the lock).
struct DeferredFrees {
  std::mutex m;
  std::vector<std::pair<std::shared_ptr<HandleBox>, o3ds_cloud>> dead;
};
inline DeferredFrees& deferredFrees() {
  static DeferredFrees d;
  return d;
}
inline void freeOrDefer(const std::shared_ptr<HandleBox>& box, o3ds_cloud id) {
  std::unique_lock<std::recursive_mutex> lck(box->m, std::try_to_lock);
  if (lck.owns_lock()) {
    o3ds_cloud_free(box->h.get(), id);
    return;
  }
  DeferredFrees& d = deferredFrees();
  std::lock_guard<std::mutex> dl(d.m);
  d.dead.emplace_back(box, id);
}
inline void drainDeferredFrees(const std::shared_ptr<HandleBox>& box) {  // box->m held by the caller
  DeferredFrees& d = deferredFrees();
  std::vector<o3ds_cloud> mine;
  {
    std::lock_guard<std::mutex> dl(d.m);
    if (d.dead.empty()) return;
    for (size_t i = 0; i < d.dead.size();)
      if (d.dead[i].first.get() == box.get()) {
        mine.push_back(d.dead[i].second);
        d.dead[i] = d.dead.back();
        d.dead.pop_back();
      } else {
        ++i;
      }
  }
  for (o3ds_cloud id : mine) o3ds_cloud_free(box->h.get(), id);
}

// ---- a pre-processed scan that is still on the device ------------------------------------------------------------------------------
// The device copy belongs to the handle (box) of the thread that made it and is freed with the last PointCloud that refers to it.
struct DeviceRef {
  std::shared_ptr<HandleBox> box;
  o3ds_cloud id = 0;
  size_t n = 0;
  uint64_t stamp = 0;  // fingerprint of the host arrays when the copy was made (see fingerprint)
  // What ANOTHER worker's handle needs to copy the cloud without this one's lock (o3ds_cloud_export_view): open3d_slam's odometry and
  // mapping workers run on two threads (SlamWrapper.cpp:227-236), and the mapper asks for the scan while the odometry worker is inside
  // its registration -- holding its handle for hundreds of microseconds.  Exported by the owner right after it made the cloud.
  o3ds_cloud_view view{};
  DeviceRef() = default;
  DeviceRef(const DeviceRef&) = delete;
  DeviceRef& operator=(const DeviceRef&) = delete;
  void exportView() {  // box->m held by the caller
    if (id && box && !view.event) (void)o3ds_cloud_export_view(box->h.get(), id, &view);  // (a failure leaves the view empty: readers fall back)
  }
  ~DeviceRef() {
    o3ds_cloud_view_release(&view);
    if (id && box) freeOrDefer(box, id);
  }
};
// the cloud behind `ref`, which lives on another worker's handle, as a cloud of handle h: from the exported view if there is one (no
// lock of the other handle needed), else device to device if the other handle happens to be free; 0 if neither (the caller uploads)
inline o3ds_cloud copyFromOtherHandle(o3ds_handle h, const DeviceRef& ref) {
  o3ds_cloud mine = 0;
  if (ref.view.event && ref.view.n == ref.n && o3ds_cloud_import_view(h, &ref.view, &mine) == O3DS_OK) return mine;
  std::unique_lock<std::recursive_mutex> other(ref.box->m, std::try_to_lock);  // never wait for a second handle while holding one
  if (other.owns_lock() && o3ds_cloud_copy_across(h, ref.box->h.get(), ref.id, &mine) == O3DS_OK) return mine;
  return 0;
}
// what the seams hand out instead of a plain PointCloud: the same object for every reader, plus the device copy
class ScanOnDevice : public PointCloud {
 public:
  std::shared_ptr<const DeviceRef> device_;
  std::shared_ptr<const DeviceRef> before_draw_;  // the cloud before its RandomDownSample, kept for the other worker (preprocessScan's memo)
};
// Guards against a caller that edits the host arrays after the device copy was made (the copy would be stale): sizes, presence of
// normals / colours and the bits of up to 64 evenly spaced points, normals and colours.  Not a proof of equality: open3d_slam's own flow
// never edits a pre-processed scan between the seams; code that does calls invalidateDeviceCopy() after the edit, and O3DS_FINGERPRINT_FULL=1
// hashes every element (a debugging aid: one pass over the host arrays per seam).
inline uint64_t fingerprint(const PointCloud& c) {
  static const bool full = std::getenv("O3DS_FINGERPRINT_FULL") && std::atoi(std::getenv("O3DS_FINGERPRINT_FULL")) != 0;
  uint64_t f = 0x9e3779b97f4a7c15ull ^ (uint64_t)c.points_.size() ^ ((uint64_t)c.normals_.size() << 32) ^ ((uint64_t)c.colors_.size() << 17);
  const size_t n = c.points_.size();
  if (n == 0) return f;
  const size_t step = (!full && n > 64) ? n / 64 : 1;
  for (size_t i = 0; i < n; i += step) {
    uint64_t w[3];
    std::memcpy(w, c.points_[i].data(), sizeof(w));
    f = (f << 7 | f >> 57) ^ w[0] ^ (w[1] << 1) ^ (w[2] << 2);
    if (c.normals_.size() == n) {
      std::memcpy(w, c.normals_[i].data(), sizeof(w));
      f = (f << 11 | f >> 53) ^ w[0] ^ (w[1] << 3) ^ (w[2] << 5);
    }
    if (c.colors_.size() == n) {
      std::memcpy(w, c.colors_[i].data(), sizeof(w));
      f = (f << 13 | f >> 51) ^ w[0] ^ (w[1] << 7) ^ (w[2] << 9);
    }
  }
  return f;
}
// for code that edits a cloud the seams handed out: the next seam uploads the host arrays again
inline void invalidateDeviceCopy(PointCloud& c) {
  if (ScanOnDevice* s = dynamic_cast<ScanOnDevice*>(&c)) s->device_.reset();
}
// PointCloud::colors_ (typedefs.hpp:24) cross the seam with the points: the reference's crop / VoxelDownSample / RandomDownSample keep
// them on the pre-processed scan that reaches mapCloud_, and so do the device operations (o3ds_backend.h: o3ds_cloud_set_colors)
inline void uploadColors(o3ds_handle h, o3ds_cloud id, const PointCloud& c) {
  if (!c.points_.empty() && c.colors_.size() == c.points_.size()) check(h, o3ds_cloud_set_colors(h, id, reinterpret_cast<const double*>(c.colors_.data())));
}
inline std::shared_ptr<const DeviceRef> deviceCopyOf(const PointCloud& c) {
  const ScanOnDevice* s = dynamic_cast<const ScanOnDevice*>(&c);
  if (!s || !s->device_ || s->device_->n != c.points_.size() || c.points_.empty()) return nullptr;
  if (std::getenv("O3DS_NO_DEVICE_SCANS")) return nullptr;  // A/B switch: always upload, as round 2 did
  return s->device_->stamp == fingerprint(c) ? s->device_ : nullptr;
}

// a PointCloud on the device for the duration of a call
class DeviceCloud {
 public:
  DeviceCloud(o3ds_handle h, const PointCloud& c) : h_(h) {
    const double* nrm = c.HasNormals() ? reinterpret_cast<const double*>(c.normals_.data()) : nullptr;
    check(h_, o3ds_cloud_upload(h_, reinterpret_cast<const double*>(c.points_.data()), nrm, c.points_.size(), &id_));
    colorsOrFree(c);
  }
  // The cloud on handle h, whose lock the caller holds: borrowed if its device copy already lives on this handle (`mine` is the
  // box h belongs to, null for a handle no ScanOnDevice can belong to), copied device to device if it lives on another handle,
  // uploaded from the host arrays otherwise.
  DeviceCloud(o3ds_handle h, const HandleBox* mine, const PointCloud& c) : h_(h) {
    if (std::shared_ptr<const DeviceRef> ref = deviceCopyOf(c)) {
      if (ref->box.get() == mine) {
        id_ = ref->id;
        borrowed_ = ref;  // keeps the copy alive; nothing to free here
        return;
      }
      id_ = copyFromOtherHandle(h_, *ref);
      if (id_) {
        fromDevice_ = true;
        return;
      }
    }
    const double* nrm = c.HasNormals() ? reinterpret_cast<const double*>(c.normals_.data()) : nullptr;
    check(h_, o3ds_cloud_upload(h_, reinterpret_cast<const double*>(c.points_.data()), nrm, c.points_.size(), &id_));
    colorsOrFree(c);
  }
  DeviceCloud(const DeviceCloud&) = delete;
  DeviceCloud& operator=(const DeviceCloud&) = delete;
  ~DeviceCloud() {
    if (id_ && !borrowed_) o3ds_cloud_free(h_, id_);
  }
  o3ds_cloud id() const { return id_; }
  bool uploaded() const { return !borrowed_ && !fromDevice_; }

 private:
  void colorsOrFree(const PointCloud& c) {  // in a constructor: what was uploaded does not outlive a failure
    try {
      uploadColors(h_, id_, c);
    } catch (...) {
      o3ds_cloud_free(h_, id_);
      id_ = 0;
      throw;
    }
  }
  o3ds_handle h_;
  o3ds_cloud id_ = 0;
  std::shared_ptr<const DeviceRef> borrowed_;
  bool fromDevice_ = false;
};

inline void downloadCloud(o3ds_handle h, o3ds_cloud id, PointCloud* out) {
  size_t n = 0;
  int hasNormals = 0;
  check(h, o3ds_cloud_size(h, id, &n, &hasNormals));
  out->points_.resize(n);
  out->normals_.resize(hasNormals ? n : 0);
  out->colors_.clear();
  out->covariances_.clear();
  if (n == 0) return;
  check(h, o3ds_cloud_download(h, id, reinterpret_cast<double*>(out->points_.data()),
                               hasNormals ? reinterpret_cast<double*>(out->normals_.data()) : nullptr, n));
  int hasColors = 0;
  check(h, o3ds_cloud_has_colors(h, id, &hasColors));
  if (hasColors) {
    out->colors_.resize(n);
    check(h, o3ds_cloud_get_colors(h, id, reinterpret_cast<double*>(out->colors_.data()), n));
  }
}

inline RegistrationResult toResult(const o3ds_icp_result& r) {
  RegistrationResult out(Eigen::Map<const Eigen::Matrix4d>(r.transformation));
  out.fitness_ = r.fitness;
  out.inlier_rmse_ = r.inlier_rmse;
  return out;  // correspondence_set_ is never read for ICP results in open3d_slam and is not materialised
}
inline o3ds_icp_params icpParams(int method, double maxCorrespondenceDistance, const ICPConvergenceCriteria& c) {
  o3ds_icp_params p{};
  p.max_correspondence_distance = maxCorrespondenceDistance;
  p.max_iteration = c.max_iteration_;
  p.method = method;
  p.relative_fitness = c.relative_fitness_;
  p.relative_rmse = c.relative_rmse_;
  return p;
}

// Seam 1: [O3D] RegistrationICP / RegistrationGeneralizedICP (CloudRegistration.cpp:16-21,44-48,69-73).  A cloud that is still on the
// device (ScanOnDevice) is used where it is -- LidarOdometry::addRangeScan then registers two resident clouds, the target with the
// search grid its normal estimation left behind; a host cloud is uploaded and its index built per call, as the reference builds its
// KD-tree per call.
// Work for the time the calling thread would wait for its NEXT stateless registration (o3ds_icp_overlap_next, include/o3ds_backend.h): `fn`
// runs once, on this thread, after that registration's launches are queued -- a driver that feeds LidarOdometry::addRangeScan from one thread
// queues the pre-processing of the following scan there.  It may call the seams of this header (the handle's lock is
// recursive); it must not touch the two clouds being registered.
inline std::function<void()>& overlapHeld() {
  static thread_local std::function<void()> held;
  return held;
}
inline void overlapNext(std::function<void()> fn) {
  const std::shared_ptr<HandleBox> box = threadBox();
  std::lock_guard<std::recursive_mutex> lck(box->m);
  overlapHeld() = std::move(fn);
  o3ds_overlap_fn cb = nullptr;
  if (overlapHeld())
    cb = [](void*) {
      std::function<void()> f = std::move(overlapHeld());
      overlapHeld() = nullptr;
      if (f) f();
    };
  check(box->h.get(), o3ds_icp_overlap_next(box->h.get(), cb, nullptr));
}

inline RegistrationResult registerClouds(int method /* o3ds_icp_method */, const PointCloud& source, const PointCloud& target,
                                         const Eigen::Matrix4d& init, double maxCorrespondenceDistance, const ICPConvergenceCriteria& criteria) {
  const std::shared_ptr<HandleBox> box = threadBox();
  std::lock_guard<std::recursive_mutex> lck(box->m);
  const o3ds_handle h = box->h.get();
  DeviceCloud s(h, box.get(), source), t(h, box.get(), target);
  if (t.uploaded()) check(h, o3ds_cloud_build_index(h, t.id(), maxCorrespondenceDistance, 0.0));  // otherwise o3ds_icp_register_dev keeps or builds it
  const o3ds_icp_params p = icpParams(method, maxCorrespondenceDistance, criteria);
  o3ds_icp_result r{};
  check(h, o3ds_icp_register_dev(h, s.id(), t.id(), nullptr, init.data(), &p, &r));
  return toResult(r);
}

// Seam 1b: EstimateNormals(Hybrid) + NormalizeNormals + OrientNormalsTowardsCameraLocation (CloudRegistration.cpp:22-30,49-56)
inline void estimateNormals(PointCloud* cloud, double maxRadius, int knn) {
  if (cloud->points_.empty()) return;
  const std::shared_ptr<HandleBox> box = threadBox();
  std::lock_guard<std::recursive_mutex> lck(box->m);
  const o3ds_handle h = box->h.get();
  DeviceCloud c(h, *cloud);
  check(h, o3ds_estimate_normals(h, c.id(), maxRadius, knn));
  cloud->normals_.resize(cloud->points_.size());
  check(h, o3ds_cloud_download(h, c.id(), nullptr, reinterpret_cast<double*>(cloud->normals_.data()), cloud->points_.size()));
}

// ---- Seam 2, first half: the scan chain ---------------------------------------------------------------------------------------------
// [O3D] RandomDownSample draws from std::mt19937 seeded by std::random_device, per call; setRandomDownSampleSeed pins the generator of
// the calling thread (tests, reproducible runs).
inline std::mt19937& downSampleGenerator() {
  static thread_local std::mt19937 g{std::random_device{}()};
  return g;
}
inline void setRandomDownSampleSeed(uint32_t seed) { downSampleGenerator().seed(seed); }

struct ScanChain {
  o3ds_crop crop;                // cropper of the chain in the sensor frame (noCrop() for the base CroppingVolume)
  double voxelSize = 0.0;        // o3d_slam::voxelize: <= 0 skips (helpers.cpp:108-110)
  bool estimateNormals = true;   // the registration type's estimateNormalsOrCovariancesIfNeeded (no-op for point-to-point)
  double normalRadius = 0.0;
  int normalKnn = 0;
  double downSamplingRatio = 1.0;
};
inline std::shared_ptr<PointCloud> adopt(const std::shared_ptr<HandleBox>& box, o3ds_cloud id) {  // box->m held; takes over `id`
  auto ref = std::make_shared<DeviceRef>();
  auto out = std::make_shared<ScanOnDevice>();
  downloadCloud(box->h.get(), id, out.get());  // may throw: `id` is still the caller's then
  ref->box = box;
  ref->id = id;
  ref->n = out->points_.size();
  ref->stamp = fingerprint(*out);
  ref->exportView();
  out->device_ = ref;
  return out;
}
// ScanToMapIcp::preprocess (ScanToMapRegistration.cpp:35-40) and LidarOdometry::preprocess (Odometry.cpp:25-30):
//   cropper->crop(in); voxelize(voxelSize, cropped); estimateNormalsOrCovariancesIfNeeded(cropped); cropped->RandomDownSample(ratio)
// Both run this chain on the same raw scan, and the shipped configuration gives them the same parameters
// (parameter_structure_definitions.lua: both `scan_processing` blocks and the map builder's `scan_cropping` are copies of one table), so the
// second caller gets what the first computed UP TO the RandomDownSample -- the crop, the voxel grid and the normal estimation are the
// expensive steps; each caller draws its own index list afterwards, as in the reference (at the shipped ratio of 0.3 as well as at 1).
// WHICH scan a call is about is said by the reference itself: both seams receive the scan's Time stamp (Odometry.hpp:27 addRangeScan(cloud,
// timestamp), Mapper.hpp:47 addRangeMeasurement(cloud, timestamp)); the patched call sites pass it on (setScanStamp) and the memo is keyed
// on it, the raw scan's size and every parameter of the chain.  Without a stamp (a caller that does not say) nothing is shared.
// The memo holds ONE entry weakly: it lives as long as a caller still holds the cloud (the odometry does until the next scan).
// O3DS_SHARE_PREPROCESS=0 computes everything twice, as the reference does.
inline int64_t& scanStamp() {  // of the calling thread: set by the patched addRangeScan / addRangeMeasurement, 0 = unknown
  static thread_local int64_t stamp = 0;
  return stamp;
}
inline void setScanStamp(int64_t ticks) { scanStamp() = ticks; }
struct ScanStampScope {  // for the duration of a seam call
  int64_t saved;
  explicit ScanStampScope(int64_t ticks) : saved(scanStamp()) { scanStamp() = ticks; }
  ~ScanStampScope() { scanStamp() = saved; }
};
struct PreprocessMemoEntry {
  int64_t stamp = 0;
  size_t n = 0;
  uint64_t raw_print = 0;  // fingerprint of the raw scan (64 sampled points): two workers' motion compensations give clouds of one stamp and size
  ScanChain chain{};
  std::shared_ptr<PointCloud> result;            // ratio >= 1: the finished cloud (host arrays + device copy), the same object for both callers
  std::shared_ptr<const DeviceRef> before_draw;  // ratio < 1: the cloud before RandomDownSample, on the device only
};
// The last few scans, HELD (a few megabytes each, host and device).  More than one, and not weakly: open3d_slam's two workers are some
// scans apart (SlamWrapper.cpp:258-347: the mapper takes scan k from its buffer while the odometry worker is on k + 1, k + 2 ...) -- with
// a single weak entry the mapper found the NEXT scan's entry in place of its own, or its own already released by the odometry, and
// computed everything again (0.5 ms per scan of the two-thread runs of round 5).
struct PreprocessMemo {
  std::mutex m;
  static constexpr int kEntries = 8;
  PreprocessMemoEntry e[kEntries];
  int next = 0;
};
inline PreprocessMemo& preprocessMemo() {
  static PreprocessMemo memo;
  return memo;
}
inline bool sameChain(const ScanChain& a, const ScanChain& b) {  // everything in front of the RandomDownSample
  return std::memcmp(&a.crop, &b.crop, sizeof(o3ds_crop)) == 0 && a.voxelSize == b.voxelSize && a.estimateNormals == b.estimateNormals &&
         a.normalRadius == b.normalRadius && a.normalKnn == b.normalKnn;
}
inline bool sharePreprocess() {
  static const bool on = !(std::getenv("O3DS_SHARE_PREPROCESS") && std::atoi(std::getenv("O3DS_SHARE_PREPROCESS")) == 0);
  return on;
}

// [O3D] RandomDownSample(ratio) of a device cloud of n points: shuffled indices, the first int(ratio * n) kept (SelectByIndex).
// [O3D] SelectByIndex marks the listed indices in a mask and walks the cloud once, so the kept points come out in CLOUD order (v0.15.1
// PointCloud.cpp; restated, unpinned); O3DS_SELECT_SHUFFLED=1 keeps them in the order of the shuffled list instead (SURVEY A.7's reading).
inline bool selectByIndexKeepsCloudOrder() {
  static const bool on = !(std::getenv("O3DS_SELECT_SHUFFLED") && std::atoi(std::getenv("O3DS_SELECT_SHUFFLED")) != 0);
  return on;
}
inline o3ds_cloud drawOnDevice(o3ds_handle h, o3ds_cloud cloud, size_t n, double ratio) {
  if (selectByIndexKeepsCloudOrder()) {
    // the draw itself on the device (o3ds_random_down_sample): one 64-bit seed from the generator names the subset; no shuffle of n
    // indices on the host, no index list to upload.  [O3D] seeds an mt19937 from std::random_device per call -- any uniformly drawn
    // k-subset is its outcome, none is reproducible; setRandomDownSampleSeed makes this one so
    std::mt19937& gen = downSampleGenerator();
    const uint64_t hi = gen(), lo = gen();
    o3ds_cloud kept = 0;
    check(h, o3ds_random_down_sample(h, cloud, ratio, (hi << 32) | lo, &kept));
    return kept;
  }
  std::vector<uint32_t> idx(n);
  std::iota(idx.begin(), idx.end(), 0u);
  std::shuffle(idx.begin(), idx.end(), downSampleGenerator());
  idx.resize((size_t)(int)(ratio * (double)n));
  if (selectByIndexKeepsCloudOrder()) std::sort(idx.begin(), idx.end());
  o3ds_cloud kept = 0;
  check(h, o3ds_select_by_index(h, cloud, idx.data(), idx.size(), &kept));
  return kept;
}

// One upload of the raw scan, the chain on the device, one download; the result stays on the device behind the returned cloud.
// With ratio >= 1 every index is kept and [O3D] SelectByIndex walks the cloud in its own order: the cloud itself (no shuffle needed).
inline std::shared_ptr<PointCloud> preprocessScan(const PointCloud& raw, const ScanChain& p) {
  if (p.estimateNormals) {
    if (!(p.normalRadius > 0.0)) throw std::runtime_error("maxRadiusNormalEstimation_");  // assert_gt, CloudRegistration.cpp:50-51
    if (!(p.normalKnn > 0)) throw std::runtime_error("knnNormalEstimation_");
  }
  if (p.downSamplingRatio < 0.0 || p.downSamplingRatio > 1.0)
    throw std::runtime_error("[RandomDownSample] Illegal sampling_ratio, sampling_ratio must be between 0 and 1.");  // [O3D]
  const std::shared_ptr<HandleBox> box = threadBox();
  std::lock_guard<std::recursive_mutex> lck(box->m);
  drainDeferredFrees(box);
  const o3ds_handle h = box->h.get();
  if (raw.points_.empty()) return std::make_shared<PointCloud>();
  const int64_t stamp = scanStamp();
  const bool share = sharePreprocess() && stamp != 0;
  const bool draw = p.downSamplingRatio < 1.0;
  if (share) {  // what the other worker made of this very scan
    PreprocessMemo& memo = preprocessMemo();
    std::shared_ptr<PointCloud> whole;
    std::shared_ptr<const DeviceRef> before;
    double memo_ratio = 0.0;
    {
      // (the reference runs separate motionCompensationOdom_ / motionCompensationMap_ objects, SlamWrapper.cpp:39-40,266,301: with a
      // non-trivial compensation the two workers hold DIFFERENT clouds of the same stamp and size -- the content is part of the key)
      std::lock_guard<std::mutex> ml(memo.m);
      const uint64_t print = fingerprint(raw);
      for (const PreprocessMemoEntry& e : memo.e)
        if (e.stamp == stamp && e.n == raw.points_.size() && sameChain(e.chain, p) && e.raw_print == print) {
          whole = e.result;
          before = e.before_draw;
          memo_ratio = e.chain.downSamplingRatio;
          if (whole || before) break;
        }
    }
    if (!draw && whole && memo_ratio >= 1.0) return whole;
    if (draw && before && before->n > 0) {  // its cloud before the draw: here (or copied here device to device), then this caller's own draw
      o3ds_cloud mine = 0, use = before->id;
      bool ok = true;
      if (before->box.get() != box.get()) {
        mine = copyFromOtherHandle(h, *before);
        ok = mine != 0;
        use = mine;
      }
      if (ok) {
        o3ds_cloud kept = 0;
        try {
          kept = drawOnDevice(h, use, before->n, p.downSamplingRatio);
          if (mine) o3ds_cloud_free(h, mine);
          mine = 0;
          return adopt(box, kept);
        } catch (...) {
          if (mine) o3ds_cloud_free(h, mine);
          if (kept) o3ds_cloud_free(h, kept);
          throw;
        }
      }
    }
  }
  DeviceCloud in(h, box.get(), raw);
  o3ds_cloud cur = 0;
  check(h, o3ds_crop_voxel_down_sample(h, in.id(), &p.crop, p.voxelSize, &cur));
  std::shared_ptr<DeviceRef> before;  // ratio < 1: keeps the cloud before the draw alive for the other worker
  try {
    size_t n = 0;
    check(h, o3ds_cloud_size(h, cur, &n, nullptr));
    if (p.estimateNormals && n > 0) check(h, o3ds_estimate_normals(h, cur, p.normalRadius, p.normalKnn));
    std::shared_ptr<PointCloud> out;
    if (draw && n > 0) {
      const o3ds_cloud kept = drawOnDevice(h, cur, n, p.downSamplingRatio);
      if (share) {
        before = std::make_shared<DeviceRef>();
        before->box = box;
        before->id = cur;
        before->n = n;
        before->exportView();
        cur = 0;  // (owned by `before` now)
      } else {
        o3ds_cloud_free(h, cur);
        cur = 0;
      }
      try {
        out = adopt(box, kept);
      } catch (...) {
        o3ds_cloud_free(h, kept);
        throw;
      }
    } else {
      out = adopt(box, cur);
      cur = 0;  // (adopted)
    }
    if (share) {
      PreprocessMemo& memo = preprocessMemo();
      std::lock_guard<std::mutex> ml(memo.m);
      PreprocessMemoEntry& e = memo.e[memo.next];
      memo.next = (memo.next + 1) % PreprocessMemo::kEntries;
      e.stamp = stamp;
      e.n = raw.points_.size();
      e.raw_print = fingerprint(raw);
      e.chain = p;
      e.result = out;
      e.before_draw = before;
      if (before) {  // the memo's entry lives as long as the caller's cloud does: the cloud carries the reference
        if (ScanOnDevice* sd = dynamic_cast<ScanOnDevice*>(out.get())) sd->before_draw_ = before;
      }
    }
    return out;
  } catch (...) {
    if (cur) o3ds_cloud_free(h, cur);
    throw;
  }
}
// CroppingVolume::crop (croppers.cpp:76-106) of a pre-processed scan: the narrow crop of processForScanMatchingAndMerging
// (ScanToMapRegistration.cpp:47-48), on the device when the scan is still there
inline std::shared_ptr<PointCloud> cropScan(const PointCloud& cloud, const o3ds_crop& crop) {
  const std::shared_ptr<HandleBox> box = threadBox();
  std::lock_guard<std::recursive_mutex> lck(box->m);
  const o3ds_handle h = box->h.get();
  if (cloud.points_.empty()) return std::make_shared<PointCloud>();
  DeviceCloud in(h, box.get(), cloud);
  o3ds_cloud out = 0;
  check(h, o3ds_crop_cloud(h, in.id(), &crop, &out));
  try {
    return adopt(box, out);
  } catch (...) {
    o3ds_cloud_free(h, out);
    throw;
  }
}

// Seam 3: Submap's mapCloud_, resident in HBM together with its search index.  Value semantics, because open3d_slam keeps its
// submaps in a std::vector<Submap> and copies them (SubmapCollection.hpp:79, SubmapCollection.cpp:139): copies share the device map
// until one of them changes it (copy on write), so a vector that grows does not move maps through host memory.
class DeviceSubmap {
 public:
  DeviceSubmap() : s_(std::make_shared<State>()) {}
  DeviceSubmap(const DeviceSubmap& o) : s_(o.shared()) {}
  DeviceSubmap& operator=(const DeviceSubmap& o) {
    if (this != &o) {
      std::shared_ptr<State> keep = o.shared();
      std::lock_guard<std::mutex> lck(ptrMutex_);
      s_ = keep;
    }
    return *this;
  }

  bool empty() const { return size() == 0; }
  size_t size() const {
    auto s = shared();
    std::lock_guard<std::mutex> lck(s->m);
    return s->size();
  }
  // changes with every mutation: lets Submap keep mapCloud_ as a host mirror and refresh it only when the device map changed
  uint64_t version() const {
    auto s = shared();
    std::lock_guard<std::mutex> lck(s->m);
    return s->version;
  }

  // the isUseInitialMap_ branch of Submap::insertScan (Submap.cpp:47-52): map = voxelize(scan), no transform, no crop volume
  void setInitialMap(const PointCloud& preProcessedScan, double mapVoxelSize, double maxCorrespondenceDistance) {
    auto s = unique();
    std::lock_guard<std::mutex> lck(s->m);
    DeviceCloud in(s->h.get(), nullptr, preProcessedScan);
    o3ds_cloud v = 0;
    check(s->h.get(), o3ds_voxel_down_sample(s->h.get(), in.id(), mapVoxelSize, &v));  // voxel <= 0: a copy (helpers.cpp:108-110)
    s->adopt(v, maxCorrespondenceDistance);
  }
  // Submap::insertScan's tail (Submap.cpp:54,69-72): transform, mapCloud_ +=, voxelizeWithinCroppingVolume, search index
  void insertScan(const PointCloud& preProcessedScan, const Eigen::Isometry3d& mapToRangeSensor, double mapVoxelSize,
                  const o3ds_crop& mapBuilderCrop, double maxCorrespondenceDistance) {
    if (preProcessedScan.IsEmpty()) return;
    auto s = unique();
    std::lock_guard<std::mutex> lck(s->m);
    DeviceCloud in(s->h.get(), nullptr, preProcessedScan);
    check(s->h.get(), o3ds_map_insert_scan(s->h.get(), s->id(), in.id(), mapToRangeSensor.matrix().data(), mapVoxelSize, &mapBuilderCrop,
                                           maxCorrespondenceDistance));
    ++s->version;
  }
  // Submap::carve for the sparse map (Submap.cpp:109-125 -> getIdxsOfCarvedPoints); the caller applies the every-N-scans gate.  toRemove /
  // scanRef receive what the reference keeps in its members of those names for the visualisation (Submap.cpp:119-120): the carved points
  // and the placed scan
  size_t carve(const PointCloud& rawScan, const Eigen::Isometry3d& mapToRangeSensor, const o3ds_crop& mapBuilderCrop,
               const o3d_slam::SpaceCarvingParameters& p, PointCloud* toRemove = nullptr, PointCloud* scanRef = nullptr) {
    auto s = unique();
    std::lock_guard<std::mutex> lck(s->m);
    if (s->size() == 0) return 0;  // Submap.cpp:111-113: an empty map returns before toRemove_ / scanRef_ are touched
    if (rawScan.IsEmpty()) {       // no rays: nothing is carved, and the reference's members become the (empty) results of this call
      if (toRemove) *toRemove = PointCloud();
      if (scanRef) *scanRef = PointCloud();
      return 0;
    }
    const o3ds_handle h = s->h.get();
    struct Owned {  // a device cloud this call made: freed on every way out
      o3ds_handle h;
      o3ds_cloud id = 0;
      ~Owned() {
        if (id) o3ds_cloud_free(h, id);
      }
    };
    DeviceCloud in(h, rawScan);
    const o3ds_carving_params cp{p.voxelSize_, p.maxRaytracingLength_, p.truncationDistance_, p.minDotProductWithNormal_};
    size_t removed = 0;
    Owned gone{h}, placed{h};
    check(h, o3ds_map_carve_removed(h, s->id(), in.id(), mapToRangeSensor.matrix().data(), &mapBuilderCrop, &cp, &removed, toRemove ? &gone.id : nullptr));
    if (removed > 0) ++s->version;  // an unchanged map keeps its host mirror
    if (toRemove) downloadCloud(h, gone.id, toRemove);
    if (scanRef) {
      check(h, o3ds_transform_cloud(h, in.id(), mapToRangeSensor.matrix().data(), &placed.id));
      downloadCloud(h, placed.id, scanRef);
    }
    return removed;
  }
  // mapCloud_.Transform(T) (Submap::transform, Submap.cpp:94-107); the index is rebuilt for the moved points
  void transform(const Eigen::Isometry3d& T, double maxCorrespondenceDistance) {
    auto s = unique();
    std::lock_guard<std::mutex> lck(s->m);
    if (s->size() == 0) return;
    o3ds_cloud moved = 0;
    check(s->h.get(), o3ds_transform_cloud(s->h.get(), s->id(), T.matrix().data(), &moved));
    s->adopt(moved, maxCorrespondenceDistance);
  }
  // Seam 2: cloudRegistration->registerClouds(scan, scanMatcherCropper_->crop(map), initialGuess) (ScanToMapRegistration.cpp:55-62)
  // with the crop volume as a predicate inside the search and the index the last insertion left behind
  RegistrationResult registerScan(int method, const PointCloud& scan, const o3ds_crop& scanMatcherCrop, const Eigen::Isometry3d& initialGuess,
                                  double maxCorrespondenceDistance, const ICPConvergenceCriteria& criteria) const {
    auto s = shared();
    std::lock_guard<std::mutex> lck(s->m);
    if (s->size() == 0) throw std::runtime_error("map patch size is zero");  // assert_gt, ScanToMapRegistration.cpp:60
    DeviceCloud in(s->h.get(), nullptr, scan);
    const o3ds_icp_params p = icpParams(method, maxCorrespondenceDistance, criteria);
    o3ds_icp_result r{};
    check(s->h.get(), o3ds_icp_register_dev(s->h.get(), in.id(), s->map, &scanMatcherCrop, initialGuess.matrix().data(), &p, &r));
    return toResult(r);
  }
  void download(PointCloud* out) const {
    auto s = shared();
    std::lock_guard<std::mutex> lck(s->m);
    if (!s->map) {
      *out = PointCloud();
      return;
    }
    downloadCloud(s->h.get(), s->map, out);
  }

 private:
  struct State {
    OwnedHandle h;
    std::mutex m;
    o3ds_cloud map = 0;
    uint64_t version = 0;
    double maxCorr = 0.0;
    ~State() {
      if (map) o3ds_cloud_free(h.get(), map);
    }
    o3ds_cloud id() {  // the (initially empty) device cloud, made on first use
      if (!map) check(h.get(), o3ds_cloud_upload(h.get(), nullptr, nullptr, 0, &map));
      return map;
    }
    size_t size() const {
      size_t n = 0;
      if (map) check(h.get(), o3ds_cloud_size(h.get(), map, &n, nullptr));
      return n;
    }
    void adopt(o3ds_cloud fresh, double maxCorrespondenceDistance) {
      if (map) o3ds_cloud_free(h.get(), map);
      map = fresh;
      maxCorr = maxCorrespondenceDistance;
      if (maxCorrespondenceDistance > 0.0) check(h.get(), o3ds_cloud_build_index(h.get(), map, maxCorrespondenceDistance, 0.0));
      ++version;
    }
  };
  std::shared_ptr<State> shared() const {
    std::lock_guard<std::mutex> lck(ptrMutex_);
    return s_;
  }
  // the state for a mutation: this object's own copy (clouds belong to the handle that made them, so a clone goes through host memory)
  std::shared_ptr<State> unique() {
    std::lock_guard<std::mutex> lck(ptrMutex_);
    if (s_.use_count() > 1) {
      auto fresh = std::make_shared<State>();
      {
        std::lock_guard<std::mutex> l2(s_->m);
        if (s_->map && s_->size() > 0) {
          PointCloud host;
          downloadCloud(s_->h.get(), s_->map, &host);
          DeviceCloud up(fresh->h.get(), host);
          o3ds_cloud copy = 0;
          const double identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
          check(fresh->h.get(), o3ds_transform_cloud(fresh->h.get(), up.id(), identity, &copy));
          fresh->adopt(copy, s_->maxCorr);
        }
        fresh->version = s_->version + 1;
      }
      s_ = fresh;
    }
    return s_;
  }
  mutable std::mutex ptrMutex_;
  std::shared_ptr<State> s_;
};

}  // namespace o3ds
