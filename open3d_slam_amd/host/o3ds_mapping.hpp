// o3ds_mapping.hpp -- C++ host side of Seams 2 and 3: the reference's scan-to-map registration and map-fusion
// classes over the C-ABI of include/o3ds_backend.h, with the submap resident in HBM.  Same names, members, argument meaning and
// error behaviour (std::runtime_error where the reference's assert_* throw) as:
//   ScanToMapRegistration / ScanToMapIcp / ProcessedScans / scanToMapRegistrationFactory / toCloudRegistrationType
//        include/open3d_slam/ScanToMapRegistration.hpp:24-59, src/ScanToMapRegistration.cpp:19-129
//   Submap (insertScan, insertScanDenseMap, carve x2, transform, getMapPointCloud ...)
//        include/open3d_slam/Submap.hpp:27-93, src/Submap.cpp:39-149,208-226
//   VoxelizedPointCloud (the dense map)                 include/open3d_slam/Voxel.hpp:59-76, src/Voxel.cpp:18-114
//   randomDownSample, computeIndicesOfOverlappingPoints include/open3d_slam/helpers.hpp:24,44-46, src/helpers.cpp:102-106,307-332
//   [O3D] GetInformationMatrixFromPointClouds           call sites src/constraint_builders.cpp:70-73, src/PlaceRecognition.cpp:148-149
//   ConstantVelocityMotionCompensation (per-point part) src/MotionCompensation.cpp:64-139
//   MapperParameters and the structs inside it          include/open3d_slam/Parameters.hpp:44-98,145-178
// What is different from the reference, on purpose: mapCloud_ lives on the device (with its NN index) and the host copy that
// getMapPointCloud() returns is refreshed lazily, only when somebody asks for it after the map changed; scanToMapRegistration never
// copies a map patch -- the scan-matcher volume is a predicate inside the search (ScanToMapRegistration.cpp:58-59); the
// cloudRegistration object is a member of ScanToMapIcp instead of a file-static shared by all instances (.cpp:17).
// Header-only; link with -lo3ds_backend.
#pragma once
#include <algorithm>
#include <array>
#include <numeric>
#include <random>

#include "o3ds_adapter.hpp"

namespace o3d_slam {

// ---- Parameters (include/open3d_slam/Parameters.hpp; same names and defaults) --------------------------------------------------
enum class ScanToMapRegistrationType : int { PointToPlaneIcp, PointToPointIcp, GeneralizedIcp };  // :44
static const std::map<std::string, ScanToMapRegistrationType> ScanToMapRegistrationStringToEnumMap{  // :46-49
    {"PointToPlaneIcp", ScanToMapRegistrationType::PointToPlaneIcp},
    {"PointToPointIcp", ScanToMapRegistrationType::PointToPointIcp},
    {"GeneralizedIcp", ScanToMapRegistrationType::GeneralizedIcp}};
struct ScanProcessingParameters {  // :59-64
  double downSamplingRatio_ = 1.0;
  double voxelSize_ = 0.03;
  int pointCloudBufferSize_ = 1;
  ScanCroppingParameters cropper_;
};
struct OdometryParameters {  // :78-83
  CloudRegistrationParameters scanMatcher_;
  ScanProcessingParameters scanProcessing_;
  bool isPublishOdometryMsgs_ = false;
  int odometryBufferSize_ = 1;
};
struct SpaceCarvingParameters {  // :85-92
  double voxelSize_ = 0.1;
  double maxRaytracingLength_ = 20.0;
  double truncationDistance_ = 0.1;
  int carveSpaceEveryNscans_ = 10;
  double minDotProductWithNormal_ = 0.5;
  double neighborhoodRadiusDenseMap_ = 0.1;
};
struct MapBuilderParameters {  // :94-98
  double mapVoxelSize_ = 0.03;
  ScanCroppingParameters cropper_;
  SpaceCarvingParameters carving_;
};
struct ScanToMapRegistrationParameters {  // :145-149
  ScanToMapRegistrationType scanToMapRegType_ = ScanToMapRegistrationType::PointToPlaneIcp;
  double minRefinementFitness_ = 0.7;
  IcpParameters icp_;
};
struct MapperParameters {  // :158-178, the members the hot path reads
  ScanToMapRegistrationParameters scanMatcher_;
  ScanProcessingParameters scanProcessing_;
  double minMovementBetweenMappingSteps_ = 0.0;
  bool isIgnoreMinRefinementFitness_ = false;
  MapBuilderParameters mapBuilder_;
  MapBuilderParameters denseMapBuilder_;
  bool isBuildDenseMap_ = true;
  bool isUseInitialMap_ = false;
  bool isMergeScansIntoMap_ = true;
  int mappingBufferSize_ = 1;
};
struct ConstantVelocityMotionCompensationParameters {  // :192-197
  bool isUndistortInputCloud_ = false;
  bool isSpinningClockwise_ = true;
  double scanDuration_ = 0.1;
  int numPosesVelocityEstimation_ = 3;
};

// ---- helpers (helpers.hpp) -----------------------------------------------------------------------------------------------------
namespace o3ds_detail {
// [O3D] RandomDownSample seeds std::mt19937 from std::random_device, so the reference is not reproducible.  The same
// generator is used here, one per calling thread; setRandomDownSampleSeed() pins the calling thread's for tests and benchmarks.
inline std::mt19937& downSampleRng() {
  thread_local std::mt19937 rng{std::random_device{}()};
  return rng;
}
// the index list [O3D] RandomDownSample builds: iota, shuffle, keep the first int(ratio * n)
inline std::vector<uint32_t> randomKeepList(size_t n, double ratio) {
  std::vector<uint32_t> idx(n);
  std::iota(idx.begin(), idx.end(), 0u);
  std::shuffle(idx.begin(), idx.end(), downSampleRng());
  idx.resize(static_cast<size_t>(static_cast<double>(n) * ratio));
  return idx;
}
// randomDownSample on a device cloud; returns the cloud itself for ratio >= 1 (helpers.cpp:102-105)
inline DevCloud randomDownSampleDev(DevCloud in, double ratio) {
  if (ratio >= 1.0) return in;
  if (ratio < 0.0) throw std::runtime_error("RandomDownSample: sampling_ratio must be in [0, 1]");  // [O3D] LogError
  const std::vector<uint32_t> keep = randomKeepList(in.size(), ratio);
  o3ds_cloud out = 0;
  Handle::check(o3ds_select_by_index(in.handle(), in.id(), keep.data(), keep.size(), &out));
  return DevCloud(out, in.handle());
}
}  // namespace o3ds_detail

inline void setRandomDownSampleSeed(uint32_t seed) { o3ds_detail::downSampleRng().seed(seed); }

inline void randomDownSample(double downSamplingRatio, PointCloud* pcl) {  // helpers.cpp:102-106
  if (downSamplingRatio >= 1.0) return;
  o3ds_detail::randomDownSampleDev(o3ds_detail::DevCloud(*pcl), downSamplingRatio).download(pcl);
}

// computeIndicesOfOverlappingPoints (helpers.cpp:307-332).  Indices come back ascending; the reference's order is its hash map's.
inline void computeIndicesOfOverlappingPoints(const PointCloud& source, const PointCloud& target, const Transform& sourceToTarget,
                                              double voxelSize, size_t minNumPointsPerVoxel, std::vector<size_t>* idxsSource,
                                              std::vector<size_t>* idxsTarget) {
  if (!(minNumPointsPerVoxel >= 1)) throw std::runtime_error("computeIndicesOfOverlappingPoints: minNumPointsPerVoxel");  // assert_ge
  idxsSource->clear();
  idxsTarget->clear();
  if (source.points_.empty() || target.points_.empty()) return;
  o3ds_detail::DevCloud s(source), t(target);
  std::vector<uint64_t> is(source.points_.size()), it(target.points_.size());
  size_t ns = 0, nt = 0;
  o3ds_detail::Handle::check(o3ds_overlap_indices(s.handle(), s.id(), t.id(), o3ds_detail::pose_data(sourceToTarget), voxelSize,
                                                  minNumPointsPerVoxel, is.data(), &ns, it.data(), &nt));
  idxsSource->assign(is.begin(), is.begin() + ns);
  idxsTarget->assign(it.begin(), it.begin() + nt);
}

// [O3D] GetInformationMatrixFromPointClouds(source, target, maxCorrespondenceDistance, transformation): 6x6, symmetric, so the 36
// doubles are Eigen::Matrix6d::data() in either storage order (constraint_builders.cpp:70-73, PlaceRecognition.cpp:148-149)
inline std::array<double, 36> getInformationMatrixFromPointClouds(const PointCloud& source, const PointCloud& target,
                                                                  double maxCorrespondenceDistance, const Transform& transformation) {
  std::array<double, 36> info{};
  o3ds_detail::Handle::check(o3ds_information_matrix(o3ds_detail::Handle::get(), o3ds_detail::xyz(source.points_), source.points_.size(),
                                                     o3ds_detail::xyz(target.points_), target.points_.size(),
                                                     o3ds_detail::pose_data(transformation), maxCorrespondenceDistance, info.data()));
  return info;
}

// The per-point part of ConstantVelocityMotionCompensation::undistortInputPointCloud (MotionCompensation.cpp:64-118,120-139): the
// velocities come from the caller's pose buffer (estimateLinearAndAngularVelocity, :33-58, stays on the host with the buffer).
inline std::shared_ptr<PointCloud> undistortInputPointCloud(const PointCloud& input, const std::array<double, 3>& linearVelocity,
                                                            const std::array<double, 3>& angularVelocityRpy,
                                                            const ConstantVelocityMotionCompensationParameters& params) {
  if (!(params.scanDuration_ > 0.0)) throw std::runtime_error("lidar scanDuration_: ");  // assert_gt, MotionCompensation.cpp:61
  auto output = std::make_shared<PointCloud>(input);
  if (input.points_.empty()) return output;
  PointCloud bare;  // the kernel drops normals (the reference de-skews raw scans and leaves normals_/colors_ as they are)
  bare.points_ = input.points_;
  o3ds_detail::DevCloud d(bare);
  o3ds_detail::Handle::check(o3ds_cloud_undistort(d.handle(), d.id(), linearVelocity.data(), angularVelocityRpy.data(), params.scanDuration_,
                                                  params.isSpinningClockwise_ ? 1 : 0));
  d.download(&bare);
  output->points_ = std::move(bare.points_);
  return output;
}

// ---- VoxelizedPointCloud: the dense map (Voxel.hpp:59-76) ------------------------------------------------------------------------
// A persistent voxel -> {count, sums} table in HBM with its own backend handle (denseMapWorker fills it while mappingWorker works on
// the sparse map, SlamWrapper.cpp:363-386); the caller serialises access as the reference does with denseMapMutex_.
class VoxelizedPointCloud {
 public:
  VoxelizedPointCloud() : VoxelizedPointCloud(0.25) {}  // Voxel.cpp:38: default voxel size 0.25
  explicit VoxelizedPointCloud(double voxelSize) : voxelSize_(voxelSize) {}
  ~VoxelizedPointCloud() { reset(); }
  VoxelizedPointCloud(const VoxelizedPointCloud&) = delete;
  VoxelizedPointCloud& operator=(const VoxelizedPointCloud&) = delete;
  // a fresh, empty map with another voxel size (Submap::update move-assigns a new VoxelizedPointCloud, Submap.cpp:211)
  void reinitialize(double voxelSize) {
    reset();
    voxelSize_ = voxelSize;
  }
  double getVoxelSize() const { return voxelSize_; }
  void insert(const PointCloud& cloud) { insert(cloud, nullptr); }  // Voxel.cpp:66-88
  // insert of o3d_slam::transform(T, cloud) without the intermediate copy (Submap.cpp:81-84)
  void insert(const PointCloud& cloud, const Transform* T) {
    if (cloud.points_.empty()) return;
    o3ds_detail::DevCloud d(cloud, h_.get());
    o3ds_detail::Handle::check(o3ds_dense_map_insert(h_.get(), id(), d.id(), T ? o3ds_detail::pose_data(*T) : nullptr));
  }
  PointCloud toPointCloud() const {  // Voxel.cpp:18-36; voxels in ascending key order
    PointCloud out;
    if (!map_) return out;
    o3ds_cloud c = 0;
    o3ds_detail::Handle::check(o3ds_dense_map_to_cloud(h_.get(), map_, &c));
    o3ds_detail::DevCloud(c, h_.get()).download(&out);
    return out;
  }
  void transform(const Transform& T) {  // Voxel.cpp:49-64, as written there
    if (map_) o3ds_detail::Handle::check(o3ds_dense_map_transform(h_.get(), map_, o3ds_detail::pose_data(T)));
  }
  size_t size() const {
    size_t n = 0;
    if (map_) o3ds_detail::Handle::check(o3ds_dense_map_size(h_.get(), map_, &n));
    return n;
  }
  bool empty() const { return size() == 0; }
  // Submap::carve for the dense map (Submap.cpp:126-136 -> removeDuplicatePointsWithinSameVoxels + getKeysOfCarvedPoints + removeKey);
  // NB the reference hands over the RAW scan (sensor frame) with the map-frame sensor position (Submap.cpp:88); returns #removed voxels
  size_t carve(const PointCloud& scan, const std::array<double, 3>& sensorPosition, const SpaceCarvingParameters& param) {
    if (!map_ || scan.points_.empty()) return 0;
    o3ds_detail::DevCloud d(scan, h_.get());
    size_t removed = 0;
    o3ds_detail::Handle::check(o3ds_dense_map_carve(h_.get(), map_, d.id(), nullptr, sensorPosition.data(), param.neighborhoodRadiusDenseMap_,
                                                    param.maxRaytracingLength_, param.truncationDistance_, &removed));
    return removed;
  }
  // number of points of T * cloud inside an occupied voxel: the numerator of isSwitchingSubmapsConsistant (SubmapCollection.cpp:352-364)
  size_t countPointsInOccupiedVoxels(const PointCloud& cloud, const Transform* T = nullptr) const {
    if (!map_ || cloud.points_.empty()) return 0;
    o3ds_detail::DevCloud d(cloud, h_.get());
    size_t hits = 0;
    o3ds_detail::Handle::check(o3ds_dense_map_count_occupied(h_.get(), map_, d.id(), T ? o3ds_detail::pose_data(*T) : nullptr, &hits));
    return hits;
  }

 private:
  o3ds_dense_map id() {
    if (!map_) o3ds_detail::Handle::check(o3ds_dense_map_create(h_.get(), voxelSize_, &map_));
    return map_;
  }
  void reset() {
    if (map_) o3ds_dense_map_free(h_.get(), map_);
    map_ = 0;
  }
  o3ds_detail::OwnedHandle h_;
  o3ds_dense_map map_ = 0;
  double voxelSize_ = 0.25;
};

// ---- Submap (Submap.hpp:27-93) ----------------------------------------------------------------------------------------------------
class Submap {
 public:
  using PointCloud = open3d::geometry::PointCloud;
  using SubmapId = size_t;

  Submap(size_t id, size_t parentId) : id_(id), parentId_(parentId) { update(params_); }  // Submap.cpp:26-28
  ~Submap() = default;
  Submap(const Submap&) = delete;  // the reference's copy constructor deep-copies the host clouds; a device map is not copied implicitly
  Submap& operator=(const Submap&) = delete;

  void setParameters(const MapperParameters& mapperParams) {  // Submap.cpp:146-149
    params_ = mapperParams;
    update(mapperParams);
  }

  // Submap.cpp:39-75
  bool insertScan(const PointCloud& rawScan, const PointCloud& preProcessedScan, const Transform& mapToRangeSensor, const Time& /*time*/,
                  bool isPerformCarving) {
    if (preProcessedScan.IsEmpty()) return true;
    mapToRangeSensor_ = mapToRangeSensor;
    const double maxCorr = params_.scanMatcher_.icp_.maxCorrespondenceDistance_;
    if (params_.isUseInitialMap_ && map_.size() == 0) {  // :47-52
      map_.setInitialMap(preProcessedScan, params_.mapBuilder_.mapVoxelSize_, maxCorr);
      return true;
    }
    if (isPerformCarving) carve(rawScan, mapToRangeSensor, *mapBuilderCropper_, params_.mapBuilder_.carving_);  // :55-68
    // mapCloud_ += T * scan; setPose; voxelizeInsideCroppingVolume (:69-72, 138-144) + the index the next registration needs.
    // mapVoxelSize_ <= 0 skips the voxelisation in the reference and in o3ds_voxelize_within_volume alike.
    map_.insertScan(preProcessedScan, mapToRangeSensor, params_.mapBuilder_.mapVoxelSize_, mapBuilderCropper_.get(), maxCorr);
    ++nScansInsertedMap_;
    return true;
  }

  // Submap.cpp:77-92.  ColorRangeCropper keeps the default bounds [0, 1]^3 in the reference (nobody calls setMinBounds/setMaxBounds),
  // which every colour produced by rosToOpen3d satisfies except raw intensities > 1; those points are dropped here on the host
  // exactly as ColorRangeCropper::crop does (croppers.cpp:203-236).
  bool insertScanDenseMap(const PointCloud& rawScan, const Transform& mapToRangeSensor, const Time& /*time*/, bool isPerformCarving) {
    denseMapCropper_->setPose(Transform::Identity());
    std::shared_ptr<PointCloud> cropped = denseMapCropper_->crop(rawScan);
    if (cropped->HasColors()) cropped = cropColorRange(*cropped);
    {
      std::lock_guard<std::mutex> lck(denseMapMutex_);
      denseMap_.insert(*cropped, &mapToRangeSensor);
    }
    if (isPerformCarving) {
      std::lock_guard<std::mutex> lck(denseMapMutex_);
      const SpaceCarvingParameters& c = params_.denseMapBuilder_.carving_;
      if (!denseMap_.empty() && nScansInsertedDenseMap_ % static_cast<size_t>(c.carveSpaceEveryNscans_) == 1) {  // :127-129
        const double* m = o3ds_detail::pose_data(mapToRangeSensor);
        denseMap_.carve(rawScan, {{m[12], m[13], m[14]}}, c);
      }
    }
    ++nScansInsertedDenseMap_;
    return true;
  }

  const Transform& getMapToSubmapOrigin() const { return mapToSubmap_; }
  void setMapToSubmapOrigin(const Transform& T) { mapToSubmap_ = T; }
  const Transform& getMapToRangeSensor() const { return mapToRangeSensor_; }
  // the host view of the device map, refreshed only when the map changed since the last call (Submap.cpp:181-183)
  const PointCloud& getMapPointCloud() const {
    std::lock_guard<std::mutex> lck(mirrorMutex_);
    const uint64_t v = map_.version();
    if (v != mirrorVersion_) {
      map_.getMapPointCloud(&mirror_);
      mirrorVersion_ = v;
    }
    return mirror_;
  }
  PointCloud getMapPointCloudCopy() const {  // Submap.cpp:184-188
    PointCloud copy;
    map_.getMapPointCloud(&copy);
    return copy;
  }
  const VoxelizedPointCloud& getDenseMap() const { return denseMap_; }
  bool isEmpty() const { return map_.size() == 0; }  // Submap.cpp:218-220
  size_t getId() const { return id_; }
  size_t getParentId() const { return parentId_; }
  size_t getNumScansInsertedMap() const { return nScansInsertedMap_; }

  void transform(const Transform& T) {  // Submap.cpp:94-107 (sparseMapCloud_ / features belong to place recognition: out of scope)
    map_.transform(T, params_.scanMatcher_.icp_.maxCorrespondenceDistance_);
    {
      std::lock_guard<std::mutex> lck(denseMapMutex_);
      denseMap_.transform(T);
    }
    mapToRangeSensor_ = mapToRangeSensor_ * T;
  }

  // the device map (for ScanToMapIcp and saveToFile)
  const DeviceSubmap& deviceMap() const { return map_; }
  bool saveToFile(const std::string& filename) const { return map_.saveToFile(filename); }

 private:
  // Submap::carve for the sparse map (Submap.cpp:109-125): every carveSpaceEveryNscans_-th insertion, never on an empty map.
  // mapBuilderCropper_ still holds the pose of the PREVIOUS insertion here (setPose comes after, :71), as in the reference.
  void carve(const PointCloud& rawScan, const Transform& mapToRangeSensor, const CroppingVolume& cropper, const SpaceCarvingParameters& p) {
    if (map_.size() == 0 || !(nScansInsertedMap_ % static_cast<size_t>(p.carveSpaceEveryNscans_) == 1)) return;
    map_.carve(rawScan, mapToRangeSensor, cropper, p.voxelSize_, p.maxRaytracingLength_, p.truncationDistance_, p.minDotProductWithNormal_);
  }
  void update(const MapperParameters& p) {  // Submap.cpp:208-216 (voxelMap_ of the adjacency check: see VoxelizedPointCloud::countPointsInOccupiedVoxels)
    mapBuilderCropper_ = croppingVolumeFactory(p.mapBuilder_.cropper_);
    denseMapCropper_ = croppingVolumeFactory(p.denseMapBuilder_.cropper_);
    std::lock_guard<std::mutex> lck(denseMapMutex_);
    denseMap_.reinitialize(p.denseMapBuilder_.mapVoxelSize_);
  }
  static std::shared_ptr<PointCloud> cropColorRange(const PointCloud& cloud) {  // ColorRangeCropper::crop, bounds [0,1]^3
    auto out = std::make_shared<PointCloud>();
    const bool hn = cloud.HasNormals();
    for (size_t i = 0; i < cloud.points_.size(); ++i) {
      const auto& c = cloud.colors_[i];
      if (0.0 <= c[0] && c[0] <= 1.0 && 0.0 <= c[1] && c[1] <= 1.0 && 0.0 <= c[2] && c[2] <= 1.0) {
        out->points_.push_back(cloud.points_[i]);
        out->colors_.push_back(c);
        if (hn) out->normals_.push_back(cloud.normals_[i]);
      }
    }
    return out;
  }

  DeviceSubmap map_;
  mutable PointCloud mirror_;
  mutable uint64_t mirrorVersion_ = ~uint64_t(0);
  mutable std::mutex mirrorMutex_;
  Transform mapToSubmap_ = Transform::Identity();
  Transform mapToRangeSensor_ = Transform::Identity();
  std::shared_ptr<CroppingVolume> denseMapCropper_, mapBuilderCropper_;
  MapperParameters params_;
  size_t nScansInsertedMap_ = 0;
  size_t nScansInsertedDenseMap_ = 0;
  size_t id_ = 0;
  size_t parentId_ = 0;
  VoxelizedPointCloud denseMap_;
  mutable std::mutex denseMapMutex_;
};

// ---- ScanToMapRegistration seam (ScanToMapRegistration.hpp:24-59) -----------------------------------------------------------------
struct ProcessedScans {
  PointCloudPtr merge_;
  PointCloudPtr match_;
};

class ScanToMapRegistration {
 public:
  ScanToMapRegistration() = default;
  virtual ~ScanToMapRegistration() = default;
  virtual ProcessedScans processForScanMatchingAndMerging(const PointCloud& in, const Transform& mapToRangeSensor) const = 0;
  virtual RegistrationResult scanToMapRegistration(const PointCloud& scan, const Submap& activeSubmap, const Transform& mapToRangeSensor,
                                                   const Transform& initialGuess) const = 0;
  virtual bool isMergeScanValid(const PointCloud& in) const = 0;
  virtual void prepareInitialMap(PointCloud* map) const = 0;
};

inline CloudRegistrationParameters toCloudRegistrationType(const ScanToMapRegistrationParameters& p) {  // ScanToMapRegistration.cpp:104-127
  CloudRegistrationParameters retVal;
  retVal.icp_ = p.icp_;
  switch (p.scanToMapRegType_) {
    case ScanToMapRegistrationType::PointToPlaneIcp:
      retVal.regType_ = CloudRegistrationType::PointToPlaneIcp;
      break;
    case ScanToMapRegistrationType::PointToPointIcp:
      retVal.regType_ = CloudRegistrationType::PointToPointIcp;
      break;
    case ScanToMapRegistrationType::GeneralizedIcp:
      retVal.regType_ = CloudRegistrationType::GeneralizedIcp;
      break;
    default:
      throw std::runtime_error(
          "Conversion not possible from ScanToMapRegistrationParameters to CloudRegistrationParameters, for this particular scan to map "
          "reg type");
  }
  return retVal;
}

class ScanToMapIcp : public ScanToMapRegistration {
 public:
  ScanToMapIcp() { update(params_); }
  ~ScanToMapIcp() override = default;
  void setParameters(const MapperParameters& p) {  // .cpp:24-27
    params_ = p;
    update(params_);
  }

  // .cpp:42-54: crop -> voxelize -> normals -> random down-sample on the device, one upload of the raw scan, two downloads
  ProcessedScans processForScanMatchingAndMerging(const PointCloud& in, const Transform& /*mapToRangeSensor*/) const final {
    ProcessedScans retVal;
    retVal.merge_ = std::make_shared<PointCloud>();
    retVal.match_ = std::make_shared<PointCloud>();
    if (!in.points_.empty()) {
      o3ds_detail::DevCloud wide = preprocessDev(in);
      wide.download(retVal.merge_.get());
      scanMatcherCropper_->setPose(Transform::Identity());
      const o3ds_crop c = scanMatcherCropper_->toAbi();
      o3ds_cloud narrow = 0;
      o3ds_detail::Handle::check(o3ds_crop_cloud(wide.handle(), wide.id(), &c, &narrow));
      o3ds_detail::DevCloud(narrow, wide.handle()).download(retVal.match_.get());
    }
    if (!(retVal.match_->points_.size() > 0)) throw std::runtime_error("ScanToMapIcp::narrow cropped size is zero");  // assert_gt :51
    if (!(retVal.merge_->points_.size() > 0)) throw std::runtime_error("ScanToMapIcp::wideCropped cropped size is zero");
    return retVal;
  }

  // .cpp:55-62 against the device-resident submap: no map patch is copied, no KD-tree is rebuilt
  RegistrationResult scanToMapRegistration(const PointCloud& scan, const Submap& activeSubmap, const Transform& mapToRangeSensor,
                                           const Transform& initialGuess) const final {
    scanMatcherCropper_->setPose(mapToRangeSensor);
    o3ds_icp_params p{};
    if (cloudRegistration_->toAbi(&p)) return activeSubmap.deviceMap().registerScan(scan, scanMatcherCropper_->toAbi(), initialGuess, p);
    // a CloudRegistration the backend does not know: the reference's own sequence on host clouds
    const PointCloudPtr mapPatch = scanMatcherCropper_->crop(activeSubmap.getMapPointCloud());
    if (!(mapPatch->points_.size() > 0)) throw std::runtime_error("map patch size is zero");
    return cloudRegistration_->registerClouds(scan, *mapPatch, initialGuess);
  }

  bool isMergeScanValid(const PointCloud& in) const final {  // .cpp:64-80 (covariances are never materialised here: normals stand for them)
    switch (params_.scanMatcher_.scanToMapRegType_) {
      case ScanToMapRegistrationType::PointToPlaneIcp:
        return in.HasNormals();
      case ScanToMapRegistrationType::PointToPointIcp:
        return true;
      case ScanToMapRegistrationType::GeneralizedIcp:
        return in.HasNormals();
      default:
        throw std::runtime_error("cannot check whether merge scan is valid for this registration type");
    }
  }

  void prepareInitialMap(PointCloud* map) const final { cloudRegistration_->estimateNormalsOrCovariancesIfNeeded(map); }  // .cpp:81-84

  // ScanToMapIcp::preprocess (.cpp:35-40) as the reference exposes it to nobody (private there); public here for tests
  PointCloudPtr preprocess(const PointCloud& in) const {
    auto out = std::make_shared<PointCloud>();
    if (!in.points_.empty()) preprocessDev(in).download(out.get());
    return out;
  }
  const CloudRegistration& getCloudRegistration() const { return *cloudRegistration_; }

 private:
  o3ds_detail::DevCloud preprocessDev(const PointCloud& in) const {
    o3ds_detail::DevCloud raw(in);
    const o3ds_handle h = raw.handle();
    const o3ds_crop c = mapBuilderCropper_->toAbi();
    o3ds_cloud id = 0;
    // mapBuilderCropper_->crop(in) and o3d_slam::voxelize (helpers.cpp:107-113) in one call, the same cloud bit for bit
    o3ds_detail::Handle::check(o3ds_crop_voxel_down_sample(h, raw.id(), &c, params_.scanProcessing_.voxelSize_, &id));
    o3ds_detail::DevCloud cropped(id, h);
    raw.reset();
    if (cropped.size() > 0 && !cloudRegistration_->estimateNormalsOrCovariancesIfNeededDev(h, cropped.id())) {
      PointCloud host;  // unknown registration: its own host-side hook, then back to the device
      cropped.download(&host);
      cloudRegistration_->estimateNormalsOrCovariancesIfNeeded(&host);
      cropped = o3ds_detail::DevCloud(host, h);
    }
    return o3ds_detail::randomDownSampleDev(std::move(cropped), params_.scanProcessing_.downSamplingRatio_);  // RandomDownSample(ratio)
  }
  void update(const MapperParameters& p) {  // .cpp:28-32
    mapBuilderCropper_ = croppingVolumeFactory(params_.mapBuilder_.cropper_);
    scanMatcherCropper_ = croppingVolumeFactory(params_.scanProcessing_.cropper_);
    cloudRegistration_ = cloudRegistrationFactory(toCloudRegistrationType(p.scanMatcher_));
  }

  MapperParameters params_;
  std::shared_ptr<CroppingVolume> scanMatcherCropper_;
  std::shared_ptr<CroppingVolume> mapBuilderCropper_;
  std::shared_ptr<CloudRegistration> cloudRegistration_;
};

inline std::unique_ptr<ScanToMapIcp> createScanToMapIcp(const MapperParameters& p) {  // .cpp:86-90
  auto ret = std::make_unique<ScanToMapIcp>();
  ret->setParameters(p);
  return ret;
}
inline std::unique_ptr<ScanToMapRegistration> scanToMapRegistrationFactory(const MapperParameters& p) {  // .cpp:91-102
  switch (p.scanMatcher_.scanToMapRegType_) {
    case ScanToMapRegistrationType::PointToPlaneIcp:
    case ScanToMapRegistrationType::GeneralizedIcp:
    case ScanToMapRegistrationType::PointToPointIcp:
      return createScanToMapIcp(p);
    default:
      throw std::runtime_error("scanToMapRegistrationFactory: unknown type of registration scan to map");
  }
}

}  // namespace o3d_slam
