// C entry points into open3d_slam's OWN sources (croppers.cpp, helpers.cpp, Voxel.cpp, VoxelHashMap.cpp, MotionCompensation.cpp, ...),
// compiled unchanged from /root/reference against the stand-in headers of oracle/ref_build/shim (oracle/ref_build/Makefile ->
// oracle/_ref/libo3dslam_ref.so).  This file only marshals plain arrays into the reference's types and calls the reference's functions;
// it holds no algorithm of its own.  tests/test_oracle_vs_reference.py checks oracle/o3d_oracle.c against it; scripts under
// tests/golden/ turn its outputs into fixtures for the GPU tests.  Test infrastructure only.
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <thread>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "open3d_slam/Mapper.hpp"
#include "open3d_slam/MotionCompensation.hpp"
#include "open3d_slam/Odometry.hpp"
#include "open3d_slam/Submap.hpp"
#include "open3d_slam/SubmapCollection.hpp"
#include "open3d_slam/Parameters.hpp"
#include "open3d_slam/TransformInterpolationBuffer.hpp"
#include "open3d_slam/Voxel.hpp"
#include "open3d_slam/croppers.hpp"
#include "open3d_slam/helpers.hpp"
#include "open3d_slam/math.hpp"
#include "open3d_slam/time.hpp"

#include <cxxabi.h>
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>

using open3d::geometry::PointCloud;

namespace {
// Where a worker's WALL time goes (REF_SAMPLE_OUT=<file> in the environment, scripts/sample_patched_reference.py): a 2 kHz timer signal to
// the calling thread records its call stack; the report counts, per function, the samples whose stack holds it.  No perf on the GPU box.
struct StackSampler {
  static constexpr int kDepth = 40, kMax = 1 << 15;
  void* pc[kMax][kDepth];
  int depth[kMax];
  std::atomic<int> n{0};
  timer_t timer{};
  bool on = false;
  static StackSampler*& self() {
    static StackSampler* p = nullptr;
    return p;
  }
  static void tick(int) {
    StackSampler* s = self();
    if (!s) return;
    const int i = s->n.fetch_add(1);
    if (i < kMax) s->depth[i] = backtrace(s->pc[i], kDepth);
  }
  void start() {
    void* warm[4];
    backtrace(warm, 4);  // (loads the unwinder outside the handler)
    self() = this;
    struct sigaction sa {};
    sa.sa_handler = &StackSampler::tick;
    sa.sa_flags = SA_RESTART;
    sigaction(SIGRTMIN + 3, &sa, nullptr);
    struct sigevent ev {};
    ev.sigev_notify = SIGEV_THREAD_ID;
    ev.sigev_signo = SIGRTMIN + 3;
    ev._sigev_un._tid = (pid_t)syscall(SYS_gettid);
    if (timer_create(CLOCK_MONOTONIC, &ev, &timer) != 0) return;
    struct itimerspec its {};
    its.it_interval.tv_nsec = its.it_value.tv_nsec = 500000;
    timer_settime(timer, 0, &its, nullptr);
    on = true;
  }
  void stop(const char* path, const char* title) {
    if (!on) return;
    timer_delete(timer);
    on = false;
    self() = nullptr;
    const int total = std::min(n.load(), (int)kMax);
    std::map<std::string, int> inclusive, innermost;
    for (int i = 0; i < total; ++i) {
      std::vector<std::string> seen;
      bool first = true;
      for (int d = 0; d < depth[i]; ++d) {
        Dl_info info{};
        std::string name = "?";
        if (dladdr((char*)pc[i][d] - 1, &info)) {
          if (info.dli_sname) {
            int st = 0;
            char* dm = abi::__cxa_demangle(info.dli_sname, nullptr, nullptr, &st);
            name = st == 0 && dm ? dm : info.dli_sname;
            std::free(dm);
          } else if (info.dli_fname) {
            const char* b = std::strrchr(info.dli_fname, '/');
            name = std::string("[") + (b ? b + 1 : info.dli_fname) + "]";
          }
        }
        if (name.find("StackSampler") != std::string::npos || name.find("__restore_rt") != std::string::npos || name == "[libc.so.6]" && d < 3) continue;
        if (name.size() > 150) name.resize(150);
        if (first) ++innermost[name], first = false;
        if (std::find(seen.begin(), seen.end(), name) == seen.end()) seen.push_back(name), ++inclusive[name];
      }
    }
    std::FILE* f = std::fopen(path, "a");
    if (!f) return;
    std::fprintf(f, "== %s: %d samples at 2 kHz of wall time on the calling thread\n-- samples whose stack holds the function (inclusive)\n", title, total);
    auto dump = [&](const std::map<std::string, int>& m, int top) {
      std::vector<std::pair<int, std::string>> v;
      for (const auto& kv : m) v.emplace_back(kv.second, kv.first);
      std::sort(v.rbegin(), v.rend());
      for (int i = 0; i < (int)v.size() && i < top; ++i) std::fprintf(f, "%6.1f %%  %s\n", 100.0 * v[i].first / std::max(total, 1), v[i].second.c_str());
    };
    dump(inclusive, 70);
    std::fprintf(f, "-- innermost resolved frame\n");
    dump(innermost, 30);
    std::fclose(f);
  }
};
PointCloud make_cloud(const double* pts, const double* nrm, const double* col, size_t n) {
  PointCloud c;
  c.points_.resize(n);
  for (size_t i = 0; i < n; ++i) c.points_[i] = Eigen::Vector3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  if (nrm) {
    c.normals_.resize(n);
    for (size_t i = 0; i < n; ++i) c.normals_[i] = Eigen::Vector3d(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
  }
  if (col) {
    c.colors_.resize(n);
    for (size_t i = 0; i < n; ++i) c.colors_[i] = Eigen::Vector3d(col[3 * i], col[3 * i + 1], col[3 * i + 2]);
  }
  return c;
}
void store(const std::vector<Eigen::Vector3d>& v, double* out) {
  if (!out) return;
  for (size_t i = 0; i < v.size(); ++i)
    for (int a = 0; a < 3; ++a) out[3 * i + a] = v[i](a);
}
Eigen::Matrix4d matrix_from_colmajor(const double T[16]) {
  Eigen::Matrix4d M;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) M(r, c) = T[c * 4 + r];
  return M;
}
// kind: 0 = CroppingVolume (the base class: everything is inside), 1 MaxRadius, 2 MinRadius, 3 MinMaxRadius, 4 Cylinder (oracle numbering)
std::unique_ptr<o3d_slam::CroppingVolume> make_cropper(int kind, double rmin, double rmax, double zmin, double zmax, const double pose[16], int invert) {
  o3d_slam::ScanCroppingParameters p;
  p.croppingMinRadius_ = rmin;
  p.croppingMaxRadius_ = rmax;
  p.croppingMinZ_ = zmin;
  p.croppingMaxZ_ = zmax;
  std::unique_ptr<o3d_slam::CroppingVolume> c;
  switch (kind) {
    case 0:
      c = std::make_unique<o3d_slam::CroppingVolume>();
      break;
    case 1:
      p.cropperName_ = "MaxRadius";
      c = o3d_slam::croppingVolumeFactory(p);
      break;
    case 2:
      p.cropperName_ = "MinRadius";
      c = o3d_slam::croppingVolumeFactory(p);
      break;
    case 3:
      p.cropperName_ = "MinMaxRadius";
      c = o3d_slam::croppingVolumeFactory(p);
      break;
    default:
      p.cropperName_ = "Cylinder";
      c = o3d_slam::croppingVolumeFactory(p);
      break;
  }
  c->setPose(Eigen::Isometry3d(matrix_from_colmajor(pose)));
  c->setIsInvertVolume(invert != 0);
  return c;
}
}  // namespace

extern "C" {

// CroppingVolume::getIndicesWithinVolume + CroppingVolume::crop (croppers.cpp:65-106); returns the kept count
size_t ref_crop(int kind, double rmin, double rmax, double zmin, double zmax, const double pose[16], int invert, const double* pts, const double* nrm,
                const double* col, size_t n, int64_t* out_idx, double* out_pts, double* out_nrm, double* out_col) {
  const auto cropper = make_cropper(kind, rmin, rmax, zmin, zmax, pose, invert);
  const PointCloud cloud = make_cloud(pts, nrm, col, n);
  const auto idx = cropper->getIndicesWithinVolume(cloud);
  const auto cropped = cropper->crop(cloud);
  if (cropped->points_.size() != idx.size()) return (size_t)-1;
  if (out_idx)
    for (size_t i = 0; i < idx.size(); ++i) out_idx[i] = (int64_t)idx[i];
  store(cropped->points_, out_pts);
  if (nrm) store(cropped->normals_, out_nrm);
  if (col) store(cropped->colors_, out_col);
  return idx.size();
}

// voxelizeWithinCroppingVolume (helpers.cpp:115-183); output in the reference's own order (pass-through points, then the hash map's
// iteration order); returns the count
size_t ref_voxelize_within_cropping_volume(double voxel, int kind, double rmin, double rmax, double zmin, double zmax, const double pose[16], int invert,
                                           const double* pts, const double* nrm, const double* col, size_t n, double* out_pts, double* out_nrm,
                                           double* out_col) {
  const auto cropper = make_cropper(kind, rmin, rmax, zmin, zmax, pose, invert);
  const PointCloud cloud = make_cloud(pts, nrm, col, n);
  const auto out = o3d_slam::voxelizeWithinCroppingVolume(voxel, *cropper, cloud);
  store(out->points_, out_pts);
  if (nrm) store(out->normals_, out_nrm);
  if (col) store(out->colors_, out_col);
  return out->points_.size();
}

// o3d_slam::transform (helpers.cpp:273-305); the out arrays hold 2 n points: for a T within 1e-4 of the identity the reference returns the
// input cloud FOLLOWED by the transformed points (helpers.cpp:276-279 copies the cloud and the loop then appends to it).  Returns the
// number of points; *has_colors_out = whether the result still counts as coloured (its colour array keeps n entries)
size_t ref_transform(const double T[16], const double* pts, const double* nrm, const double* col, size_t n, double* out_pts, double* out_nrm,
                     int* has_colors_out) {
  const auto out = o3d_slam::transform(matrix_from_colmajor(T), make_cloud(pts, nrm, col, n));
  store(out->points_, out_pts);
  if (nrm) store(out->normals_, out_nrm);
  if (has_colors_out) *has_colors_out = out->HasColors() ? 1 : 0;
  return out->points_.size();
}

// getIdxsOfCarvedPoints (helpers.cpp:221-271); subset may be null (all map points); returns the number of ids (unordered)
size_t ref_carved_idxs(const double* scan, size_t n_scan, const double sensor[3], const double* map_pts, const double* map_nrm, size_t N,
                       const int64_t* subset, size_t n_subset, double voxel, double max_length, double truncation, double min_dot, int64_t* out_ids) {
  o3d_slam::SpaceCarvingParameters p;
  p.voxelSize_ = voxel;
  p.maxRaytracingLength_ = max_length;
  p.truncationDistance_ = truncation;
  p.minDotProductWithNormal_ = min_dot;
  const PointCloud s = make_cloud(scan, nullptr, nullptr, n_scan), m = make_cloud(map_pts, map_nrm, nullptr, N);
  const Eigen::Vector3d sp(sensor[0], sensor[1], sensor[2]);
  std::vector<size_t> ids;
  if (subset) {
    std::vector<size_t> sub(subset, subset + n_subset);
    ids = o3d_slam::getIdxsOfCarvedPoints(s, m, sp, sub, p);
  } else {
    ids = o3d_slam::getIdxsOfCarvedPoints(s, m, sp, p);
  }
  for (size_t i = 0; i < ids.size(); ++i) out_ids[i] = (int64_t)ids[i];
  return ids.size();
}

// computeIndicesOfOverlappingPoints (helpers.cpp:307-332); the index lists in the reference's order
void ref_overlap(const double* src, size_t n_src, const double* tgt, size_t n_tgt, const double T[16], double voxel, size_t min_points, int64_t* out_src,
                 size_t* n_out_src, int64_t* out_tgt, size_t* n_out_tgt) {
  std::vector<size_t> is, it;
  o3d_slam::computeIndicesOfOverlappingPoints(make_cloud(src, nullptr, nullptr, n_src), make_cloud(tgt, nullptr, nullptr, n_tgt),
                                              o3d_slam::Transform(matrix_from_colmajor(T)), voxel, min_points, &is, &it);
  for (size_t i = 0; i < is.size(); ++i) out_src[i] = (int64_t)is[i];
  for (size_t i = 0; i < it.size(); ++i) out_tgt[i] = (int64_t)it[i];
  *n_out_src = is.size();
  *n_out_tgt = it.size();
}

// VoxelizedPointCloud::insert (n_batches consecutive ranges of the input, as consecutive scans) [+ ::transform] + toPointCloud (Voxel.cpp:49-114), plus the
// voxel keys and counts in the same (hash map) order; returns the number of voxels
size_t ref_dense_fuse(const double* pts, const double* nrm, size_t n, double voxel, int n_batches, const double* T_after /* may be null */, double* out_pts,
                      double* out_nrm, int32_t* out_counts, int32_t* out_keys) {
  o3d_slam::VoxelizedPointCloud map(Eigen::Vector3d::Constant(voxel));
  if (n_batches < 1) n_batches = 1;
  for (int b = 0; b < n_batches; ++b) {
    const size_t lo = n * (size_t)b / (size_t)n_batches, hi = n * (size_t)(b + 1) / (size_t)n_batches;
    map.insert(make_cloud(pts + 3 * lo, nrm ? nrm + 3 * lo : nullptr, nullptr, hi - lo));
  }
  if (T_after) map.transform(o3d_slam::Transform(matrix_from_colmajor(T_after)));  // VoxelizedPointCloud::transform (Voxel.cpp:49-64), as written
  const PointCloud out = map.toPointCloud();
  store(out.points_, out_pts);
  if (nrm) store(out.normals_, out_nrm);
  size_t k = 0;
  for (const auto& v : map.voxels_) {
    if (v.second.numAggregatedPoints_ > 0) {
      if (out_counts) out_counts[k] = v.second.numAggregatedPoints_;
      if (out_keys)
        for (int a = 0; a < 3; ++a) out_keys[3 * k + a] = v.first(a);
      ++k;
    }
  }
  return out.points_.size() == k ? k : (size_t)-1;
}

// Submap::carve for the dense map (Submap.cpp:126-136): removeDuplicatePointsWithinSameVoxels (Voxel.cpp:162-191) on the scan, then
// getKeysOfCarvedPoints (helpers.cpp:347-377) against a VoxelizedPointCloud holding map_pts; returns the number of keys (3 ints each)
size_t ref_dense_carve_keys(const double* scan, size_t n_scan, const double sensor[3], const double* map_pts, size_t N, double voxel, double radius,
                            double max_length, double truncation, int dedup_scan, int32_t* out_keys, size_t cap) {
  o3d_slam::SpaceCarvingParameters p;
  p.maxRaytracingLength_ = max_length;
  p.truncationDistance_ = truncation;
  p.neighborhoodRadiusDenseMap_ = radius;
  o3d_slam::VoxelizedPointCloud map(Eigen::Vector3d::Constant(voxel));
  map.insert(make_cloud(map_pts, nullptr, nullptr, N));
  PointCloud s = make_cloud(scan, nullptr, nullptr, n_scan);
  if (dedup_scan) s = *o3d_slam::removeDuplicatePointsWithinSameVoxels(s, Eigen::Vector3d::Constant(voxel));
  const auto keys = o3d_slam::getKeysOfCarvedPoints(s, map, Eigen::Vector3d(sensor[0], sensor[1], sensor[2]), p);
  for (size_t i = 0; i < keys.size() && i < cap; ++i)
    for (int a = 0; a < 3; ++a) out_keys[3 * i + a] = keys[i](a);
  return keys.size();
}

// getVoxelIdx (VoxelHashMap.hpp:47-50), the world-anchored voxel index every map structure of the reference uses
void ref_voxel_idx(const double p[3], double voxel, int32_t out[3]) {
  const auto k = o3d_slam::getVoxelIdx(Eigen::Vector3d(p[0], p[1], p[2]), o3d_slam::fromVoxelSize(Eigen::Vector3d::Constant(voxel)));
  for (int a = 0; a < 3; ++a) out[a] = k(a);
}

int ref_is_valid_color(const double c[3]) { return o3d_slam::isValidColor(Eigen::Vector3d(c[0], c[1], c[2])) ? 1 : 0; }
double ref_icp_max_correspondence_distance(double voxel) { return o3d_slam::icpMaxCorrespondenceDistance(voxel); }
double ref_information_matrix_max_correspondence_distance(double voxel) { return o3d_slam::informationMatrixMaxCorrespondenceDistance(voxel); }

// ConstantVelocityMotionCompensation::undistortInputPointCloud (MotionCompensation.cpp:64-139) with the velocity it would estimate from
// two buffered poses `dt` apart (estimateLinearAndAngularVelocity, :27-57): start = identity, finish = (xyz, rpy); out = the moved points,
// vel_out = {linear velocity, angular velocity (rpy)} as estimated
void ref_undistort(const double* pts, size_t n, const double finish_xyz[3], const double finish_rpy[3], double dt, double scan_duration, int clockwise,
                   double* out_pts, double vel_out[6]) {
  o3d_slam::TransformInterpolationBuffer buffer;
  const o3d_slam::Time t0 = o3d_slam::fromUniversal(1000000);
  const o3d_slam::Time t1 = t0 + o3d_slam::fromSeconds(dt);
  buffer.push(t0, o3d_slam::Transform::Identity());
  buffer.push(t1, o3d_slam::fromXYZandRPY(Eigen::Vector3d(finish_xyz[0], finish_xyz[1], finish_xyz[2]),
                                          Eigen::Vector3d(finish_rpy[0], finish_rpy[1], finish_rpy[2])));
  o3d_slam::ConstantVelocityMotionCompensation mc(buffer);
  o3d_slam::ConstantVelocityMotionCompensationParameters p;
  p.scanDuration_ = scan_duration;
  p.isSpinningClockwise_ = clockwise != 0;
  p.numPosesVelocityEstimation_ = 1;
  mc.setParameters(p);
  const auto out = mc.undistortInputPointCloud(make_cloud(pts, nullptr, nullptr, n), t1 + o3d_slam::fromSeconds(0.05));
  store(out->points_, out_pts);
  if (vel_out) {
    // the same estimate, through the public pieces it is made of (the method itself is private)
    const auto finish = buffer.latest_measurement();
    const auto start = buffer.latest_measurement(1);
    const o3d_slam::Transform dT = start.transform_.inverse() * finish.transform_;
    const double d = o3d_slam::toSeconds(finish.time_ - start.time_);
    const Eigen::Vector3d lin = dT.translation() / (d + 1e-6);
    const Eigen::Vector3d ang = o3d_slam::toRPY(Eigen::Quaterniond(dT.rotation()).normalized()) / (d + 1e-6);
    for (int a = 0; a < 3; ++a) vel_out[a] = lin(a), vel_out[3 + a] = ang(a);
  }
}

// ---- the reference's own frame loop: LidarOdometry::addRangeScan (Odometry.cpp:32-79) then Mapper::addRangeMeasurement (Mapper.cpp:101-181)
// per scan, wired as SlamWrapper wires them (SlamWrapper.cpp:179-186: the mapper reads the odometry's pose buffer), one worker after the
// other.  All of it is the reference's code; the Open3D algorithms underneath are served by the oracle (open3d_served_by_oracle.cpp).
struct ref_slam_params {
  // odometry (OdometryParameters): scan processing + ICP
  double odo_voxel, odo_ratio, odo_rmin, odo_rmax, odo_max_corr, odo_knn_radius;
  int32_t odo_knn, odo_max_iter;
  // mapper (MapperParameters): scan processing, scan matcher, map builder
  double map_voxel, map_ratio, map_rmin, map_rmax, map_max_corr, map_knn_radius, min_refinement_fitness, min_movement;
  int32_t map_knn, map_max_iter;
  double builder_voxel, builder_rmin, builder_rmax;
  // carving (SpaceCarvingParameters) and submaps
  double carve_voxel, carve_max_length, carve_truncation, carve_min_dot;
  int32_t carve_every_n_scans;
  double submap_radius;
  int32_t generalized;  // 1: cloud_registration_type / scan_to_map_refinement_type = GeneralizedIcp (the shipped Lua), 0: PointToPlaneIcp
};
struct RefSlam {
  std::shared_ptr<o3d_slam::LidarOdometry> odometry;
  std::shared_ptr<o3d_slam::SubmapCollection> submaps;
  std::shared_ptr<o3d_slam::Mapper> mapper;
};

// open3d_slam reports on std::cout ("Created submap ...", carving statistics): silenced for a caller whose own standard output is a contract
// (bench.py prints one JSON line); 0 restores it
void ref_set_quiet(int on) {
  if (on)
    std::cout.setstate(std::ios_base::failbit);
  else
    std::cout.clear();
}

void* ref_slam_create(const ref_slam_params* q) {
  o3d_slam::OdometryParameters op;
  op.scanProcessing_.voxelSize_ = q->odo_voxel;
  op.scanProcessing_.downSamplingRatio_ = q->odo_ratio;
  op.scanProcessing_.cropper_.cropperName_ = "MinMaxRadius";
  op.scanProcessing_.cropper_.croppingMinRadius_ = q->odo_rmin;
  op.scanProcessing_.cropper_.croppingMaxRadius_ = q->odo_rmax;
  op.scanMatcher_.regType_ = q->generalized ? o3d_slam::CloudRegistrationType::GeneralizedIcp : o3d_slam::CloudRegistrationType::PointToPlaneIcp;
  op.scanMatcher_.icp_.maxCorrespondenceDistance_ = q->odo_max_corr;
  op.scanMatcher_.icp_.maxDistanceKnn_ = q->odo_knn_radius;
  op.scanMatcher_.icp_.knn_ = q->odo_knn;
  op.scanMatcher_.icp_.maxNumIter_ = q->odo_max_iter;
  o3d_slam::MapperParameters mp;
  mp.scanProcessing_.voxelSize_ = q->map_voxel;
  mp.scanProcessing_.downSamplingRatio_ = q->map_ratio;
  mp.scanProcessing_.cropper_.cropperName_ = "MinMaxRadius";
  mp.scanProcessing_.cropper_.croppingMinRadius_ = q->map_rmin;
  mp.scanProcessing_.cropper_.croppingMaxRadius_ = q->map_rmax;
  mp.scanMatcher_.scanToMapRegType_ =
      q->generalized ? o3d_slam::ScanToMapRegistrationType::GeneralizedIcp : o3d_slam::ScanToMapRegistrationType::PointToPlaneIcp;
  mp.scanMatcher_.minRefinementFitness_ = q->min_refinement_fitness;
  mp.scanMatcher_.icp_.maxCorrespondenceDistance_ = q->map_max_corr;
  mp.scanMatcher_.icp_.maxDistanceKnn_ = q->map_knn_radius;
  mp.scanMatcher_.icp_.knn_ = q->map_knn;
  mp.scanMatcher_.icp_.maxNumIter_ = q->map_max_iter;
  mp.minMovementBetweenMappingSteps_ = q->min_movement;
  mp.mapBuilder_.mapVoxelSize_ = q->builder_voxel;
  mp.mapBuilder_.cropper_.cropperName_ = "MinMaxRadius";
  mp.mapBuilder_.cropper_.croppingMinRadius_ = q->builder_rmin;
  mp.mapBuilder_.cropper_.croppingMaxRadius_ = q->builder_rmax;
  mp.mapBuilder_.carving_.voxelSize_ = q->carve_voxel;
  mp.mapBuilder_.carving_.maxRaytracingLength_ = q->carve_max_length;
  mp.mapBuilder_.carving_.truncationDistance_ = q->carve_truncation;
  mp.mapBuilder_.carving_.minDotProductWithNormal_ = q->carve_min_dot;
  mp.mapBuilder_.carving_.carveSpaceEveryNscans_ = q->carve_every_n_scans;
  mp.submaps_.radius_ = q->submap_radius;
  mp.isBuildDenseMap_ = false;
  mp.isAttemptLoopClosures_ = false;
  mp.isPrintTimingStatistics_ = false;
  auto* s = new RefSlam();
  s->odometry = std::make_shared<o3d_slam::LidarOdometry>();
  s->odometry->setParameters(op);
  s->submaps = std::make_shared<o3d_slam::SubmapCollection>();
  s->mapper = std::make_shared<o3d_slam::Mapper>(s->odometry->getBuffer(), s->submaps);
  s->mapper->setParameters(mp);
  return s;
}
void ref_slam_free(void* h) { delete static_cast<RefSlam*>(h); }

// one scan through both workers; poses out column-major; returns 1 if both accepted the scan, 0 if the odometry refused it, -1 if the mapper did
int ref_slam_add_scan(void* h, const double* pts, size_t n, double t_seconds, double odom_to_sensor[16], double map_to_sensor[16], size_t* map_points,
                      size_t* n_submaps) {
  auto* s = static_cast<RefSlam*>(h);
  const PointCloud cloud = make_cloud(pts, nullptr, nullptr, n);
  const o3d_slam::Time t = o3d_slam::fromUniversal(0) + o3d_slam::fromSeconds(1000.0 + t_seconds);
  int rc = 1;
  if (!s->odometry->addRangeScan(cloud, t))
    rc = 0;
  else if (!s->mapper->addRangeMeasurement(cloud, t))
    rc = -1;
  const Eigen::Matrix4d O = s->odometry->getOdomToRangeSensor(t).matrix(), M = s->mapper->getMapToRangeSensor(t).matrix();
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) odom_to_sensor[c * 4 + r] = O(r, c), map_to_sensor[c * 4 + r] = M(r, c);
  *map_points = s->mapper->getActiveSubmap().getMapPointCloud().points_.size();
  *n_submaps = s->submaps->getNumSubmaps();
  return rc;
}
// the active submap's cloud (points, normals); returns its size (call with null outputs first)
size_t ref_slam_map(void* h, double* out_pts, double* out_nrm) {
  auto* s = static_cast<RefSlam*>(h);
  const PointCloud& m = s->mapper->getActiveSubmap().getMapPointCloud();
  store(m.points_, out_pts);
  if (m.HasNormals()) store(m.normals_, out_nrm);
  return m.points_.size();
}
// the scan the mapper matched last (ScanToMapIcp::processForScanMatchingAndMerging's match_) -- for checking the pre-processing chain
size_t ref_slam_preprocessed_scan(void* h, double* out_pts, double* out_nrm) {
  auto* s = static_cast<RefSlam*>(h);
  const PointCloud& m = s->mapper->getPreprocessedScan();
  store(m.points_, out_pts);
  if (m.HasNormals()) store(m.normals_, out_nrm);
  return m.points_.size();
}

// A whole stream through the two workers, timed inside (the scans are converted to PointClouds of doubles first, untimed: that is the
// ROS callback's job in the reference).  threads = 0: one worker after the other per scan; 1: odometry and mapping on two threads as
// SlamWrapper runs them (SlamWrapper.cpp:228-229), the mapper waiting for the odometry of its scan.  poses_out: per frame 16 doubles
// mapToRangeSensor + 16 doubles odomToRangeSensor (column-major).  Returns the number of frames both workers accepted.
int ref_slam_run_stream(void* h, const float* scans, size_t n_pts, int n_frames, double dt, int threads, double* poses_out, double* ms_total,
                        size_t* map_points, double* ms_workers /* may be null: {odometry, mapping} busy time summed over frames 1.. (two threads: each worker's own clock) */) {
  auto* s = static_cast<RefSlam*>(h);
  std::vector<PointCloud> clouds((size_t)n_frames);
  for (int k = 0; k < n_frames; ++k) {
    clouds[k].points_.resize(n_pts);
    const float* p = scans + (size_t)k * n_pts * 3;
    for (size_t i = 0; i < n_pts; ++i) clouds[k].points_[i] = Eigen::Vector3d(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
  }
  auto stamp = [&](int k) { return o3d_slam::fromUniversal(0) + o3d_slam::fromSeconds(1000.0 + dt * k); };
  auto record = [&](int k) {
    const Eigen::Matrix4d M = s->mapper->getMapToRangeSensor(stamp(k)).matrix(), O = s->odometry->getOdomToRangeSensor(stamp(k)).matrix();
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 4; ++r) poses_out[(size_t)k * 32 + c * 4 + r] = M(r, c), poses_out[(size_t)k * 32 + 16 + c * 4 + r] = O(r, c);
  };
  int ok = 0;
  const char* samplePath = std::getenv("REF_SAMPLE_OUT");
  std::unique_ptr<StackSampler> sampler;
  if (samplePath && !threads) {
    sampler.reset(new StackSampler());
    sampler->start();
  }
  const auto t0 = std::chrono::steady_clock::now();
  double msOdo = 0.0, msMap = 0.0;
  if (!threads) {
    for (int k = 0; k < n_frames; ++k) {
      const auto a0 = std::chrono::steady_clock::now();
      const bool a = s->odometry->addRangeScan(clouds[k], stamp(k));
      const auto a1 = std::chrono::steady_clock::now();
      const bool b = a && s->mapper->addRangeMeasurement(clouds[k], stamp(k));
      const auto a2 = std::chrono::steady_clock::now();
      if (k > 0) {
        msOdo += std::chrono::duration<double, std::milli>(a1 - a0).count();
        msMap += std::chrono::duration<double, std::milli>(a2 - a1).count();
      }
      record(k);
      ok += (a && b) ? 1 : 0;
    }
  } else {
    std::mutex m;
    std::condition_variable cv;
    int odomDone = 0, mapDone = 0;
    // the odometry worker is at most kLead scans ahead of the mapper: SlamWrapper hands scans over through buffers of
    // odometryBufferSize_ / mappingBufferSize_ entries (Parameters.hpp:82,175: 1 each; SlamWrapper.cpp:204-205) and, reading a bag, waits while
    // they are full -- an unbounded lead is not a state the reference can be in (and it lets the faster worker run dozens of scans ahead)
    const int kLead = threads > 1 ? threads - 1 : 2;  // (threads > 1: a lead of threads - 1, for the sweep in tests/test_patched_reference_gpu.py)
    std::vector<char> odomOk((size_t)n_frames, 0);
    std::thread odometryWorker([&] {
      for (int k = 0; k < n_frames; ++k) {
        {
          std::unique_lock<std::mutex> l(m);
          cv.wait(l, [&] { return mapDone >= k - kLead; });
        }
        const auto a0 = std::chrono::steady_clock::now();
        const bool a = s->odometry->addRangeScan(clouds[k], stamp(k));
        if (k > 0) msOdo += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count();  // (busy time of this worker)
        std::lock_guard<std::mutex> l(m);
        odomOk[k] = a ? 1 : 0;
        odomDone = k + 1;
        cv.notify_all();
      }
    });
    std::thread mappingWorker([&] {
      for (int k = 0; k < n_frames; ++k) {
        bool a;
        {
          std::unique_lock<std::mutex> l(m);
          cv.wait(l, [&] { return odomDone > k; });
          a = odomOk[k] != 0;
        }
        const auto a1 = std::chrono::steady_clock::now();
        const bool b = a && s->mapper->addRangeMeasurement(clouds[k], stamp(k));
        if (k > 0) msMap += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a1).count();
        record(k);
        ok += (a && b) ? 1 : 0;
        {
          std::lock_guard<std::mutex> l(m);
          mapDone = k + 1;
        }
        cv.notify_all();
      }
    });
    odometryWorker.join();
    mappingWorker.join();
  }
  *ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (sampler) sampler->stop(samplePath, "ref_slam_run_stream, one thread");
  if (ms_workers) ms_workers[0] = msOdo, ms_workers[1] = msMap;
  *map_points = s->mapper->getActiveSubmap().getMapPointCloud().points_.size();
  return ok;
}

// Submap::insertScanDenseMap (Submap.cpp:77-92) for a sequence of raw scans (n_pts each, sensor frame) at the given poses, carving asked
// for every time as SlamWrapper's dense-map worker does (the every-N-scans gate is Submap::carve's, :126-136); then, optionally,
// Submap::transform (Submap.cpp:94-107).  Out: the dense map as toPointCloud gives it, with the voxel keys; sizes_out[k] = voxels after scan k.
size_t ref_submap_dense(const double* scans, size_t n_pts, int n_scans, const double* poses /* n_scans x 16, column-major */, double voxel, double crop_rmax,
                        int carve_every, double radius, double max_length, double truncation, const double* T_after, double* out_pts, int32_t* out_keys,
                        int32_t* out_counts, size_t* sizes_out, size_t cap) {
  o3d_slam::MapperParameters mp;
  mp.denseMapBuilder_.mapVoxelSize_ = voxel;
  mp.denseMapBuilder_.cropper_.cropperName_ = "MaxRadius";
  mp.denseMapBuilder_.cropper_.croppingMaxRadius_ = crop_rmax;
  mp.denseMapBuilder_.carving_.carveSpaceEveryNscans_ = carve_every;
  mp.denseMapBuilder_.carving_.neighborhoodRadiusDenseMap_ = radius;
  mp.denseMapBuilder_.carving_.maxRaytracingLength_ = max_length;
  mp.denseMapBuilder_.carving_.truncationDistance_ = truncation;
  o3d_slam::Submap sub(0, 0);
  sub.setParameters(mp);
  for (int k = 0; k < n_scans; ++k) {
    const PointCloud raw = make_cloud(scans + (size_t)k * n_pts * 3, nullptr, nullptr, n_pts);
    sub.insertScanDenseMap(raw, o3d_slam::Transform(matrix_from_colmajor(poses + 16 * k)), o3d_slam::fromUniversal(1000 + k), true);
    if (sizes_out) sizes_out[k] = sub.getDenseMap().size();
  }
  if (T_after) sub.transform(o3d_slam::Transform(matrix_from_colmajor(T_after)));
  const auto& map = sub.getDenseMap();
  size_t k = 0;
  for (const auto& v : map.voxels_) {
    if (v.second.numAggregatedPoints_ <= 0) continue;
    if (k < cap) {
      const Eigen::Vector3d p = v.second.getAggregatedPosition();
      for (int a = 0; a < 3; ++a) out_pts[3 * k + a] = p(a), out_keys[3 * k + a] = v.first(a);
      out_counts[k] = v.second.numAggregatedPoints_;
    }
    ++k;
  }
  return k;
}
}  // extern "C"
