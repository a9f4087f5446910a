// BASELINE configs[2] through integration/o3ds_open3d_slam.hpp -- the functions the open3d_slam patch calls -- with HOST clouds at
// every seam, in the order the patched reference calls them:
//   LidarOdometry::addRangeScan (Odometry.cpp:32-79):      preprocess -> registerClouds(prev, current) -> integrate the inverse
//   Mapper::addRangeMeasurement (Mapper.cpp:101-181):      processForScanMatchingAndMerging (preprocess + narrow crop) ->
//                                                          scanToMapRegistration (odometry prior) -> Submap::insertScan
// Every raw scan arrives as a host PointCloud of doubles (what rosToOpen3d hands over), every result a caller can read is downloaded;
// what is NOT repeated is the upload of a scan the previous seam already put on the device (o3ds::ScanOnDevice).
//   stream_integration <scans.bin> <serial|threads> [poses.bin]
//     scans.bin : int32 frames, int32 points, then per frame 16 doubles (true pose, column-major) and points x 3 float32 (sensor frame)
//     threads   : odometry and mapping on two worker threads with a pose buffer in between, as SlamWrapper runs them (SlamWrapper.cpp:228-229)
//     poses.bin : per frame 16 doubles mapToRangeSensor + 16 doubles odomToRangeSensor (for the parity test)
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../integration/o3ds_open3d_slam.hpp"

using o3ds::PointCloud;
using Clock = std::chrono::steady_clock;
using M4 = Eigen::Matrix4d;

static double ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
static M4 mul(const M4& A, const M4& B) {
  M4 R;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += A(r, k) * B(k, c);
      R(r, c) = s;
    }
  return R;
}
static M4 inverseRigid(const M4& T) {  // [R t]^-1 = [R^T  -R^T t]
  M4 R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R(i, j) = T(j, i);
  for (int i = 0; i < 3; ++i) R(i, 3) = -(R(i, 0) * T(0, 3) + R(i, 1) * T(1, 3) + R(i, 2) * T(2, 3));
  R(3, 0) = R(3, 1) = R(3, 2) = 0.0;
  R(3, 3) = 1.0;
  return R;
}

struct Setup {
  o3d_slam::ScanCroppingParameters cropper;  // the shipped Lua defaults (parameter_structure_definitions.lua:49-72,94-118), point-to-plane
  double voxel = 0.1, mapVoxel = 0.1, maxCorr = 1.0, normalRadius = 3.0, minRefinementFitness = 0.7;
  int knn = 20, maxIter = 50;
  Setup() {
    cropper.cropperName_ = "MinMaxRadius";
    cropper.croppingMinRadius_ = 2.0;
    cropper.croppingMaxRadius_ = 30.0;
  }
  o3ds::ScanChain chain() const {
    o3ds::ScanChain c;
    c.crop = o3ds::makeCrop(cropper, Eigen::Isometry3d::Identity());
    c.voxelSize = voxel;
    c.estimateNormals = true;
    c.normalRadius = normalRadius;
    c.normalKnn = knn;
    c.downSamplingRatio = 1.0;
    return c;
  }
  open3d::pipelines::registration::ICPConvergenceCriteria criteria() const {
    open3d::pipelines::registration::ICPConvergenceCriteria c;
    c.max_iteration_ = maxIter;
    return c;
  }
};

// LidarOdometry as patched
struct Odometry {
  Setup s;
  std::shared_ptr<PointCloud> prev;
  M4 cumulative = M4::Identity();
  double tPre = 0, tReg = 0;
  bool add(const PointCloud& raw) {
    const auto t0 = Clock::now();
    auto pre = o3ds::preprocessScan(raw, s.chain());
    const auto t1 = Clock::now();
    tPre += ms(t0, t1);
    if (!prev) {
      prev = pre;
      return true;
    }
    const auto r = o3ds::registerClouds(O3DS_ICP_POINT_TO_PLANE, *prev, *pre, M4::Identity(), s.maxCorr, s.criteria());
    tReg += ms(t1, Clock::now());
    if (!(r.fitness_ > 0.1)) return false;
    cumulative = mul(cumulative, inverseRigid(r.transformation_));
    prev = pre;
    return true;
  }
};

// Mapper::addRangeMeasurement as patched (one submap)
struct Mapping {
  Setup s;
  o3ds::DeviceSubmap map;
  M4 T = M4::Identity(), Tprev = M4::Identity(), odomPrev = M4::Identity();
  bool first = true;
  double tPre = 0, tReg = 0, tIns = 0, tCarve = 0, minFitness = 1.0;
  int nScansInsertedMap = 0;            // Submap::nScansInsertedMap_
  size_t nCarved = 0;
  M4 Tinserted = M4::Identity();        // pose the map builder's cropper still holds when Submap::carve runs (Submap.cpp:56-71)
  o3d_slam::SpaceCarvingParameters carving;  // Parameters.hpp:85-92 defaults: voxel 0.1, 20 m rays, truncation 0.1, every 10 scans
  // Submap::insertScan as patched (Submap.cpp:54-72): SubmapCollection::insertScan always asks for carving; Submap::carve acts when the
  // map is not empty and nScansInsertedMap_ % carveSpaceEveryNscans_ == 1, with the RAW scan
  void insert(const PointCloud& raw, const PointCloud& wide, const M4& pose) {
    if (map.size() > 0 && nScansInsertedMap % carving.carveSpaceEveryNscans_ == 1) {
      const auto t0 = Clock::now();
      nCarved += map.carve(raw, Eigen::Isometry3d(pose), o3ds::makeCrop(s.cropper, Eigen::Isometry3d(Tinserted)), carving);
      tCarve += ms(t0, Clock::now());
    }
    map.insertScan(wide, Eigen::Isometry3d(pose), s.mapVoxel, o3ds::makeCrop(s.cropper, Eigen::Isometry3d(pose)), s.maxCorr);
    ++nScansInsertedMap;
    Tinserted = pose;
  }
  bool add(const PointCloud& raw, const M4& odomNow) {
    const auto t0 = Clock::now();
    auto wide = o3ds::preprocessScan(raw, s.chain());
    const o3ds_crop narrowCrop = o3ds::makeCrop(s.cropper, Eigen::Isometry3d::Identity()), wideCrop = s.chain().crop;  // scan-matcher / map-builder volume
    auto narrow = o3ds::cropContains(narrowCrop, wideCrop) ? wide : o3ds::cropScan(*wide, narrowCrop);  // as the patched processForScanMatchingAndMerging
    if (narrow->points_.empty() || wide->points_.empty()) return false;
    const auto t1 = Clock::now();
    tPre += ms(t0, t1);
    if (first) {
      insert(raw, *wide, M4::Identity());
      tIns += ms(t1, Clock::now());
      first = false;
      odomPrev = odomNow;
      return true;
    }
    const M4 estimate = mul(Tprev, mul(inverseRigid(odomPrev), odomNow));
    const auto r = map.registerScan(O3DS_ICP_POINT_TO_PLANE, *narrow, o3ds::makeCrop(s.cropper, Eigen::Isometry3d(T)), Eigen::Isometry3d(estimate), s.maxCorr,
                                    s.criteria());
    const auto t2 = Clock::now();
    tReg += ms(t1, t2);
    if (r.fitness_ < s.minRefinementFitness) return false;
    minFitness = std::min(minFitness, r.fitness_);
    T = r.transformation_;
    insert(raw, *wide, T);
    tIns += ms(t2, Clock::now());
    Tprev = T;
    odomPrev = odomNow;
    return true;
  }
};

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: stream_integration scans.bin serial|threads [poses.bin]\n");
    return 2;
  }
  const bool threads = std::string(argv[2]) == "threads";
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int frames = 0, npts = 0;
  if (std::fread(&frames, 4, 1, f) != 1 || std::fread(&npts, 4, 1, f) != 1) return 2;
  std::vector<M4> truth(frames);
  std::vector<PointCloud> scans(frames);
  std::vector<float> buf((size_t)npts * 3);
  for (int k = 0; k < frames; ++k) {
    if (std::fread(truth[k].data(), sizeof(double), 16, f) != 16) return 2;
    if (std::fread(buf.data(), sizeof(float) * 3, npts, f) != (size_t)npts) return 2;
    scans[k].points_.resize(npts);
    for (int i = 0; i < npts; ++i) scans[k].points_[i] = Eigen::Vector3d(buf[3 * (size_t)i], buf[3 * (size_t)i + 1], buf[3 * (size_t)i + 2]);
  }
  std::fclose(f);

  Odometry odo;
  Mapping mapping;
  std::vector<M4> odomAt(frames), mapAt(frames);
  bool ok = true;
  {  // warm both paths (handle creation, first allocations) outside the timed region, on throw-away objects
    Odometry o2;
    Mapping m2;
    o2.add(scans[0]);
    m2.add(scans[0], M4::Identity());
  }
  const auto t0 = Clock::now();
  if (!threads) {
    for (int k = 0; k < frames && ok; ++k) {
      const o3ds::ScanStampScope stamp(1000 + k);  // the scan's Time stamp, as the patched addRangeScan / addRangeMeasurement pass it on
      ok = odo.add(scans[k]);
      odomAt[k] = odo.cumulative;
      ok = ok && mapping.add(scans[k], odomAt[k]);
      mapAt[k] = mapping.T;
    }
  } else {
    std::mutex m;
    std::condition_variable cv;
    int odomDone = 0, mapDone = 0;
    bool odoOk = true;
    // neither worker is more than kLead scans ahead of the other: the reference's workers hand scans over through buffers of
    // odometryBufferSize_ / mappingBufferSize_ entries (Parameters.hpp:82,175: 1 each; SlamWrapper.cpp:204-205)
    constexpr int kLead = 2;
    std::thread odometryWorker([&] {
      for (int k = 0; k < frames; ++k) {
        {
          std::unique_lock<std::mutex> l(m);
          cv.wait(l, [&] { return mapDone >= k - kLead || !ok; });
        }
        const o3ds::ScanStampScope stamp(1000 + k);
        const bool good = odo.add(scans[k]);
        std::lock_guard<std::mutex> l(m);
        odomAt[k] = odo.cumulative;
        odomDone = k + 1;
        odoOk = odoOk && good;
        cv.notify_all();
      }
    });
    std::thread mappingWorker([&] {
      for (int k = 0; k < frames && ok; ++k) {
        M4 odom;
        {
          std::unique_lock<std::mutex> l(m);
          cv.wait(l, [&] { return odomDone > k; });
          odom = odomAt[k];
        }
        const o3ds::ScanStampScope stamp(1000 + k);
        const bool good = mapping.add(scans[k], odom);
        mapAt[k] = mapping.T;
        {
          std::lock_guard<std::mutex> l(m);
          ok = good;
          mapDone = k + 1;
        }
        cv.notify_all();
      }
    });
    odometryWorker.join();
    mappingWorker.join();
    ok = ok && odoOk;
  }
  const double total = ms(t0, Clock::now());
  if (!ok) {
    std::fprintf(stderr, "a frame failed its fitness gate\n");
    return 1;
  }
  if (argc > 3) {
    std::FILE* g = std::fopen(argv[3], "wb");
    if (!g) return 2;
    for (int k = 0; k < frames; ++k) {
      std::fwrite(mapAt[k].data(), sizeof(double), 16, g);
      std::fwrite(odomAt[k].data(), sizeof(double), 16, g);
    }
    std::fclose(g);
  }
  const M4 rel = mul(inverseRigid(truth[0]), truth[frames - 1]);
  const double dx = rel(0, 3) - mapping.T(0, 3), dy = rel(1, 3) - mapping.T(1, 3), dz = rel(2, 3) - mapping.T(2, 3);
  const double dt = std::sqrt(dx * dx + dy * dy + dz * dz);
  std::printf(
      "{\"workload\": \"integration header (what the open3d_slam patch calls), host clouds at every seam, %d raw pts/scan, %d frames, %s\", "
      "\"scans_per_sec\": %.1f, \"scans_per_sec_mapping_only\": %.1f, \"ms_per_scan\": {\"odometry_preprocess\": %.3f, \"odometry_registration\": %.3f, "
      "\"mapping_preprocess\": %.3f, \"mapping_registration\": %.3f, \"insert\": %.3f, \"carve_within_insert\": %.3f}, \"map_points\": %zu, \"carved_points\": %zu, \"min_fitness\": %.4f, "
      "\"final_translation_error_m\": %.5f}\n",
      npts, frames, threads ? "odometry and mapping on two worker threads" : "one thread", 1e3 * frames / total,
      1e3 * frames / (mapping.tPre + mapping.tReg + mapping.tIns), odo.tPre / frames, odo.tReg / frames, mapping.tPre / frames, mapping.tReg / frames,
      mapping.tIns / frames, mapping.tCarve / frames, mapping.map.size(), mapping.nCarved, mapping.minFitness, dt);
  return dt < 0.05 ? 0 : 1;
}
