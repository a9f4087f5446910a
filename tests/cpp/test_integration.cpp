// Functional test of integration/o3ds_open3d_slam.hpp -- the header the open3d_slam patch calls into -- against stand-ins with the
// spelling and memory layout of the Open3D / Eigen types (tests/cpp/open3d_shim).
//   test_integration --no-gpu : construction, copies, cropper names
//   test_integration          : Seam 1 (registerClouds, estimateNormals), Seam 2 (registerScan), Seam 3 (insertScan, copy on write,
//                               transform, carve) on the GPU, self-checked against each other and against analytic truth; the scan
//                               chain (preprocessScan, cropScan) and scans that stay on the device between the seams (ScanOnDevice)
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>

#include "../../integration/o3ds_open3d_slam.hpp"

#define CHECK(c)                                                                \
  do {                                                                          \
    if (!(c)) {                                                                 \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
      std::exit(1);                                                             \
    }                                                                           \
  } while (0)

using o3ds::PointCloud;

// a corner of a room (three orthogonal walls through the origin) seen from inside; points in the sensor frame at `origin`
static PointCloud cornerScan(int n, unsigned seed, double ox, double oy, double oz) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<double> u(0.0, 9.0), uz(0.0, 4.0);
  PointCloud c;
  for (int i = 0; i < n; ++i) {
    double p[3] = {u(rng), u(rng), uz(rng)};
    p[i % 3] = 0.0;  // on one of the three walls
    c.points_.emplace_back(p[0] - ox, p[1] - oy, p[2] - oz);
  }
  return c;
}
static Eigen::Isometry3d pose(double x, double y, double z, double yaw) {
  Eigen::Matrix4d M;
  M(0, 0) = std::cos(yaw), M(0, 1) = -std::sin(yaw), M(1, 0) = std::sin(yaw), M(1, 1) = std::cos(yaw);
  M(0, 3) = x, M(1, 3) = y, M(2, 3) = z;
  return Eigen::Isometry3d(M);
}
static double translationError(const Eigen::Matrix4d& A, const Eigen::Matrix4d& B) {
  double s = 0.0;
  for (int r = 0; r < 3; ++r) s += (A(r, 3) - B(r, 3)) * (A(r, 3) - B(r, 3));
  return std::sqrt(s);
}

int main(int argc, char** argv) {
  const bool gpu = !(argc > 1 && std::string(argv[1]) == "--no-gpu");
  o3d_slam::ScanCroppingParameters cp;
  cp.cropperName_ = "MinMaxRadius";
  cp.croppingMinRadius_ = 0.5;
  cp.croppingMaxRadius_ = 40.0;
  const o3ds_crop crop = o3ds::makeCrop(cp, pose(1.0, 2.0, 3.0, 0.3));
  CHECK(crop.kind == O3DS_CROP_MIN_MAX_RADIUS && crop.center[0] == 1.0 && crop.center[1] == 2.0 && crop.center[2] == 3.0 && crop.rmax == 40.0);
  cp.cropperName_ = "Sphere";
  bool threw = false;
  try {
    o3ds::makeCrop(cp, pose(0, 0, 0, 0));
  } catch (const std::runtime_error&) {
    threw = true;
  }
  CHECK(threw);
  {
    o3ds::DeviceSubmap a;  // needs no device until it is used
    o3ds::DeviceSubmap b(a), c;
    c = b;
  }
  if (!gpu) {
    std::printf("no-gpu checks ok\n");
    return 0;
  }
  cp.cropperName_ = "MinMaxRadius";
  const double maxCorr = 1.0, mapVoxel = 0.1;
  const Eigen::Isometry3d I = pose(0, 0, 0, 0);
  // Seam 1b: normals of the scans (walls through the frame's axes: unit normals along an axis, oriented to the sensor)
  PointCloud s1 = cornerScan(20000, 1, 3.0, 3.0, 1.5), s2 = cornerScan(20000, 2, 3.0, 3.0, 1.5);
  o3ds::estimateNormals(&s1, 1.0, 20);
  o3ds::estimateNormals(&s2, 1.0, 20);
  CHECK(s1.normals_.size() == s1.points_.size());
  size_t axis_aligned = 0;
  for (const auto& nrm : s1.normals_) {
    const double m = std::fmax(std::fabs(nrm[0]), std::fmax(std::fabs(nrm[1]), std::fabs(nrm[2])));
    axis_aligned += m > 0.999 ? 1 : 0;
  }
  CHECK(axis_aligned > s1.points_.size() * 9 / 10);  // everything but the neighbourhoods that straddle an edge
  // Seam 3: insertion, versions, copy on write
  o3ds::DeviceSubmap map;
  CHECK(map.empty());
  const uint64_t v0 = map.version();
  const o3ds_crop everything = o3ds::makeCrop(cp, I);
  map.insertScan(s1, I, mapVoxel, everything, maxCorr);
  const size_t n1 = map.size();
  CHECK(n1 > 1000 && n1 <= s1.points_.size() && map.version() != v0);
  o3ds::DeviceSubmap copy(map);  // shares the device map ...
  copy.insertScan(s2, I, mapVoxel, everything, maxCorr);  // ... until it changes: the original keeps its points
  CHECK(map.size() == n1);
  CHECK(copy.size() > n1);
  // Seam 2 against Seam 1: the scan displaced by a known motion registers back, the same way against the device map (crop volume as a
  // predicate, kept index) and against the downloaded map as a host cloud (index built per call)
  const Eigen::Isometry3d truth = pose(0.12, -0.08, 0.03, 0.01);
  PointCloud moved = cornerScan(8000, 3, 3.0, 3.0, 1.5);
  {  // express the scan in a frame displaced by `truth`: p' = truth^-1 p
    const double c = std::cos(0.01), s = std::sin(0.01);
    for (auto& p : moved.points_) {
      const double x = p[0] - 0.12, y = p[1] + 0.08, z = p[2] - 0.03;
      p = Eigen::Vector3d(c * x + s * y, -s * x + c * y, z);
    }
  }
  open3d::pipelines::registration::ICPConvergenceCriteria crit;
  crit.max_iteration_ = 30;
  const auto r2 = map.registerScan(O3DS_ICP_POINT_TO_PLANE, moved, everything, I, maxCorr, crit);
  CHECK(r2.fitness_ > 0.95);
  CHECK(translationError(r2.transformation_, truth.matrix()) < 0.02);
  PointCloud hostMap;
  map.download(&hostMap);
  CHECK(hostMap.points_.size() == n1 && hostMap.normals_.size() == n1);
  const auto r1 = o3ds::registerClouds(O3DS_ICP_POINT_TO_PLANE, moved, hostMap, I.matrix(), maxCorr, crit);
  CHECK(translationError(r1.transformation_, r2.transformation_) < 1e-4);  // f32 storage round trip of the map in between
  const auto rg = o3ds::registerClouds(O3DS_ICP_POINT_TO_POINT, moved, hostMap, I.matrix(), maxCorr, crit);
  CHECK(rg.fitness_ > 0.9);
  // transform: every point moves, the map still registers
  map.transform(pose(1.0, 0.0, 0.0, 0.0), maxCorr);
  PointCloud shifted;
  map.download(&shifted);
  CHECK(shifted.points_.size() == n1);
  double worst = 0.0;
  for (size_t i = 0; i < n1; ++i) worst = std::fmax(worst, std::fabs(shifted.points_[i][0] - hostMap.points_[i][0] - 1.0));
  CHECK(worst < 1e-5);
  const auto r3 = map.registerScan(O3DS_ICP_POINT_TO_PLANE, moved, o3ds::makeCrop(cp, pose(1.0, 0, 0, 0)), pose(1.0, 0, 0, 0), maxCorr, crit);
  CHECK(std::fabs(r3.transformation_(0, 3) - (truth.matrix()(0, 3) + 1.0)) < 0.02);
  // carve (Submap::carve -> getIdxsOfCarvedPoints): points go, the map shrinks by exactly that many and is not wiped out
  o3d_slam::SpaceCarvingParameters sc;
  const size_t before = copy.size();
  const uint64_t vc = copy.version();
  PointCloud mapBefore, toRemove, scanRef, mapAfter;
  copy.download(&mapBefore);
  const size_t removed = copy.carve(s1, I, everything, sc, &toRemove, &scanRef);  // Submap::toRemove_ / scanRef_ (Submap.cpp:119-120)
  CHECK(copy.size() == before - removed && copy.size() > before / 4 && copy.version() != vc);
  copy.download(&mapAfter);
  CHECK(toRemove.points_.size() == removed && removed > 0 && toRemove.normals_.size() == removed);
  CHECK(scanRef.points_.size() == s1.points_.size());
  {  // the carved points and the survivors, interleaved back in map order, are the map as it was (both keep their order)
    size_t a = 0, b = 0;
    for (size_t i = 0; i < mapBefore.points_.size(); ++i) {
      const bool isKept = a < mapAfter.points_.size() && mapAfter.points_[a][0] == mapBefore.points_[i][0] && mapAfter.points_[a][1] == mapBefore.points_[i][1] &&
                          mapAfter.points_[a][2] == mapBefore.points_[i][2];
      const bool isGone = b < removed && toRemove.points_[b][0] == mapBefore.points_[i][0] && toRemove.points_[b][1] == mapBefore.points_[i][1] &&
                          toRemove.points_[b][2] == mapBefore.points_[i][2];
      CHECK(isKept || isGone);
      if (isKept)
        ++a;
      else
        ++b;
    }
    CHECK(a == mapAfter.points_.size() && b == removed);
    for (size_t i = 0; i < scanRef.points_.size(); i += 997)  // identity pose: the placed scan is the scan (f32 storage)
      for (int k = 0; k < 3; ++k) CHECK(std::fabs(scanRef.points_[i][k] - s1.points_[i][k]) < 1e-5);
  }
  // ---- Seam 2, first half: the scan chain, and scans that stay on the device between the seams -------------------------------------
  {
    PointCloud raw = cornerScan(60000, 7, 3.0, 3.0, 1.5);
    o3ds::ScanChain chain;
    chain.crop = o3ds::makeCrop(cp, I);
    chain.voxelSize = 0.1;
    chain.estimateNormals = true;
    chain.normalRadius = 1.0;
    chain.normalKnn = 20;
    o3ds::setScanStamp(41);  // the scan's Time stamp, as the patched seams pass it on: what the pre-processing memo tells scans apart by
    std::shared_ptr<PointCloud> pre = o3ds::preprocessScan(raw, chain);
    CHECK(pre->points_.size() > 1000 && pre->points_.size() < raw.points_.size() && pre->HasNormals());
    CHECK(dynamic_cast<o3ds::ScanOnDevice*>(pre.get()) != nullptr);
    CHECK(o3ds::deviceCopyOf(*pre) != nullptr);
    // the second caller of the same chain on (a copy of) the same raw scan -- the mapper after the odometry -- gets the first caller's
    // cloud, also from another thread; another parameter, another scan or random down-sampling compute their own
    {
      const PointCloud rawCopy = raw;
      CHECK(o3ds::preprocessScan(rawCopy, chain).get() == pre.get());
      std::shared_ptr<PointCloud> fromThread;
      std::thread([&] {
        const o3ds::ScanStampScope sameScan(41);  // (the stamp is the calling thread's)
        fromThread = o3ds::preprocessScan(rawCopy, chain);
      }).join();
      CHECK(fromThread.get() == pre.get());
      {  // a caller that does not say which scan it has shares nothing
        const o3ds::ScanStampScope unknown(0);
        CHECK(o3ds::preprocessScan(rawCopy, chain).get() != pre.get());
      }
      {  // RandomDownSample: the chain is shared up to the draw (a memo entry of its own scan), every caller draws for itself
        const o3ds::ScanStampScope drawn(77);
        o3ds::ScanChain third = chain;
        third.downSamplingRatio = 0.3;
        std::shared_ptr<PointCloud> a = o3ds::preprocessScan(rawCopy, third), b = o3ds::preprocessScan(rawCopy, third), c;
        std::thread([&] {
          const o3ds::ScanStampScope sameScan(77);
          c = o3ds::preprocessScan(rawCopy, third);
        }).join();
        const size_t want = (size_t)(int)(0.3 * (double)pre->points_.size());
        CHECK(a->points_.size() == want && b->points_.size() == want && c->points_.size() == want && a.get() != b.get());
        size_t differ = 0;
        for (size_t i = 0; i < want; ++i) differ += (a->points_[i][0] != b->points_[i][0] || a->points_[i][1] != b->points_[i][1] || a->points_[i][2] != b->points_[i][2]) ? 1 : 0;
        CHECK(differ > want / 2);  // two draws
        // every drawn point (and its normal) is a point of the undrawn cloud
        std::vector<std::array<double, 6>> all(pre->points_.size());
        for (size_t i = 0; i < all.size(); ++i)
          all[i] = {pre->points_[i][0], pre->points_[i][1], pre->points_[i][2], pre->normals_[i][0], pre->normals_[i][1], pre->normals_[i][2]};
        std::sort(all.begin(), all.end());
        for (const std::shared_ptr<PointCloud>& d : {a, b, c})
          for (size_t i = 0; i < want; i += 7) {
            const std::array<double, 6> q = {d->points_[i][0], d->points_[i][1], d->points_[i][2], d->normals_[i][0], d->normals_[i][1], d->normals_[i][2]};
            CHECK(std::binary_search(all.begin(), all.end(), q));
          }
      }
      o3ds::ScanChain other = chain;
      other.voxelSize = 0.11;
      std::shared_ptr<PointCloud> coarser = o3ds::preprocessScan(rawCopy, other);
      CHECK(coarser.get() != pre.get() && coarser->points_.size() < pre->points_.size());
      PointCloud moved = raw;
      for (auto& q : moved.points_) q[0] += 0.5;
      o3ds::setScanStamp(42);  // another scan
      CHECK(o3ds::preprocessScan(moved, other).get() != coarser.get());
      o3ds::setScanStamp(41);
      // the memo holds the last few scans: scan 41 is still there after scan 42 (the mapper is a scan or two behind the odometry) ...
      std::shared_ptr<PointCloud> again = o3ds::preprocessScan(rawCopy, chain);
      CHECK(again.get() == pre.get());
      // ... and gone after a handful of others: computed again, same values
      for (int k = 0; k < 9; ++k) {
        o3ds::setScanStamp(100 + k);
        PointCloud shifted = raw;
        for (auto& q : shifted.points_) q[1] += 0.01 * (k + 1);
        CHECK(o3ds::preprocessScan(shifted, chain) != nullptr);
      }
      o3ds::setScanStamp(41);
      std::shared_ptr<PointCloud> recomputed = o3ds::preprocessScan(rawCopy, chain);
      CHECK(recomputed.get() != pre.get() && recomputed->points_.size() == pre->points_.size());
      for (size_t i = 0; i < pre->points_.size(); ++i)
        for (int a = 0; a < 3; ++a) CHECK(recomputed->points_[i][a] == pre->points_[i][a] && recomputed->normals_[i][a] == pre->normals_[i][a]);
    }
    // colours cross the seams with the points (the reference's crop / VoxelDownSample keep them on the scan that reaches mapCloud_)
    {
      PointCloud tinted = raw;
      tinted.colors_.assign(tinted.points_.size(), Eigen::Vector3d(0.25, 0.5, 0.75));
      o3ds::setScanStamp(43);  // (another scan: same points, with colours)
      std::shared_ptr<PointCloud> pt = o3ds::preprocessScan(tinted, chain);
      CHECK(pt.get() != pre.get() && pt->points_.size() == pre->points_.size() && pt->colors_.size() == pt->points_.size());
      for (size_t i = 0; i < pt->colors_.size(); i += 97)
        CHECK(std::fabs(pt->colors_[i][0] - 0.25) < 1e-6 && std::fabs(pt->colors_[i][1] - 0.5) < 1e-6 && std::fabs(pt->colors_[i][2] - 0.75) < 1e-6);
      o3d_slam::ScanCroppingParameters np2 = cp;
      np2.croppingMinRadius_ = 1.0;
      np2.croppingMaxRadius_ = 6.0;
      auto cropped = o3ds::cropScan(*pt, o3ds::makeCrop(np2, I));
      CHECK(cropped->points_.size() > 100 && cropped->colors_.size() == cropped->points_.size());
      // an edit of the host arrays that the 64 samples would miss: the caller says so
      CHECK(o3ds::deviceCopyOf(*pt) != nullptr);
      o3ds::invalidateDeviceCopy(*pt);
      CHECK(o3ds::deviceCopyOf(*pt) == nullptr);
    }
    // the same chain seam by seam on host clouds (what round 2's patch left to the CPU, here through the stateless calls)
    const PointCloud plain = *pre;  // sliced: an ordinary host cloud, no device copy
    CHECK(o3ds::deviceCopyOf(plain) == nullptr);
    // narrow crop: on the device copy and on the host copy -- same points, bit for bit
    o3d_slam::ScanCroppingParameters narrowP = cp;
    narrowP.croppingMinRadius_ = 1.0;
    narrowP.croppingMaxRadius_ = 6.0;
    const o3ds_crop narrowCrop = o3ds::makeCrop(narrowP, I);
    auto n1c = o3ds::cropScan(*pre, narrowCrop), n2c = o3ds::cropScan(plain, narrowCrop);
    CHECK(n1c->points_.size() == n2c->points_.size() && n1c->points_.size() > 100 && n1c->points_.size() < pre->points_.size());
    for (size_t i = 0; i < n1c->points_.size(); ++i)
      for (int a = 0; a < 3; ++a) CHECK(n1c->points_[i][a] == n2c->points_[i][a] && n1c->normals_[i][a] == n2c->normals_[i][a]);
    CHECK(o3ds::cropContains(everything, narrowCrop) && !o3ds::cropContains(narrowCrop, everything) && o3ds::cropContains(narrowCrop, narrowCrop));
    CHECK(o3ds::cropContains(o3ds::noCrop(), everything) && !o3ds::cropContains(everything, o3ds::noCrop()));
    // registration of resident clouds == registration of their host copies (LidarOdometry::addRangeScan as patched)
    PointCloud raw2 = cornerScan(60000, 8, 3.05, 2.95, 1.52);
    std::shared_ptr<PointCloud> pre2 = o3ds::preprocessScan(raw2, chain);
    const PointCloud plain2 = *pre2;
    const auto ra = o3ds::registerClouds(O3DS_ICP_POINT_TO_PLANE, *pre, *pre2, I.matrix(), maxCorr, crit);
    const auto rb = o3ds::registerClouds(O3DS_ICP_POINT_TO_PLANE, plain, plain2, I.matrix(), maxCorr, crit);
    CHECK(ra.fitness_ > 0.9);
    CHECK(translationError(ra.transformation_, rb.transformation_) < 1e-9);
    CHECK(std::fabs(ra.transformation_(0, 3) + 0.05) < 0.02 && std::fabs(ra.transformation_(1, 3) - 0.05) < 0.02);  // p_target = p_source + (o1 - o2)
    // work queued behind a registration's launches (o3ds::overlapNext): the pre-processing of a third scan, made while the host waits --
    // the registration and the scan are what they are without it; the callback runs once and is not inherited
    {
      PointCloud raw3 = cornerScan(60000, 9, 3.1, 2.9, 1.5);
      const std::shared_ptr<PointCloud> want3 = o3ds::preprocessScan(raw3, chain);
      std::shared_ptr<PointCloud> got3;
      int calls = 0;
      o3ds::overlapNext([&] {
        ++calls;
        got3 = o3ds::preprocessScan(raw3, chain);
      });
      const auto rh = o3ds::registerClouds(O3DS_ICP_POINT_TO_PLANE, *pre, *pre2, I.matrix(), maxCorr, crit);
      CHECK(calls == 1 && got3 && got3->points_.size() == want3->points_.size());
      for (size_t i = 0; i < want3->points_.size(); i += 53)
        for (int a = 0; a < 3; ++a) CHECK(got3->points_[i][a] == want3->points_[i][a] && got3->normals_[i][a] == want3->normals_[i][a]);
      const auto rn = o3ds::registerClouds(O3DS_ICP_POINT_TO_PLANE, *pre, *pre2, I.matrix(), maxCorr, crit);
      CHECK(calls == 1 && rh.fitness_ == ra.fitness_);
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) CHECK(rh.transformation_(r, c) == ra.transformation_(r, c) && rn.transformation_(r, c) == ra.transformation_(r, c));
    }
    // a caller that edits the host arrays invalidates the device copy: the edited values are what gets used
    o3ds::ScanOnDevice edited = *dynamic_cast<o3ds::ScanOnDevice*>(pre.get());
    CHECK(o3ds::deviceCopyOf(edited) != nullptr);  // a copy of the object shares the device copy
    for (auto& q : edited.points_) q[0] += 0.25;
    CHECK(o3ds::deviceCopyOf(edited) == nullptr);
    const auto rc = o3ds::registerClouds(O3DS_ICP_POINT_TO_PLANE, edited, *pre2, I.matrix(), maxCorr, crit);
    CHECK(std::fabs(rc.transformation_(0, 3) - ra.transformation_(0, 3) + 0.25) < 0.02);
    // Seam 3 / Seam 2 second half with a resident scan (Mapper::addRangeMeasurement as patched): the map handle takes the scan
    // device to device; same map and same registration as with the host copy
    o3ds::DeviceSubmap mA, mB;
    mA.insertScan(*pre, I, mapVoxel, everything, maxCorr);
    mB.insertScan(plain, I, mapVoxel, everything, maxCorr);
    PointCloud hA, hB;
    mA.download(&hA);
    mB.download(&hB);
    CHECK(hA.points_.size() == hB.points_.size() && hA.points_.size() > 1000);
    for (size_t i = 0; i < hA.points_.size(); ++i)
      for (int a = 0; a < 3; ++a) CHECK(hA.points_[i][a] == hB.points_[i][a] && hA.normals_[i][a] == hB.normals_[i][a]);
    const auto sA = mA.registerScan(O3DS_ICP_POINT_TO_PLANE, *pre2, everything, I, maxCorr, crit);
    const auto sB = mB.registerScan(O3DS_ICP_POINT_TO_PLANE, plain2, everything, I, maxCorr, crit);
    CHECK(sA.fitness_ > 0.9 && translationError(sA.transformation_, sB.transformation_) < 1e-9);
    // down-sampling inside the chain: int(ratio * n) points, every one of them a point of the full result, pinned by the seed
    chain.downSamplingRatio = 0.5;
    o3ds::setRandomDownSampleSeed(42);
    auto half = o3ds::preprocessScan(raw, chain);
    o3ds::setRandomDownSampleSeed(42);
    auto half2 = o3ds::preprocessScan(raw, chain);
    CHECK(half->points_.size() == (size_t)(int)(0.5 * (double)pre->points_.size()));
    CHECK(half2->points_.size() == half->points_.size());
    for (size_t i = 0; i < half->points_.size(); ++i)
      for (int a = 0; a < 3; ++a) CHECK(half->points_[i][a] == half2->points_[i][a]);
    // point-to-point chains skip the normals; the base CroppingVolume keeps everything
    chain.downSamplingRatio = 1.0;
    chain.estimateNormals = false;
    chain.crop = o3ds::noCrop();
    auto bare = o3ds::preprocessScan(raw, chain);
    CHECK(!bare->HasNormals() && bare->points_.size() >= pre->points_.size());
    // the device copy dies with the last cloud that refers to it, also on another thread
    std::shared_ptr<PointCloud> keep = o3ds::preprocessScan(raw2, chain);
    std::thread([moved = std::move(keep)]() mutable { moved.reset(); }).join();
  }
  std::printf("gpu checks ok\n");
  return 0;
}
