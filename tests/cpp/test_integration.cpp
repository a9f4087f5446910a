// Functional test of integration/o3ds_open3d_slam.hpp -- the header the open3d_slam patch calls into -- against stand-ins with the
// spelling and memory layout of the Open3D / Eigen types (tests/cpp/open3d_shim).
//   test_integration --no-gpu : construction, copies, cropper names
//   test_integration          : Seam 1 (registerClouds, estimateNormals), Seam 2 (registerScan), Seam 3 (insertScan, copy on write,
//                               transform, carve) on the GPU, self-checked against each other and against analytic truth
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

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
  const size_t removed = copy.carve(s1, I, everything, sc);
  CHECK(copy.size() == before - removed && copy.size() > before / 4 && copy.version() != vc);
  std::printf("gpu checks ok\n");
  return 0;
}
