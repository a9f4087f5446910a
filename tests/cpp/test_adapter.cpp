// Host-adapter test: drives the reference-named C++ classes of open3d_slam_amd/host/o3ds_adapter.hpp.
//   test_adapter --no-gpu : checks that need no device (factories, parameter copy, error behaviour)
//   test_adapter          : registration / croppers / voxelize / submap on the GPU, self-checked against analytic truth
// (parity against the CPU oracle is done by the Python tests through the same C-ABI).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../open3d_slam_amd/host/o3ds_adapter.hpp"

using namespace o3d_slam;

#define CHECK(c)                                                   \
  do {                                                             \
    if (!(c)) {                                                    \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
      std::exit(1);                                                \
    }                                                              \
  } while (0)

template <class F>
static bool throws(F f) {
  try {
    f();
  } catch (const std::runtime_error&) {
    return true;
  } catch (const std::out_of_range&) {
    return true;
  }
  return false;
}

// three orthogonal planes with analytic normals pointing to the +octant where the sensor sits
static PointCloud threePlanes(int n, unsigned seed) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<double> u(0.3, 6.0);
  PointCloud c;
  for (int i = 0; i < n; ++i) {
    const int k = i % 3;
    std::array<double, 3> p{{u(rng), u(rng), u(rng)}}, nn{{0, 0, 0}};
    p[k] = 0.0;
    nn[k] = 1.0;
    c.points_.push_back(p);
    c.normals_.push_back(nn);
  }
  return c;
}

static Transform smallPose(double tx, double ty, double tz, double yaw) {
  Transform T;
  T.m = {{std::cos(yaw), std::sin(yaw), 0, 0, -std::sin(yaw), std::cos(yaw), 0, 0, 0, 0, 1, 0, tx, ty, tz, 1}};
  return T;
}

static void noGpuChecks() {
  CloudRegistrationParameters p;
  p.icp_.maxNumIter_ = 17;
  p.icp_.maxCorrespondenceDistance_ = 0.7;
  p.icp_.knn_ = 9;
  p.icp_.maxDistanceKnn_ = 1.5;
  auto reg = cloudRegistrationFactory(p);
  auto* p2p = dynamic_cast<RegistrationIcpPointToPlane*>(reg.get());
  CHECK(p2p != nullptr);
  CHECK(p2p->maxCorrespondenceDistance_ == 0.7 && p2p->knnNormalEstimation_ == 9 && p2p->maxRadiusNormalEstimation_ == 1.5);
  CHECK(p2p->icpConvergenceCriteria_.max_iteration_ == 17);
  CHECK(p2p->icpConvergenceCriteria_.relative_fitness_ == 1e-6 && p2p->icpConvergenceCriteria_.relative_rmse_ == 1e-6);
  p.regType_ = CloudRegistrationType::GeneralizedIcp;
  auto g = cloudRegistrationFactory(p);
  CHECK(dynamic_cast<RegistrationIcpGeneralized*>(g.get()) != nullptr);
  CHECK(dynamic_cast<RegistrationIcpGeneralized*>(g.get())->icpConvergenceCriteria_.max_iteration_ == 17);
  p.regType_ = CloudRegistrationType::PointToPointIcp;
  auto pp = cloudRegistrationFactory(p);
  CHECK(dynamic_cast<RegistrationIcpPointToPoint*>(pp.get()) != nullptr);
  CHECK(dynamic_cast<RegistrationIcpPointToPoint*>(pp.get())->maxCorrespondenceDistance_ == 0.7);
  CHECK(dynamic_cast<RegistrationIcpPointToPoint*>(pp.get())->icpConvergenceCriteria_.max_iteration_ == 17);
  p.regType_ = static_cast<CloudRegistrationType>(42);
  CHECK(throws([&] { cloudRegistrationFactory(p); }));
  ScanCroppingParameters cp;
  cp.cropperName_ = "MinMaxRadius";
  cp.croppingMinRadius_ = 2.0;
  cp.croppingMaxRadius_ = 30.0;
  auto cr = croppingVolumeFactory(cp);
  cr->setPose(smallPose(1, 2, 3, 0.3));
  const o3ds_crop c = cr->toAbi();
  CHECK(c.kind == O3DS_CROP_MIN_MAX_RADIUS && c.rmin == 2.0 && c.rmax == 30.0 && c.center[0] == 1 && c.center[1] == 2 && c.center[2] == 3);
  cp.cropperName_ = "NoSuchCropper";
  CHECK(throws([&] { croppingVolumeFactory(cp); }));
  RegistrationIcpPointToPlane bad;
  bad.maxRadiusNormalEstimation_ = 0.0;
  PointCloud pc = threePlanes(30, 1);
  CHECK(throws([&] { bad.estimateNormalsOrCovariancesIfNeeded(&pc); }));
  std::puts("no-gpu checks ok");
}

static void gpuChecks() {
  // registerClouds: recover a known small pose on exact planes
  const PointCloud target = threePlanes(30000, 1);
  PointCloud srcMap = threePlanes(3000, 2);
  // source = T^-1 * srcMap with T = smallPose(0.03, -0.02, 0.015, 0.01), so that registering source->target returns T
  PointCloud source;
  const double c = std::cos(0.01), s = std::sin(0.01);
  for (auto& p : srcMap.points_) {
    const double x = p[0] - 0.03, y = p[1] + 0.02, z = p[2] - 0.015;
    source.points_.push_back({{c * x + s * y, -s * x + c * y, z}});
  }
  CloudRegistrationParameters prm;
  prm.icp_.maxCorrespondenceDistance_ = 0.5;
  prm.icp_.maxNumIter_ = 40;
  auto reg = cloudRegistrationFactory(prm);
  const RegistrationResult r = reg->registerClouds(source, target, Transform::Identity());
  CHECK(r.fitness_ > 0.99);
  CHECK(std::fabs(r.transformation_[12] - 0.03) < 1e-4 && std::fabs(r.transformation_[13] + 0.02) < 1e-4 &&
        std::fabs(r.transformation_[14] - 0.015) < 1e-4);
  CHECK(std::fabs(r.transformation_[1] - s) < 1e-4);
  // error behaviour of the seam
  PointCloud noNormals = target;
  noNormals.normals_.clear();
  CHECK(throws([&] { reg->registerClouds(source, noNormals, Transform::Identity()); }));
  auto* p2p = dynamic_cast<RegistrationIcpPointToPlane*>(reg.get());
  p2p->maxCorrespondenceDistance_ = 0.0;
  CHECK(throws([&] { reg->registerClouds(source, target, Transform::Identity()); }));
  p2p->maxCorrespondenceDistance_ = 0.5;
  // generalized ICP through the same seam
  {
    CloudRegistrationParameters gp = prm;
    gp.regType_ = CloudRegistrationType::GeneralizedIcp;
    auto greg = cloudRegistrationFactory(gp);
    PointCloud srcN = source;
    {  // source without normals: [O3D] InitializePointCloudForGeneralizedICP estimates them (KNN 20) on a copy
      const RegistrationResult r0 = greg->registerClouds(srcN, target, Transform::Identity());
      CHECK(srcN.normals_.empty());
      CHECK(r0.fitness_ > 0.99 && std::fabs(r0.transformation_[12] - 0.03) < 5e-3 && std::fabs(r0.transformation_[13] + 0.02) < 5e-3);
    }
    srcN.normals_.clear();
    for (size_t i = 0; i < srcN.points_.size(); ++i) {  // normals of the plane each source point came from, rotated into the scan frame
      std::array<double, 3> nn{{0, 0, 0}};
      nn[i % 3] = 1.0;
      srcN.normals_.push_back({{c * nn[0] + s * nn[1], -s * nn[0] + c * nn[1], nn[2]}});
    }
    const RegistrationResult rg = greg->registerClouds(srcN, target, Transform::Identity());
    CHECK(rg.fitness_ > 0.99 && std::fabs(rg.transformation_[12] - 0.03) < 1e-3 && std::fabs(rg.transformation_[13] + 0.02) < 1e-3);
  }
  // normals: plane z=0 seen from above -> +z
  PointCloud flat;
  std::mt19937 rng(5);
  std::uniform_real_distribution<double> u(-3, 3);
  for (int i = 0; i < 4000; ++i) flat.points_.push_back({{u(rng), u(rng), -1.5}});
  p2p->knnNormalEstimation_ = 20;
  p2p->maxRadiusNormalEstimation_ = 1.0;
  reg->estimateNormalsOrCovariancesIfNeeded(&flat);
  CHECK(flat.HasNormals());
  for (auto& n : flat.normals_) CHECK(std::fabs(n[2] - 1.0) < 1e-6);
  // croppers
  MaxRadiusCroppingVolume ball(2.0);
  auto cropped = ball.crop(flat);
  size_t expect = 0;
  for (auto& p : flat.points_) expect += std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) <= 2.0;
  CHECK(cropped->points_.size() == expect && cropped->normals_.size() == expect);
  // colours ride along: kept by the cropper, averaged by voxelize, written as the packed rgb field of the PCD
  {
    PointCloud tinted = flat;
    tinted.colors_.resize(tinted.points_.size());
    for (size_t i = 0; i < tinted.points_.size(); ++i) tinted.colors_[i] = {(double)(i % 5) / 4.0, 0.5, 1.0};
    auto kept = ball.crop(tinted);
    CHECK(kept->HasColors() && kept->colors_.size() == expect);
    size_t k = 0;
    for (size_t i = 0; i < tinted.points_.size() && k < 3; ++i) {
      const auto& p = tinted.points_[i];
      if (std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) <= 2.0) {
        CHECK(kept->colors_[k][0] == tinted.colors_[i][0] && kept->colors_[k][2] == 1.0);
        ++k;
      }
    }
    PointCloud small = tinted;
    voxelize(0.5, &small);
    CHECK(small.HasColors());
    for (auto& c : small.colors_) CHECK(c[0] >= 0.0 && c[0] <= 1.0 && std::fabs(c[1] - 0.5) < 1e-6 && std::fabs(c[2] - 1.0) < 1e-6);
    const char* dir = std::getenv("O3DS_TEST_TMPDIR");
    const std::string name = std::string(dir ? dir : "/tmp") + "/adapter_tinted.pcd";
    CHECK(saveToFile(name, tinted));
    std::FILE* f = std::fopen(name.c_str(), "rb");
    CHECK(f != nullptr);
    std::vector<char> blob(1 << 20);
    const size_t got = std::fread(blob.data(), 1, blob.size(), f);
    std::fclose(f);
    const std::string text(blob.data(), got);
    const size_t at = text.find("DATA binary\n");
    CHECK(text.find("FIELDS x y z normal_x normal_y normal_z rgb\n") != std::string::npos && got == at + 12 + tinted.points_.size() * 28);
    const unsigned char* row3 = reinterpret_cast<const unsigned char*>(blob.data()) + at + 12 + 3 * 28 + 24;
    CHECK(row3[0] == 255 && row3[1] == 128 && row3[2] == 191 && row3[3] == 0);  // b = 1.0, g = 0.5 (rounds to 128), r = 0.75 (191.25 -> 191)
  }
  // voxelize / transform
  PointCloud v = flat;
  voxelize(0.5, &v);
  CHECK(v.points_.size() < flat.points_.size() && v.points_.size() > 100);
  auto moved = transform(smallPose(1, 0, 0, 0), flat);
  CHECK(std::fabs((*moved).points_[7][0] - flat.points_[7][0] - 1.0) < 1e-6);
  // device submap: insert, then scan-to-map
  DeviceSubmap sub;
  MinMaxRadiusCroppingVolume mapCrop(0.0, 50.0), matchCrop(0.0, 50.0);
  sub.insertScan(target, Transform::Identity(), 0.05, &mapCrop, 0.5);
  CHECK(sub.size() > 1000 && sub.size() <= target.points_.size());
  const RegistrationResult r2 = sub.scanToMapRegistration(source, &matchCrop, Transform::Identity(), Transform::Identity(), *p2p);
  CHECK(r2.fitness_ > 0.99 && std::fabs(r2.transformation_[12] - 0.03) < 2e-3);
  DeviceSubmap emptyMap;
  CHECK(throws([&] { emptyMap.scanToMapRegistration(source, &matchCrop, Transform::Identity(), Transform::Identity(), *p2p); }));
  // saveToFile: header + float32 rows; the suffix rule of output.cpp:41-45
  {
    const char* dir = std::getenv("O3DS_TEST_TMPDIR");
    const std::string base = std::string(dir ? dir : "/tmp") + "/adapter_map";
    CHECK(saveToFile(base, flat));
    CHECK(sub.saveToFile(base + "_sub.pcd"));
    std::FILE* f = std::fopen((base + ".pcd").c_str(), "rb");
    CHECK(f != nullptr);
    std::vector<char> blob(1 << 20);
    const size_t got = std::fread(blob.data(), 1, blob.size(), f);
    std::fclose(f);
    const std::string text(blob.data(), got);
    const size_t at = text.find("DATA binary\n");
    CHECK(at != std::string::npos && text.find("FIELDS x y z normal_x normal_y normal_z\n") != std::string::npos);
    CHECK(text.find("POINTS " + std::to_string(flat.points_.size()) + "\n") != std::string::npos);
    CHECK(got == at + 12 + flat.points_.size() * 24);
    float row[6];
    std::memcpy(row, blob.data() + at + 12 + 7 * 24, sizeof(row));
    CHECK(row[0] == (float)flat.points_[7][0] && row[1] == (float)flat.points_[7][1] && row[2] == (float)flat.points_[7][2]);
    CHECK(row[3] == (float)flat.normals_[7][0] && row[5] == (float)flat.normals_[7][2]);
    CHECK(!saveToFile("/no/such/directory/x", flat));
  }
  std::puts("gpu checks ok");
}

int main(int argc, char** argv) {
  noGpuChecks();
  if (argc > 1 && !std::strcmp(argv[1], "--no-gpu")) return 0;
  gpuChecks();
  return 0;
}
