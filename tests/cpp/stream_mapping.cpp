// The per-scan mapping loop of open3d_slam (Mapper::addRangeMeasurement, Mapper.cpp:101-181, without the odometry prior and the
// bookkeeping) driven through the C++ host classes of open3d_slam_amd/host/o3ds_mapping.hpp with HOST clouds at the seam, as the
// reference hands them over: every scan is uploaded, pre-processed on the device, downloaded (merge_ / match_), the match scan is
// uploaded again for the registration against the device-resident submap, the merge scan once more for the insertion.  So the rate
// this prints is the PCIe-inclusive one of the drop-in with unchanged callers; the numbers of bench.py /
// scripts/bench_stream.py keep the clouds on the device.
//   stream_mapping <scans.bin>     scans.bin: int32 frames, int32 points, then per frame 16 doubles (map <- sensor, column-major)
//                                  and points x 3 doubles in the sensor frame (scripts/bench_stream_cpp.py writes it)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../open3d_slam_amd/host/o3ds_mapping.hpp"

using namespace o3d_slam;
using Clock = std::chrono::steady_clock;

static double ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

static Transform inverseRigid(const Transform& T) {  // [R t]^-1 = [R^T  -R^T t]
  Transform r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[j * 4 + i] = T.m[i * 4 + j];
  for (int i = 0; i < 3; ++i) r.m[12 + i] = -(r.m[i] * T.m[12] + r.m[4 + i] * T.m[13] + r.m[8 + i] * T.m[14]);
  return r;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: stream_mapping scans.bin\n");
    return 2;
  }
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int frames = 0, npts = 0;
  if (std::fread(&frames, 4, 1, f) != 1 || std::fread(&npts, 4, 1, f) != 1) return 2;
  std::vector<Transform> truth(frames);
  std::vector<PointCloud> scans(frames);
  for (int k = 0; k < frames; ++k) {
    if (std::fread(truth[k].m.data(), sizeof(double), 16, f) != 16) return 2;
    scans[k].points_.resize(npts);
    if (std::fread(scans[k].points_.data(), sizeof(double) * 3, npts, f) != (size_t)npts) return 2;
  }
  std::fclose(f);

  MapperParameters p;  // the shipped Lua defaults of the hot-path knobs (parameter_structure_definitions.lua:49-72,94-118), point-to-plane
  p.scanMatcher_.icp_.maxNumIter_ = 50;
  p.scanMatcher_.icp_.maxCorrespondenceDistance_ = 1.0;
  p.scanMatcher_.icp_.knn_ = 20;
  p.scanMatcher_.icp_.maxDistanceKnn_ = 3.0;
  p.scanProcessing_.voxelSize_ = 0.1;
  p.scanProcessing_.cropper_.cropperName_ = "MinMaxRadius";
  p.scanProcessing_.cropper_.croppingMinRadius_ = 2.0;
  p.scanProcessing_.cropper_.croppingMaxRadius_ = 30.0;
  p.mapBuilder_.mapVoxelSize_ = 0.1;
  p.mapBuilder_.cropper_ = p.scanProcessing_.cropper_;

  auto scan2MapReg = scanToMapRegistrationFactory(p);
  Submap submap(0, 0);
  submap.setParameters(p);
  Transform mapToRangeSensor = Transform::Identity();  // the map frame is the first sensor frame
  double tPre = 0, tReg = 0, tIns = 0, minFitness = 1.0;
  for (int k = 0; k < frames; ++k) {
    const auto t0 = Clock::now();
    const ProcessedScans ps = scan2MapReg->processForScanMatchingAndMerging(scans[k], mapToRangeSensor);
    const auto t1 = Clock::now();
    if (k > 0) {
      const RegistrationResult r = scan2MapReg->scanToMapRegistration(*ps.match_, submap, mapToRangeSensor, mapToRangeSensor);
      if (r.fitness_ < p.scanMatcher_.minRefinementFitness_) {  // Mapper.cpp:151-156
        std::fprintf(stderr, "frame %d: fitness %.3f below minRefinementFitness_\n", k, r.fitness_);
        return 1;
      }
      minFitness = std::min(minFitness, r.fitness_);
      for (int i = 0; i < 16; ++i) mapToRangeSensor.m[i] = r.transformation_[i];
    }
    const auto t2 = Clock::now();
    submap.insertScan(scans[k], *ps.merge_, mapToRangeSensor, Time(), true);  // SubmapCollection::insertScan always asks for carving (SubmapCollection.cpp:178,189,203)
    const auto t3 = Clock::now();
    if (k > 0) tPre += ms(t0, t1), tReg += ms(t1, t2), tIns += ms(t2, t3);
  }
  const Transform rel = inverseRigid(truth[0]) * truth[frames - 1];  // true pose of the last sensor frame in the first one
  const double dx = rel.m[12] - mapToRangeSensor.m[12], dy = rel.m[13] - mapToRangeSensor.m[13], dz = rel.m[14] - mapToRangeSensor.m[14];
  const double dt = std::sqrt(dx * dx + dy * dy + dz * dz);
  const int n = frames - 1;
  const size_t mapPoints = submap.getMapPointCloud().points_.size();  // one download of the final map
  std::printf(
      "{\"workload\": \"C++ host classes, host clouds at the seam (PCIe-inclusive), %d raw pts/scan, %d frames\", \"scans_per_sec_mapping_only\": %.1f, "
      "\"ms_per_scan\": {\"preprocess\": %.3f, \"registration\": %.3f, \"insert\": %.3f}, \"map_points\": %zu, \"min_fitness\": %.4f, "
      "\"final_translation_error_m\": %.5f}\n",
      npts, frames, 1e3 * n / (tPre + tReg + tIns), tPre / n, tReg / n, tIns / n, mapPoints, minFitness, dt);
  return dt < 0.05 ? 0 : 1;
}
