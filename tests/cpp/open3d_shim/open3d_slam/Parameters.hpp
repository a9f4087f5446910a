// NOT open3d_slam's Parameters.hpp: the two parameter structs integration/o3ds_open3d_slam.hpp reads, with the reference's member
// names and defaults (Parameters.hpp:52-58,84-91), so that the integration header can be exercised on a box without the reference
// checkout (tests/cpp/test_integration.cpp).  Inside open3d_slam the real header is found instead.
#pragma once
#include <string>
namespace o3d_slam {
struct ScanCroppingParameters {
  double croppingMinZ_ = -10.0;
  double croppingMaxZ_ = 10.0;
  double croppingMinRadius_ = 0.0;
  double croppingMaxRadius_ = 20.0;
  std::string cropperName_ = "MaxRadius";
};
struct SpaceCarvingParameters {
  double voxelSize_ = 0.1;
  double maxRaytracingLength_ = 20.0;
  double truncationDistance_ = 0.1;
  int carveSpaceEveryNscans_ = 10;
  double minDotProductWithNormal_ = 0.5;
  double neighborhoodRadiusDenseMap_ = 0.1;
};
}  // namespace o3d_slam
