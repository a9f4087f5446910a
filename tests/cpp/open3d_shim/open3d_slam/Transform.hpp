// stands in for open3d_slam/include/open3d_slam/Transform.hpp:15 (using Transform = Eigen::Isometry3d)
#pragma once
#include "../Eigen/Dense"
namespace o3d_slam {
using Transform = Eigen::Isometry3d;
}
