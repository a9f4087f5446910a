// NOT Open3D (see ../../Eigen/eigen_shim.hpp)
#pragma once
#include "../../Eigen/eigen_shim.hpp"
namespace open3d {
namespace utility {
Eigen::Matrix4d TransformVector6dToMatrix4d(const Eigen::Vector6d&);
Eigen::Vector6d TransformMatrix4dToVector6d(const Eigen::Matrix4d&);
}  // namespace utility
}  // namespace open3d
