// NOT Open3D (see ../../Eigen/eigen_shim.hpp)
#pragma once
#include <string>
#include <vector>
namespace open3d {
namespace utility {
std::vector<std::string> SplitString(const std::string&, const std::string& delimiters = " ", bool trim_empty_str = true);
}  // namespace utility
}  // namespace open3d
