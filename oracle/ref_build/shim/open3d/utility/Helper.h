// NOT Open3D
#pragma once
#include <string>
#include <vector>
namespace open3d {
namespace utility {
std::vector<std::string> SplitString(const std::string& s, const std::string& delimiters = " ", bool trim_empty_str = true);
}  // namespace utility
}  // namespace open3d
