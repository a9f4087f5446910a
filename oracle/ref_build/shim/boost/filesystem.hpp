// NOT Boost: open3d_slam/src/output.cpp creates one directory through boost::filesystem; the same through the C++17 standard library
#pragma once
#include <filesystem>
namespace boost {
namespace filesystem {
using path = std::filesystem::path;
inline bool create_directory(const path& p) {
  std::error_code ec;
  return std::filesystem::create_directory(p, ec);
}
inline bool create_directories(const path& p) {
  std::error_code ec;
  return std::filesystem::create_directories(p, ec);
}
inline bool exists(const path& p) { return std::filesystem::exists(p); }
}  // namespace filesystem
}  // namespace boost
