// NOT Open3D: file output is not part of the path; the writers report failure
#pragma once
#include <string>

#include "../geometry/PointCloud.h"
namespace open3d {
namespace io {
struct WritePointCloudOption {
  WritePointCloudOption(bool write_ascii = false, bool compressed = false, bool print_progress = false) {}
};
inline bool WritePointCloudToPCD(const std::string&, const geometry::PointCloud&, const WritePointCloudOption&) { return false; }
inline bool WritePointCloud(const std::string&, const geometry::PointCloud&, const WritePointCloudOption& = WritePointCloudOption()) { return false; }
}  // namespace io
}  // namespace open3d
