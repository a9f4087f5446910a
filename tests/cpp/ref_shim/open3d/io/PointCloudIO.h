// NOT Open3D (see ../../Eigen/eigen_shim.hpp)
#pragma once
#include <string>

#include "../geometry/PointCloud.h"
namespace open3d {
namespace io {
struct WritePointCloudOption {
  WritePointCloudOption(bool write_ascii = false, bool compressed = false, bool print_progress = false);
};
bool WritePointCloudToPCD(const std::string& filename, const geometry::PointCloud& pointcloud, const WritePointCloudOption& params);
bool WritePointCloud(const std::string& filename, const geometry::PointCloud& pointcloud, const WritePointCloudOption& params = WritePointCloudOption());
}  // namespace io
}  // namespace open3d
