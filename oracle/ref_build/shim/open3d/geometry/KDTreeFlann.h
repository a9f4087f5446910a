// NOT Open3D: KDTreeFlann is declared so that helpers.cpp compiles; its searches abort (the nearest-neighbour search is not part of
// /root/reference)
#pragma once
#include <vector>

#include "PointCloud.h"
namespace open3d {
namespace geometry {
class KDTreeFlann {
 public:
  KDTreeFlann() {}
  bool SetGeometry(const PointCloud& cloud);
  int SearchKNN(const Eigen::Vector3d& query, int knn, std::vector<int>& indices, std::vector<double>& distance2) const;
  int SearchRadius(const Eigen::Vector3d& query, double radius, std::vector<int>& indices, std::vector<double>& distance2) const;
  int SearchHybrid(const Eigen::Vector3d& query, double radius, int max_nn, std::vector<int>& indices, std::vector<double>& distance2) const;
};
}  // namespace geometry
}  // namespace open3d
