// NOT Open3D (see ../../../Eigen/eigen_shim.hpp)
#pragma once
#include <vector>

#include "../../../Eigen/eigen_shim.hpp"
namespace open3d {
namespace pipelines {
namespace registration {
class PoseGraphNode {
 public:
  PoseGraphNode(const Eigen::Matrix4d& pose = Eigen::Matrix4d::Identity());
  Eigen::Matrix4d pose_;
};
class PoseGraphEdge {
 public:
  PoseGraphEdge(int source = -1, int target = -1, const Eigen::Matrix4d& T = Eigen::Matrix4d::Identity(),
                const Eigen::Matrix<double, 6, 6>& info = Eigen::Matrix<double, 6, 6>::Identity(), bool uncertain = false, double confidence = 1.0);
  int source_node_id_, target_node_id_;
  Eigen::Matrix4d transformation_;
  Eigen::Matrix<double, 6, 6> information_;
  bool uncertain_;
  double confidence_;
};
class PoseGraph {
 public:
  std::vector<PoseGraphNode> nodes_;
  std::vector<PoseGraphEdge> edges_;
};
}  // namespace registration
}  // namespace pipelines
}  // namespace open3d
