// NOT Open3D: the pose-graph data types OptimizationProblem.hpp names (loop closure, out of scope)
#pragma once
#include <vector>

#include "../../../Eigen/mini_eigen.hpp"
namespace open3d {
namespace pipelines {
namespace registration {
class PoseGraphNode {
 public:
  PoseGraphNode(const Eigen::Matrix4d& pose = Eigen::Matrix4d::Identity()) : pose_(pose) {}
  Eigen::Matrix4d pose_;
};
class PoseGraphEdge {
 public:
  PoseGraphEdge(int source = -1, int target = -1, const Eigen::Matrix4d& T = Eigen::Matrix4d::Identity(),
                const Eigen::Matrix<double, 6, 6>& info = Eigen::Matrix<double, 6, 6>::Identity(), bool uncertain = false, double confidence = 1.0)
      : source_node_id_(source), target_node_id_(target), transformation_(T), information_(info), uncertain_(uncertain), confidence_(confidence) {}
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
