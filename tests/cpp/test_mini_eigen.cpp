// Self-checks of oracle/ref_build/shim/Eigen/mini_eigen.hpp, the stand-in that lets open3d_slam's own sources run in this image: the pieces
// whose silent failure would make a "reference" run meaningless -- the write-back of `T.matrix() *= M` / `T.matrix() = M`, the general
// inverse, rigid inverse and composition, quaternion <-> matrix, the coefficient-wise expressions the voxel index is built from.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../oracle/ref_build/shim/Eigen/Dense"

#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) {                                                         \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                         \
    }                                                                   \
  } while (0)

static double maxdiff(const Eigen::Matrix4d& a, const Eigen::Matrix4d& b) { return (a - b).array().abs().maxCoeff(); }

int main() {
  using namespace Eigen;
  // a rigid transform from a quaternion and a translation
  const Quaterniond q = (AngleAxisd(0.3, Vector3d::UnitZ()) * AngleAxisd(-0.2, Vector3d::UnitY()) * AngleAxisd(0.1, Vector3d::UnitX())).normalized();
  Isometry3d T(q);
  T.translation() = Vector3d(1.0, -2.0, 0.5);
  const Matrix4d M = T.matrix();
  CHECK(std::fabs(M(0, 3) - 1.0) < 1e-15 && std::fabs(M(3, 3) - 1.0) < 1e-15 && M(3, 0) == 0.0);
  // general inverse = rigid inverse
  CHECK(maxdiff(M.inverse(), T.inverse().matrix()) < 1e-14);
  CHECK(maxdiff(M * M.inverse(), Matrix4d::Identity()) < 1e-14);
  // composition and application
  const Vector3d p(0.3, 0.4, -1.2);
  const Vector3d via = (T * T) * p, twice = T * (T * p);
  CHECK((via - twice).norm() < 1e-14);
  // `X.matrix() *= M` and `X.matrix() = M` write through (Odometry.cpp:72, Mapper.cpp:159) -- a temporary here, written back when it dies
  Isometry3d X = Isometry3d::Identity();
  X.matrix() *= M;
  CHECK(maxdiff(X.matrix(), M) < 1e-15);
  X.matrix() *= M.inverse();
  CHECK(maxdiff(X.matrix(), Matrix4d::Identity()) < 1e-14);
  X.matrix() = M;
  CHECK(maxdiff(X.matrix(), M) == 0.0);
  const Isometry3d& cX = X;  // reading through a const reference does not write anything back
  CHECK(maxdiff(cX.matrix(), M) == 0.0);
  // quaternion <-> rotation matrix, slerp end points
  const Quaterniond back(T.rotation());
  CHECK(std::fabs(std::fabs(back.w() * q.w() + back.x() * q.x() + back.y() * q.y() + back.z() * q.z()) - 1.0) < 1e-14);
  const Quaterniond s0 = Quaterniond::Identity().slerp(0.0, q), s1 = Quaterniond::Identity().slerp(1.0, q);
  CHECK(std::fabs(s0.w() - 1.0) < 1e-15 && std::fabs(s1.w() - q.w()) < 1e-14);
  // the expressions the voxel index and the voxel centre are built from (VoxelHashMap.hpp:47-70)
  const Vector3d voxel = Vector3d::Constant(0.25), pt(-0.26, 0.0, 0.74);
  const Vector3d coord = pt.array() / voxel.array();
  CHECK(std::floor(coord(0)) == -2.0 && std::floor(coord(1)) == 0.0 && std::floor(coord(2)) == 2.0);
  const Vector3i key(-2, 0, 2);
  const Vector3d centre = key.cast<double>().array() * voxel.array() + voxel.array() * 0.5;
  CHECK(centre(0) == -0.375 && centre(2) == 0.625);
  CHECK((key.array() == Vector3i(-2, 0, 2).array()).all() && !(key.array() == Vector3i(-2, 1, 2).array()).all());
  // normalized(): a zero vector stays zero (Eigen 3.4), isValidColor's odd use of all()
  CHECK(Vector3d::Zero().normalized().norm() == 0.0 && std::fabs(Vector3d(3, 0, 4).normalized().norm() - 1.0) < 1e-15);
  CHECK(Vector3d(0.2, 0.3, 0.4).array().all() && !Vector3d(0.2, 0.0, 0.4).array().all());
  // head / block give values (assigning to them must not compile: they are const)
  const Vector4d h(1, 2, 3, 4);
  const Matrix3d blk = M.block<3, 3>(0, 0);
  CHECK(h.head<3>()(2) == 3.0 && blk(1, 1) == T.linear()(1, 1));
  std::printf("mini_eigen self-checks passed\n");
  return 0;
}
