// NOT Eigen.  Declarations with the spelling of the part of Eigen 3 that open3d_slam's mapping sources use, so that those sources --
// with integration/open3d_slam_o3ds.patch applied -- can be type-checked (g++ -fsyntax-only) in an image that has neither Eigen nor
// Open3D (tests/test_integration_patch.py).  Nothing here computes anything; most members are declared and never defined.
#pragma once
#include <cmath>
#include <cstddef>
#include <iosfwd>
#include <memory>
#include <vector>
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
namespace Eigen {
constexpr int Dynamic = -1;
enum { Isometry = 1, Affine = 2, ColMajor = 0, RowMajor = 1 };
typedef std::ptrdiff_t Index;
template <typename T>
struct aligned_allocator : std::allocator<T> {
  template <typename U>
  struct rebind {
    typedef aligned_allocator<U> other;
  };
};
template <typename S, int R, int C>
class Matrix;
template <typename S>
class ArrayExpr;
template <typename S, int R, int C>
class Matrix {
 public:
  typedef S Scalar;
  Matrix();
  Matrix(S x);
  Matrix(S x, S y);
  Matrix(S x, S y, S z);
  Matrix(S x, S y, S z, S w);
  struct Dims {
    Dims(long, long);
  };
  template <typename S2>
  Matrix(const ArrayExpr<S2>&);
  template <int R2, int C2>
  Matrix(const Matrix<S, R2, C2>&);
  static Matrix Identity();
  static Matrix Identity(int, int);
  static Matrix Zero();
  static Matrix Zero(int);
  static Matrix Zero(int, int);
  static Matrix Ones();
  static Matrix Constant(S);
  static Matrix Random();
  static Matrix UnitX();
  static Matrix UnitY();
  static Matrix UnitZ();
  S& operator()(Index i);
  const S& operator()(Index i) const;
  S& operator()(Index i, Index j);
  const S& operator()(Index i, Index j) const;
  S& operator[](Index i);
  const S& operator[](Index i) const;
  S& x();
  const S& x() const;
  S& y();
  const S& y() const;
  S& z();
  const S& z() const;
  S& w();
  const S& w() const;
  S* data();
  const S* data() const;
  Index rows() const;
  Index cols() const;
  Index size() const;
  S norm() const;
  S squaredNorm() const;
  S sum() const;
  S mean() const;
  S trace() const;
  S determinant() const;
  S maxCoeff() const;
  S minCoeff() const;
  S dot(const Matrix&) const;
  Matrix cross(const Matrix&) const;
  Matrix normalized() const;
  void normalize();
  Matrix<S, C, R> transpose() const;
  Matrix inverse() const;
  Matrix cwiseProduct(const Matrix&) const;
  Matrix cwiseAbs() const;
  Matrix cwiseMax(const Matrix&) const;
  Matrix cwiseMin(const Matrix&) const;
  ArrayExpr<S> array() const;
  Matrix& setZero();
  Matrix& setIdentity();
  Matrix& setConstant(S);
  bool allFinite() const;
  bool hasNaN() const;
  bool isApprox(const Matrix&, S prec = S()) const;
  template <int BR, int BC>
  Matrix<S, BR, BC>& block(Index, Index);
  template <int BR, int BC>
  const Matrix<S, BR, BC>& block(Index, Index) const;
  Matrix<S, Dynamic, Dynamic>& block(Index, Index, Index, Index);
  template <int N>
  Matrix<S, N, 1>& head();
  template <int N>
  const Matrix<S, N, 1>& head() const;
  template <int N>
  Matrix<S, N, 1>& tail();
  Matrix<S, Dynamic, 1>& head(Index);
  Matrix<S, R, 1>& col(Index);
  const Matrix<S, R, 1>& col(Index) const;
  Matrix<S, 1, C>& row(Index);
  Matrix<S, 3, 3>& topLeftCorner(Index, Index);
  template <int A, int B>
  Matrix<S, A, B>& topLeftCorner();
  template <int A, int B>
  Matrix<S, A, B>& topRightCorner();
  Matrix<S, 3, 1> eulerAngles(int, int, int) const;
  template <typename T>
  Matrix<T, R, C> cast() const;
  Matrix operator-() const;
  Matrix operator+(const Matrix&) const;
  Matrix operator-(const Matrix&) const;
  Matrix& operator+=(const Matrix&);
  Matrix& operator-=(const Matrix&);
  Matrix& operator*=(S);
  Matrix& operator*=(const Matrix<S, C, C>&);
  Matrix& operator/=(S);
  Matrix operator*(S) const;
  Matrix operator/(S) const;
  template <int C2>
  Matrix<S, R, C2> operator*(const Matrix<S, C, C2>&) const;
  bool operator==(const Matrix&) const;
  bool operator!=(const Matrix&) const;
  Matrix& operator<<(S);
  Matrix& operator,(S);
};
template <typename S, int R, int C>
Matrix<S, R, C> operator*(S, const Matrix<S, R, C>&);
template <typename S, int R, int C>
Matrix<S, R, C> operator*(int, const Matrix<S, R, C>&);
template <typename S, int R, int C>
std::ostream& operator<<(std::ostream&, const Matrix<S, R, C>&);
template <typename S>
class ArrayExpr {
 public:
  ArrayExpr operator*(const ArrayExpr&) const;
  ArrayExpr operator/(const ArrayExpr&) const;
  ArrayExpr operator+(const ArrayExpr&) const;
  ArrayExpr operator-(const ArrayExpr&) const;
  ArrayExpr operator*(S) const;
  ArrayExpr operator/(S) const;
  ArrayExpr<bool> operator<(const ArrayExpr&) const;
  ArrayExpr<bool> operator>(const ArrayExpr&) const;
  ArrayExpr<bool> operator<=(const ArrayExpr&) const;
  ArrayExpr<bool> operator>=(const ArrayExpr&) const;
  ArrayExpr<bool> operator&&(const ArrayExpr<bool>&) const;
  ArrayExpr floor() const;
  ArrayExpr abs() const;
  ArrayExpr sqrt() const;
  bool all() const;
  bool any() const;
  template <typename T>
  ArrayExpr<T> cast() const;
  Matrix<S, 3, 1> matrix() const;
};
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 6, 1> Vector6d;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<int, 3, 1> Vector3i;
typedef Matrix<int, 2, 1> Vector2i;
typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<int, Dynamic, 1> VectorXi;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<int, Dynamic, Dynamic> MatrixXi;
template <typename S>
class Quaternion {
 public:
  Quaternion();
  Quaternion(S w, S x, S y, S z);
  explicit Quaternion(const Matrix<S, 3, 3>&);
  static Quaternion Identity();
  S& x();
  S& y();
  S& z();
  S& w();
  const S& x() const;
  const S& y() const;
  const S& z() const;
  const S& w() const;
  Quaternion normalized() const;
  void normalize();
  Quaternion inverse() const;
  Quaternion conjugate() const;
  Quaternion slerp(S t, const Quaternion&) const;
  Matrix<S, 3, 3> toRotationMatrix() const;
  Matrix<S, 3, 3> matrix() const;
  Matrix<S, 4, 1>& coeffs();
  Quaternion operator*(const Quaternion&) const;
  Matrix<S, 3, 1> operator*(const Matrix<S, 3, 1>&) const;
  S angularDistance(const Quaternion&) const;
};
typedef Quaternion<double> Quaterniond;
template <typename S>
class AngleAxis {
 public:
  AngleAxis();
  AngleAxis(S angle, const Matrix<S, 3, 1>& axis);
  explicit AngleAxis(const Matrix<S, 3, 3>&);
  explicit AngleAxis(const Quaternion<S>&);
  S angle() const;
  Matrix<S, 3, 1> axis() const;
  Matrix<S, 3, 3> toRotationMatrix() const;
  Quaternion<S> operator*(const AngleAxis&) const;
  operator Quaternion<S>() const;
};
typedef AngleAxis<double> AngleAxisd;
template <typename S, int D>
class Translation {
 public:
  Translation();
  Translation(S, S, S);
  explicit Translation(const Matrix<S, D, 1>&);
};
typedef Translation<double, 3> Translation3d;
template <typename S, int D, int Mode>
class Transform {
 public:
  Transform();
  Transform(const Matrix<S, D + 1, D + 1>&);
  explicit Transform(const Quaternion<S>&);
  explicit Transform(const Matrix<S, D, D>&);
  Transform(const Translation<S, D>&);
  static Transform Identity();
  Matrix<S, D + 1, D + 1>& matrix();
  const Matrix<S, D + 1, D + 1>& matrix() const;
  Matrix<S, D, 1>& translation();
  const Matrix<S, D, 1>& translation() const;
  Matrix<S, D, D> rotation() const;
  Matrix<S, D, D>& linear();
  const Matrix<S, D, D>& linear() const;
  Transform inverse() const;
  Transform& setIdentity();
  Transform& translate(const Matrix<S, D, 1>&);
  Transform& pretranslate(const Matrix<S, D, 1>&);
  Transform& rotate(const Quaternion<S>&);
  Transform& rotate(const Matrix<S, D, D>&);
  Transform& rotate(const AngleAxis<S>&);
  Transform operator*(const Transform&) const;
  Transform& operator*=(const Transform&);
  Matrix<S, D, 1> operator*(const Matrix<S, D, 1>&) const;
  Transform operator*(const Translation<S, D>&) const;
  Transform operator*(const Quaternion<S>&) const;
  Transform& operator=(const Matrix<S, D + 1, D + 1>&);
  bool isApprox(const Transform&, S prec = S()) const;
  template <typename T>
  Transform<T, D, Mode> cast() const;
  S& operator()(Index, Index);
  const S& operator()(Index, Index) const;
};
template <typename S, int D, int Mode>
Transform<S, D, Mode> operator*(const Translation<S, D>&, const Quaternion<S>&);
template <typename S, int D>
Transform<S, D, Isometry> operator*(const Translation<S, D>&, const Transform<S, D, Isometry>&);
template <typename S, int D, int Mode>
std::ostream& operator<<(std::ostream&, const Transform<S, D, Mode>&);
typedef Transform<double, 3, Isometry> Isometry3d;
typedef Transform<double, 3, Affine> Affine3d;
template <typename T>
class Map : public T {
 public:
  explicit Map(const typename T::Scalar*);
  Map(const typename T::Scalar*, Index, Index);
};
template <typename T>
class Map<const T> : public T {
 public:
  explicit Map(const typename T::Scalar*);
  Map(const typename T::Scalar*, Index, Index);
};
template <typename M>
class SelfAdjointEigenSolver {
 public:
  explicit SelfAdjointEigenSolver(const M&);
  M eigenvectors() const;
  Matrix<double, 3, 1> eigenvalues() const;
};
}  // namespace Eigen
