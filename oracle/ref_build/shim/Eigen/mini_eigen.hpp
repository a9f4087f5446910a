// NOT Eigen.  A small, FUNCTIONAL stand-in with the spelling of the part of Eigen 3 that open3d_slam's own sources use, written so that
// those sources -- croppers.cpp, helpers.cpp, Voxel.cpp, VoxelHashMap.cpp, ... exactly as they lie under /root/reference -- can be
// COMPILED AND RUN in an image that has neither Eigen nor Open3D (oracle/ref_build/Makefile -> oracle/_ref/libo3dslam_ref.so).
// What that pins: the reference's own LOGIC on the hot path (which side of a comparison is inclusive, what is inverted, how a voxel index
// is formed, what is accumulated and in which order, what is normalised, what is emitted first).  What it does NOT pin: Eigen's
// arithmetic -- every operation here is the plain left-to-right scalar expression (dot = a0*b0 + a1*b1 + a2*b2, norm = sqrt(dot),
// v / s = true division per coefficient, M * v row by row), value semantics throughout, no expression templates, no vectorisation.
// Test infrastructure only (see oracle/ref_build/README.md).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <iosfwd>
#include <memory>
#include <ostream>
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
using aligned_allocator = std::allocator<T>;

template <typename S, int R, int C>
class Matrix;

// ---- coefficient-wise values (x.array())
template <typename S, int R, int C>
class Array {
 public:
  S v[R * C > 0 ? R * C : 1];
  Array() : v{} {}
  static constexpr int N = R * C;
#define O3DS_ARR_BIN(op)                            \
  Array operator op(const Array& o) const {         \
    Array r;                                        \
    for (int i = 0; i < N; ++i) r.v[i] = v[i] op o.v[i]; \
    return r;                                       \
  }                                                 \
  Array operator op(S s) const {                    \
    Array r;                                        \
    for (int i = 0; i < N; ++i) r.v[i] = v[i] op s; \
    return r;                                       \
  }
  O3DS_ARR_BIN(+)
  O3DS_ARR_BIN(-)
  O3DS_ARR_BIN(*)
  O3DS_ARR_BIN(/)
#undef O3DS_ARR_BIN
#define O3DS_ARR_CMP(op)                                 \
  Array<bool, R, C> operator op(const Array& o) const {  \
    Array<bool, R, C> r;                                 \
    for (int i = 0; i < N; ++i) r.v[i] = v[i] op o.v[i]; \
    return r;                                            \
  }                                                      \
  Array<bool, R, C> operator op(S s) const {             \
    Array<bool, R, C> r;                                 \
    for (int i = 0; i < N; ++i) r.v[i] = v[i] op s;      \
    return r;                                            \
  }
  O3DS_ARR_CMP(<)
  O3DS_ARR_CMP(<=)
  O3DS_ARR_CMP(>)
  O3DS_ARR_CMP(>=)
  O3DS_ARR_CMP(==)
#undef O3DS_ARR_CMP
  Array<bool, R, C> operator&&(const Array<bool, R, C>& o) const {
    Array<bool, R, C> r;
    for (int i = 0; i < N; ++i) r.v[i] = v[i] && o.v[i];
    return r;
  }
  bool all() const {  // Eigen: every coefficient is "true" (non-zero)
    for (int i = 0; i < N; ++i)
      if (!v[i]) return false;
    return true;
  }
  bool any() const {
    for (int i = 0; i < N; ++i)
      if (v[i]) return true;
    return false;
  }
  Array abs() const {
    Array r;
    for (int i = 0; i < N; ++i) r.v[i] = v[i] < S(0) ? S(-v[i]) : v[i];
    return r;
  }
  Array floor() const {
    Array r;
    for (int i = 0; i < N; ++i) r.v[i] = std::floor(v[i]);
    return r;
  }
  Array sqrt() const {
    Array r;
    for (int i = 0; i < N; ++i) r.v[i] = std::sqrt(v[i]);
    return r;
  }
  S maxCoeff() const {
    S m = v[0];
    for (int i = 1; i < N; ++i) m = v[i] > m ? v[i] : m;
    return m;
  }
  S minCoeff() const {
    S m = v[0];
    for (int i = 1; i < N; ++i) m = v[i] < m ? v[i] : m;
    return m;
  }
  template <typename T>
  Array<T, R, C> cast() const {
    Array<T, R, C> r;
    for (int i = 0; i < N; ++i) r.v[i] = (T)v[i];
    return r;
  }
  Matrix<S, R, C> matrix() const;
};
template <typename S, int R, int C>
Array<S, R, C> operator*(S s, const Array<S, R, C>& a) {
  return a * s;
}

// ---- fixed-size matrices, column-major, value semantics
template <typename S, int R, int C>
class Matrix {
 public:
  typedef S Scalar;
  static constexpr int N = R * C;
  S v[R * C];
  Matrix() : v{} {}
  Matrix(S x, S y) : v{} {
    static_assert(N == 2, "two coefficients");
    v[0] = x, v[1] = y;
  }
  Matrix(S x, S y, S z) : v{} {
    static_assert(N == 3, "three coefficients");
    v[0] = x, v[1] = y, v[2] = z;
  }
  Matrix(S x, S y, S z, S w) : v{} {
    static_assert(N == 4, "four coefficients");
    v[0] = x, v[1] = y, v[2] = z, v[3] = w;
  }
  Matrix(const Array<S, R, C>& a) : v{} {
    for (int i = 0; i < N; ++i) v[i] = a.v[i];
  }
  static Matrix Zero() { return Matrix(); }
  static Matrix Constant(S s) {
    Matrix m;
    for (int i = 0; i < N; ++i) m.v[i] = s;
    return m;
  }
  static Matrix Ones() { return Constant(S(1)); }
  static Matrix Identity() {
    Matrix m;
    for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = S(1);
    return m;
  }
  static Matrix UnitX() {
    Matrix m;
    m.v[0] = S(1);
    return m;
  }
  static Matrix UnitY() {
    Matrix m;
    m.v[1] = S(1);
    return m;
  }
  static Matrix UnitZ() {
    Matrix m;
    m.v[2] = S(1);
    return m;
  }
  S& operator()(Index i) { return v[i]; }
  const S& operator()(Index i) const { return v[i]; }
  S& operator()(Index i, Index j) { return v[j * R + i]; }
  const S& operator()(Index i, Index j) const { return v[j * R + i]; }
  S& operator[](Index i) { return v[i]; }
  const S& operator[](Index i) const { return v[i]; }
  S& x() { return v[0]; }
  const S& x() const { return v[0]; }
  S& y() { return v[1]; }
  const S& y() const { return v[1]; }
  S& z() { return v[2]; }
  const S& z() const { return v[2]; }
  S& w() { return v[3]; }
  const S& w() const { return v[3]; }
  S* data() { return v; }
  const S* data() const { return v; }
  Index rows() const { return R; }
  Index cols() const { return C; }
  Index size() const { return N; }
  S dot(const Matrix& o) const {
    S s = v[0] * o.v[0];
    for (int i = 1; i < N; ++i) s = s + v[i] * o.v[i];
    return s;
  }
  S squaredNorm() const { return dot(*this); }
  S norm() const { return std::sqrt(squaredNorm()); }
  S sum() const {
    S s = v[0];
    for (int i = 1; i < N; ++i) s = s + v[i];
    return s;
  }
  S maxCoeff() const { return array().maxCoeff(); }
  S minCoeff() const { return array().minCoeff(); }
  Matrix normalized() const {  // Eigen 3.4 MatrixBase::normalized(): "z > 0 ? n / sqrt(z) : n" with z = squaredNorm()
    const S z = squaredNorm();
    return z > S(0) ? *this / std::sqrt(z) : *this;
  }
  void normalize() { *this = normalized(); }
  Matrix cross(const Matrix& o) const {
    static_assert(N == 3, "cross product of 3-vectors");
    return Matrix(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]);
  }
  const Matrix<S, C, R> transpose() const {
    Matrix<S, C, R> t;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) t(j, i) = (*this)(i, j);
    return t;
  }
  Array<S, R, C> array() const {
    Array<S, R, C> a;
    for (int i = 0; i < N; ++i) a.v[i] = v[i];
    return a;
  }
  Matrix& setZero() { return *this = Matrix(); }
  Matrix& setIdentity() { return *this = Identity(); }
  Matrix& setConstant(S s) { return *this = Constant(s); }
  bool allFinite() const {
    for (int i = 0; i < N; ++i)
      if (!std::isfinite((double)v[i])) return false;
    return true;
  }
  bool hasNaN() const {
    for (int i = 0; i < N; ++i)
      if (v[i] != v[i]) return true;
    return false;
  }
  // (sub-matrix accessors return CONST values: writing through one -- which real Eigen allows and this stand-in could only drop --
  //  then fails to compile instead of silently doing nothing)
  template <int BR, int BC>
  const Matrix<S, BR, BC> block(Index r0, Index c0) const {
    Matrix<S, BR, BC> b;
    for (int i = 0; i < BR; ++i)
      for (int j = 0; j < BC; ++j) b(i, j) = (*this)(r0 + i, c0 + j);
    return b;
  }
  template <int K>
  const Matrix<S, K, 1> head() const {
    Matrix<S, K, 1> h;
    for (int i = 0; i < K; ++i) h.v[i] = v[i];
    return h;
  }
  template <int K>
  const Matrix<S, K, 1> tail() const {
    Matrix<S, K, 1> h;
    for (int i = 0; i < K; ++i) h.v[i] = v[N - K + i];
    return h;
  }
  const Matrix<S, R, 1> col(Index j) const {
    Matrix<S, R, 1> c;
    for (int i = 0; i < R; ++i) c.v[i] = (*this)(i, j);
    return c;
  }
  template <typename T>
  Matrix<T, R, C> cast() const {
    Matrix<T, R, C> r;
    for (int i = 0; i < N; ++i) r.v[i] = (T)v[i];
    return r;
  }
  Matrix operator-() const {
    Matrix r;
    for (int i = 0; i < N; ++i) r.v[i] = -v[i];
    return r;
  }
  Matrix operator+(const Matrix& o) const {
    Matrix r;
    for (int i = 0; i < N; ++i) r.v[i] = v[i] + o.v[i];
    return r;
  }
  Matrix operator-(const Matrix& o) const {
    Matrix r;
    for (int i = 0; i < N; ++i) r.v[i] = v[i] - o.v[i];
    return r;
  }
  Matrix& operator+=(const Matrix& o) { return *this = *this + o; }
  Matrix& operator-=(const Matrix& o) { return *this = *this - o; }
  Matrix operator*(S s) const {
    Matrix r;
    for (int i = 0; i < N; ++i) r.v[i] = v[i] * s;
    return r;
  }
  Matrix operator/(S s) const {
    Matrix r;
    for (int i = 0; i < N; ++i) r.v[i] = v[i] / s;
    return r;
  }
  // an int divisor (aggregatedPosition_ / numAggregatedPoints_): Eigen promotes it to the scalar type first
  template <typename I, typename = typename std::enable_if<std::is_integral<I>::value && !std::is_same<I, S>::value>::type>
  Matrix operator/(I s) const {
    return *this / (S)s;
  }
  Matrix& operator*=(S s) { return *this = *this * s; }
  Matrix& operator*=(const Matrix<S, C, C>& o) { return *this = *this * o; }
  Matrix& operator/=(S s) { return *this = *this / s; }
  template <int C2>
  Matrix<S, R, C2> operator*(const Matrix<S, C, C2>& o) const {
    Matrix<S, R, C2> r;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C2; ++j) {
        S s = (*this)(i, 0) * o(0, j);
        for (int k = 1; k < C; ++k) s = s + (*this)(i, k) * o(k, j);
        r(i, j) = s;
      }
    return r;
  }
  // general inverse by Gauss-Jordan elimination with partial pivoting (Eigen uses cofactors up to 4x4: same value, other rounding)
  const Matrix inverse() const {
    static_assert(R == C, "square");
    Matrix a = *this, inv = Identity();
    for (int c = 0; c < C; ++c) {
      int piv = c;
      for (int r = c + 1; r < R; ++r)
        if (std::fabs((double)a(r, c)) > std::fabs((double)a(piv, c))) piv = r;
      if (piv != c)
        for (int k = 0; k < C; ++k) {
          std::swap(a(c, k), a(piv, k));
          std::swap(inv(c, k), inv(piv, k));
        }
      const S d = a(c, c);
      for (int k = 0; k < C; ++k) {
        a(c, k) = a(c, k) / d;
        inv(c, k) = inv(c, k) / d;
      }
      for (int r = 0; r < R; ++r)
        if (r != c) {
          const S f = a(r, c);
          for (int k = 0; k < C; ++k) {
            a(r, k) = a(r, k) - f * a(c, k);
            inv(r, k) = inv(r, k) - f * inv(c, k);
          }
        }
    }
    return inv;
  }
  bool isApprox(const Matrix& o, S prec = S(1e-12)) const {  // Eigen: ||a - b||^2 <= prec^2 min(||a||^2, ||b||^2)
    return (*this - o).squaredNorm() <= prec * prec * std::min(squaredNorm(), o.squaredNorm());
  }
  S trace() const {
    S t = S(0);
    for (int i = 0; i < (R < C ? R : C); ++i) t = t + (*this)(i, i);
    return t;
  }
  S mean() const { return sum() / S(N); }
  bool operator==(const Matrix& o) const {
    for (int i = 0; i < N; ++i)
      if (!(v[i] == o.v[i])) return false;
    return true;
  }
  bool operator!=(const Matrix& o) const { return !(*this == o); }
};
template <typename S, int R, int C>
Matrix<S, R, C> operator*(S s, const Matrix<S, R, C>& m) {
  return m * s;
}
template <typename S, int R, int C>
Matrix<S, R, C> Array<S, R, C>::matrix() const {
  return Matrix<S, R, C>(*this);
}
template <typename S, int R, int C>
std::ostream& operator<<(std::ostream& os, const Matrix<S, R, C>& m) {
  for (int i = 0; i < R; ++i) {
    for (int j = 0; j < C; ++j) os << (j ? " " : "") << m(i, j);
    if (i + 1 < R) os << "\n";
  }
  return os;
}

// ---- run-time sized column vector: only what VoxelHashMap.hpp's isWithinBounds<Scalar> needs (a fixed vector converts to it)
template <typename S>
class ArrayX {
 public:
  std::vector<S> v;
  ArrayX<bool> operator<=(const ArrayX& o) const {
    ArrayX<bool> r;
    for (size_t i = 0; i < v.size(); ++i) r.v.push_back(v[i] <= o.v[i]);
    return r;
  }
  bool all() const {
    for (size_t i = 0; i < v.size(); ++i)
      if (!v[i]) return false;
    return true;
  }
};
template <typename S>
class Matrix<S, Dynamic, 1> {
 public:
  typedef S Scalar;
  std::vector<S> v;
  Matrix() {}
  template <int K>
  Matrix(const Matrix<S, K, 1>& f) : v(f.v, f.v + K) {}
  ArrayX<S> array() const {
    ArrayX<S> a;
    a.v = v;
    return a;
  }
  Index size() const { return (Index)v.size(); }
  S& operator()(Index i) { return v[i]; }
  const S& operator()(Index i) const { return v[i]; }
};

// Eigen::Map<const M>(ptr): a read-only view in Eigen, a copy here (the writable Map<M> is deliberately left undefined)
template <typename T>
class Map;
template <typename S, int R, int C>
class Map<const Matrix<S, R, C>> : public Matrix<S, R, C> {
 public:
  explicit Map(const S* p) {
    for (int i = 0; i < R * C; ++i) this->v[i] = p[i];
  }
};

typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 6, 1> Vector6d;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<int, 3, 1> Vector3i;
typedef Matrix<int, 2, 1> Vector2i;
typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<double, 6, 6> Matrix6d;  // (open3d/utility/Eigen.h puts Matrix6d / Vector6d into namespace Eigen)

// ---- rotations: Hamilton quaternion (w, x, y, z), as much as math.cpp / Transform.cpp / MotionCompensation.cpp use
template <typename S>
class AngleAxis;
template <typename S>
class Quaternion {
 public:
  S w_, x_, y_, z_;
  Quaternion() : w_(1), x_(0), y_(0), z_(0) {}
  Quaternion(S w, S x, S y, S z) : w_(w), x_(x), y_(y), z_(z) {}
  explicit Quaternion(const Matrix<S, 3, 3>& m) {  // Eigen quaternionbase_assign_impl<Matrix3>: Shepperd's method
    S t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > S(0)) {
      t = std::sqrt(t + S(1.0));
      w_ = S(0.5) * t;
      t = S(0.5) / t;
      x_ = (m(2, 1) - m(1, 2)) * t;
      y_ = (m(0, 2) - m(2, 0)) * t;
      z_ = (m(1, 0) - m(0, 1)) * t;
    } else {
      int i = 0;
      if (m(1, 1) > m(0, 0)) i = 1;
      if (m(2, 2) > m(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      S q[3];
      t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + S(1.0));
      q[i] = S(0.5) * t;
      t = S(0.5) / t;
      w_ = (m(k, j) - m(j, k)) * t;
      q[j] = (m(j, i) + m(i, j)) * t;
      q[k] = (m(k, i) + m(i, k)) * t;
      x_ = q[0], y_ = q[1], z_ = q[2];
    }
  }
  Quaternion(const AngleAxis<S>& aa);
  static Quaternion Identity() { return Quaternion(); }
  S& x() { return x_; }
  S& y() { return y_; }
  S& z() { return z_; }
  S& w() { return w_; }
  const S& x() const { return x_; }
  const S& y() const { return y_; }
  const S& z() const { return z_; }
  const S& w() const { return w_; }
  S squaredNorm() const { return x_ * x_ + y_ * y_ + z_ * z_ + w_ * w_; }
  S norm() const { return std::sqrt(squaredNorm()); }
  Quaternion normalized() const {
    const S n = norm();
    return Quaternion(w_ / n, x_ / n, y_ / n, z_ / n);
  }
  void normalize() { *this = normalized(); }
  Quaternion conjugate() const { return Quaternion(w_, -x_, -y_, -z_); }
  Quaternion inverse() const {
    const S n2 = squaredNorm();
    return n2 > S(0) ? Quaternion(w_ / n2, -x_ / n2, -y_ / n2, -z_ / n2) : Quaternion(0, 0, 0, 0);
  }
  Quaternion operator*(const Quaternion& b) const {  // Eigen quat_product (scalar path)
    const Quaternion& a = *this;
    return Quaternion(a.w_ * b.w_ - a.x_ * b.x_ - a.y_ * b.y_ - a.z_ * b.z_, a.w_ * b.x_ + a.x_ * b.w_ + a.y_ * b.z_ - a.z_ * b.y_,
                      a.w_ * b.y_ + a.y_ * b.w_ + a.z_ * b.x_ - a.x_ * b.z_, a.w_ * b.z_ + a.z_ * b.w_ + a.x_ * b.y_ - a.y_ * b.x_);
  }
  Matrix<S, 3, 3> toRotationMatrix() const {  // Eigen QuaternionBase::toRotationMatrix
    Matrix<S, 3, 3> res;
    const S tx = S(2) * x_, ty = S(2) * y_, tz = S(2) * z_;
    const S twx = tx * w_, twy = ty * w_, twz = tz * w_;
    const S txx = tx * x_, txy = ty * x_, txz = tz * x_;
    const S tyy = ty * y_, tyz = tz * y_, tzz = tz * z_;
    res(0, 0) = S(1) - (tyy + tzz);
    res(0, 1) = txy - twz;
    res(0, 2) = txz + twy;
    res(1, 0) = txy + twz;
    res(1, 1) = S(1) - (txx + tzz);
    res(1, 2) = tyz - twx;
    res(2, 0) = txz - twy;
    res(2, 1) = tyz + twx;
    res(2, 2) = S(1) - (txx + tyy);
    return res;
  }
  Matrix<S, 3, 3> matrix() const { return toRotationMatrix(); }
  Matrix<S, 3, 1> operator*(const Matrix<S, 3, 1>& v) const { return toRotationMatrix() * v; }
  Quaternion slerp(S t, const Quaternion& other) const {  // Eigen QuaternionBase::slerp
    const S one = S(1) - std::numeric_limits<S>::epsilon();
    const S d = w_ * other.w_ + x_ * other.x_ + y_ * other.y_ + z_ * other.z_;
    const S absD = d < S(0) ? -d : d;
    S scale0, scale1;
    if (absD >= one) {
      scale0 = S(1) - t;
      scale1 = t;
    } else {
      const S theta = std::acos(absD);
      const S sinTheta = std::sin(theta);
      scale0 = std::sin((S(1) - t) * theta) / sinTheta;
      scale1 = std::sin((t * theta)) / sinTheta;
    }
    if (d < S(0)) scale1 = -scale1;
    return Quaternion(scale0 * w_ + scale1 * other.w_, scale0 * x_ + scale1 * other.x_, scale0 * y_ + scale1 * other.y_,
                      scale0 * z_ + scale1 * other.z_);
  }
};
typedef Quaternion<double> Quaterniond;
template <typename S>
class AngleAxis {
 public:
  S angle_;
  Matrix<S, 3, 1> axis_;
  AngleAxis() : angle_(0) {}
  AngleAxis(S angle, const Matrix<S, 3, 1>& axis) : angle_(angle), axis_(axis) {}
  S angle() const { return angle_; }
  const Matrix<S, 3, 1>& axis() const { return axis_; }
  Quaternion<S> operator*(const AngleAxis& o) const { return Quaternion<S>(*this) * Quaternion<S>(o); }
  Quaternion<S> operator*(const Quaternion<S>& o) const { return Quaternion<S>(*this) * o; }
  Matrix<S, 3, 3> toRotationMatrix() const { return Quaternion<S>(*this).toRotationMatrix(); }
};
template <typename S>
Quaternion<S> operator*(const Quaternion<S>& q, const AngleAxis<S>& a) {
  return q * Quaternion<S>(a);
}
template <typename S>
Quaternion<S>::Quaternion(const AngleAxis<S>& aa) {  // Eigen QuaternionBase::operator=(AngleAxis)
  const S ha = S(0.5) * aa.angle();
  w_ = std::cos(ha);
  const S s = std::sin(ha);
  x_ = s * aa.axis()(0), y_ = s * aa.axis()(1), z_ = s * aa.axis()(2);
}
typedef AngleAxis<double> AngleAxisd;
template <typename S, int D>
class Translation {
 public:
  Matrix<S, D, 1> t_;
  Translation() {}
  Translation(S x, S y, S z) : t_(x, y, z) {}
  explicit Translation(const Matrix<S, D, 1>& t) : t_(t) {}
};
typedef Translation<double, 3> Translation3d;

// ---- rigid / affine transform stored as linear part + translation (no projective row is ever read by the sources compiled here)
template <typename S, int D, int Mode>
class Transform {
 public:
  Matrix<S, D, D> lin_;
  Matrix<S, D, 1> t_;
  Transform() : lin_(Matrix<S, D, D>::Identity()) {}
  Transform(const Matrix<S, D + 1, D + 1>& m) { *this = m; }
  explicit Transform(const Quaternion<S>& q) : lin_(q.toRotationMatrix()) {}
  explicit Transform(const Matrix<S, D, D>& r) : lin_(r) {}
  Transform(const Translation<S, D>& t) : lin_(Matrix<S, D, D>::Identity()), t_(t.t_) {}
  static Transform Identity() { return Transform(); }
  Transform& operator=(const Matrix<S, D + 1, D + 1>& m) {
    for (int i = 0; i < D; ++i) {
      for (int j = 0; j < D; ++j) lin_(i, j) = m(i, j);
      t_(i) = m(i, D);
    }
    return *this;
  }
  const Matrix<S, D + 1, D + 1> matrix() const {
    Matrix<S, D + 1, D + 1> m = Matrix<S, D + 1, D + 1>::Identity();
    for (int i = 0; i < D; ++i) {
      for (int j = 0; j < D; ++j) m(i, j) = lin_(i, j);
      m(i, D) = t_(i);
    }
    return m;
  }
  // matrix() of a non-const transform is writable in Eigen (`T.matrix() *= M`, `T.matrix() = M`): here a copy that is written back into
  // the transform when the full expression ends
  struct MatrixAccess : Matrix<S, D + 1, D + 1> {
    Transform* owner;
    MatrixAccess(Transform* o, const Matrix<S, D + 1, D + 1>& m) : Matrix<S, D + 1, D + 1>(m), owner(o) {}
    MatrixAccess(const MatrixAccess&) = delete;
    ~MatrixAccess() { *owner = static_cast<const Matrix<S, D + 1, D + 1>&>(*this); }
    MatrixAccess& operator=(const Matrix<S, D + 1, D + 1>& m) {
      Matrix<S, D + 1, D + 1>::operator=(m);
      return *this;
    }
  };
  MatrixAccess matrix() { return MatrixAccess(this, static_cast<const Transform*>(this)->matrix()); }
  Matrix<S, D, 1>& translation() { return t_; }
  const Matrix<S, D, 1>& translation() const { return t_; }
  Matrix<S, D, D>& linear() { return lin_; }
  const Matrix<S, D, D>& linear() const { return lin_; }
  const Matrix<S, D, D> rotation() const { return lin_; }  // Isometry: the linear part IS the rotation
  Transform inverse() const {  // Isometry: [R t]^-1 = [R^T  -R^T t]
    Transform r;
    r.lin_ = lin_.transpose();
    r.t_ = -(r.lin_ * t_);
    return r;
  }
  Transform& setIdentity() { return *this = Transform(); }
  Transform operator*(const Transform& o) const {
    Transform r;
    r.lin_ = lin_ * o.lin_;
    r.t_ = lin_ * o.t_ + t_;
    return r;
  }
  Transform& operator*=(const Transform& o) { return *this = *this * o; }
  Matrix<S, D, 1> operator*(const Matrix<S, D, 1>& p) const { return lin_ * p + t_; }
  Transform operator*(const Quaternion<S>& q) const {
    Transform r = *this;
    r.lin_ = lin_ * q.toRotationMatrix();
    return r;
  }
  Transform& translate(const Matrix<S, D, 1>& d) {
    t_ = lin_ * d + t_;
    return *this;
  }
  Transform& pretranslate(const Matrix<S, D, 1>& d) {
    t_ = t_ + d;
    return *this;
  }
  Transform& rotate(const Quaternion<S>& q) {
    lin_ = lin_ * q.toRotationMatrix();
    return *this;
  }
  S operator()(Index i, Index j) const { return matrix()(i, j); }
  bool isApprox(const Transform& o, S prec = S(1e-12)) const { return matrix().isApprox(o.matrix(), prec); }
  template <typename T2>
  Transform<T2, D, Mode> cast() const {
    Transform<T2, D, Mode> r;
    r.lin_ = lin_.template cast<T2>();
    r.t_ = t_.template cast<T2>();
    return r;
  }
};
template <typename S, int D>
Transform<S, D, Isometry> operator*(const Translation<S, D>& t, const Quaternion<S>& q) {
  Transform<S, D, Isometry> r(q);
  r.t_ = t.t_;
  return r;
}
typedef Transform<double, 3, Isometry> Isometry3d;
typedef Transform<double, 3, Affine> Affine3d;
}  // namespace Eigen
