// det_math.hpp -- the arithmetic of [O3D] ComputeNormal (FastEigen3x3, Geometric Tools "robust eigensolver for 3x3 symmetric
// matrices") + NormalizeNormals + OrientNormalsTowardsCameraLocation, written so that the device produces THE SAME BITS as the CPU
// checker under oracle/ (its orc_fast_eigen3x3_min_evec, orc_acos, orc_cos, orc_estimate_normals; test infrastructure, never linked here):
//   * only +, -, *, / and sqrt of IEEE-754 binary64 (all correctly rounded on gfx950), never a fused multiply-add: everything
//     between the two pragmas below is compiled with contraction OFF (hipcc's default for device code is -ffp-contract=fast);
//   * acos / cos are the same Horner series on the same reduced arguments with the same coefficient tables as the oracle
//     (scripts/gen_det_trig.py prints both), because std::acos / std::cos differ in the last bit between math libraries and, where
//     two eigenvalues of a neighbourhood nearly coincide, that bit decides which eigenvector comes out;
//   * every expression keeps the oracle's operation order (a*b + c*d + e*f is ((a*b) + (c*d)) + (e*f)).
#pragma once
#include <hip/hip_runtime.h>

namespace o3ds {
namespace det {

#pragma clang fp contract(off)

__device__ __forceinline__ double asin_tail(double z) {  // P(z): asin(s) = s + s*z*P(z), z = s*s <= 1/4
  constexpr double c[27] = {
      0x1.5555555555555p-3, 0x1.3333333333333p-4, 0x1.6db6db6db6db7p-5, 0x1.f1c71c71c71c7p-6, 0x1.6e8ba2e8ba2e9p-6, 0x1.1c4ec4ec4ec4fp-6,
      0x1.c99999999999ap-7, 0x1.7a87878787878p-7, 0x1.3fde50d79435ep-7, 0x1.12ef3cf3cf3cfp-7, 0x1.df3bd37a6f4dfp-8, 0x1.a6863d70a3d71p-8,
      0x1.782dda12f684cp-8, 0x1.51ba308d3dcb1p-8, 0x1.31683bdef7bdfp-8, 0x1.15ee9d45d1746p-8, 0x1.fcaf8fb6db6dbp-9, 0x1.d3d2a8e0dd67dp-9,
      0x1.b026f57b13b14p-9, 0x1.90cb77f60c7cep-9, 0x1.750de64d7d05fp-9, 0x1.5c5f56efaaaabp-9, 0x1.464c0950f7d47p-9, 0x1.3275586c5f2f0p-9,
      0x1.208d3570ae5a6p-9, 0x1.1052bc5fa960ap-9, 0x1.018f963c229bfp-9};
  double p = c[26];
#pragma unroll
  for (int k = 25; k >= 0; --k) p = p * z + c[k];
  return p;
}

constexpr double kPio2Hi = 0x1.921fb54442d18p+0, kPio2Lo = 0x1.1a62633145c07p-54;
constexpr double kPiHi = 0x1.921fb54442d18p+1, kPiLo = 0x1.1a62633145c07p-53;
constexpr double kPio4 = 0x1.921fb54442d18p-1, kPi3o4 = 0x1.2d97c7f3321d2p+1;

__device__ __forceinline__ double acos_det(double x) {  // x in [-1, 1]
  if (x >= 0.5) {
    const double z = (1.0 - x) * 0.5, s = sqrt(z);
    return 2.0 * (s + s * z * asin_tail(z));
  }
  if (x <= -0.5) {
    const double z = (1.0 + x) * 0.5, s = sqrt(z);
    return (kPiHi - 2.0 * (s + s * z * asin_tail(z))) + kPiLo;
  }
  const double z = x * x;
  return (kPio2Hi - (x + x * z * asin_tail(z))) + kPio2Lo;
}

__device__ __forceinline__ double cos_series(double y) {  // |y| <= pi/4
  constexpr double c[11] = {-0x1.0000000000000p-1, 0x1.5555555555555p-5,  -0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-16,
                            -0x1.27e4fb7789f5cp-22, 0x1.1eed8eff8d898p-29, -0x1.93974a8c07c9dp-37, 0x1.ae7f3e733b81fp-45,
                            -0x1.6827863b97d97p-53, 0x1.e542ba4020225p-62, -0x1.0ce396db7f853p-70};
  const double w = y * y;
  double p = c[10];
#pragma unroll
  for (int k = 9; k >= 0; --k) p = p * w + c[k];
  return 1.0 + w * p;
}
__device__ __forceinline__ double sin_series(double y) {  // |y| <= pi/4
  constexpr double c[10] = {-0x1.5555555555555p-3,  0x1.1111111111111p-7,  -0x1.a01a01a01a01ap-13, 0x1.71de3a556c734p-19,
                            -0x1.ae64567f544e4p-26, 0x1.6124613a86d09p-33, -0x1.ae7f3e733b81fp-41, 0x1.952c77030ad4ap-49,
                            -0x1.2f49b46814157p-57, 0x1.71b8ef6dcf572p-66};
  const double w = y * y;
  double p = c[9];
#pragma unroll
  for (int k = 8; k >= 0; --k) p = p * w + c[k];
  return y + y * w * p;
}
__device__ __forceinline__ double cos_det(double x) {  // x in [0, pi]
  if (x <= kPio4) return cos_series(x);
  if (x < kPi3o4) return sin_series((kPio2Hi - x) + kPio2Lo);
  return -cos_series((kPiHi - x) + kPiLo);
}

__device__ __forceinline__ void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// A = {a00 a01 a02 a11 a12 a22}
__device__ __forceinline__ void eigvec0(const double A[6], double ev, double out[3]) {
  const double r0[3] = {A[0] - ev, A[1], A[2]}, r1[3] = {A[1], A[3] - ev, A[4]}, r2[3] = {A[2], A[4], A[5] - ev};
  double c01[3], c02[3], c12[3];
  cross3(r0, r1, c01);
  cross3(r0, r2, c02);
  cross3(r1, r2, c12);
  const double d0 = dot3(c01, c01), d1 = dot3(c02, c02), d2 = dot3(c12, c12);
  double dm = d0, b0 = c01[0], b1 = c01[1], b2 = c01[2];
  if (d1 > dm) dm = d1, b0 = c02[0], b1 = c02[1], b2 = c02[2];
  if (d2 > dm) dm = d2, b0 = c12[0], b1 = c12[1], b2 = c12[2];
  const double s = sqrt(dm);
  out[0] = b0 / s;
  out[1] = b1 / s;
  out[2] = b2 / s;
}

__device__ __forceinline__ void eigvec1(const double A[6], const double e0[3], double ev, double out[3]) {
  double U[3], V[3];
  if (fabs(e0[0]) > fabs(e0[1])) {
    const double inv = 1.0 / sqrt(e0[0] * e0[0] + e0[2] * e0[2]);
    U[0] = -e0[2] * inv;
    U[1] = 0.0;
    U[2] = e0[0] * inv;
  } else {
    const double inv = 1.0 / sqrt(e0[1] * e0[1] + e0[2] * e0[2]);
    U[0] = 0.0;
    U[1] = e0[2] * inv;
    U[2] = -e0[1] * inv;
  }
  cross3(e0, U, V);
  const double AU[3] = {A[0] * U[0] + A[1] * U[1] + A[2] * U[2], A[1] * U[0] + A[3] * U[1] + A[4] * U[2],
                        A[2] * U[0] + A[4] * U[1] + A[5] * U[2]};
  const double AV[3] = {A[0] * V[0] + A[1] * V[1] + A[2] * V[2], A[1] * V[0] + A[3] * V[1] + A[4] * V[2],
                        A[2] * V[0] + A[4] * V[1] + A[5] * V[2]};
  double m00 = dot3(U, AU) - ev, m01 = dot3(U, AV), m11 = dot3(V, AV) - ev;
  const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
  if (a00 >= a11) {
    const double mx = a00 > a01 ? a00 : a01;
    if (mx > 0.0) {
      if (a00 >= a01) {
        m01 /= m00;
        m00 = 1.0 / sqrt(1.0 + m01 * m01);
        m01 *= m00;
      } else {
        m00 /= m01;
        m01 = 1.0 / sqrt(1.0 + m00 * m00);
        m00 *= m01;
      }
      for (int i = 0; i < 3; ++i) out[i] = m01 * U[i] - m00 * V[i];
    } else {
      for (int i = 0; i < 3; ++i) out[i] = U[i];
    }
  } else {
    const double mx = a11 > a01 ? a11 : a01;
    if (mx > 0.0) {
      if (a11 >= a01) {
        m01 /= m11;
        m11 = 1.0 / sqrt(1.0 + m01 * m01);
        m01 *= m11;
      } else {
        m11 /= m01;
        m01 = 1.0 / sqrt(1.0 + m11 * m11);
        m11 *= m01;
      }
      for (int i = 0; i < 3; ++i) out[i] = m11 * U[i] - m01 * V[i];
    } else {
      for (int i = 0; i < 3; ++i) out[i] = U[i];
    }
  }
}

// cov = {c00 c01 c02 c11 c12 c22}; eigenvector of the smallest eigenvalue (not normalised where it is a cross product)
__device__ __forceinline__ void fast_eigen3x3_min(const double cov[6], double out[3]) {
  double mc = cov[0];
  for (int i = 1; i < 6; ++i)
    if (cov[i] > mc) mc = cov[i];
  if (mc == 0.0) {
    out[0] = out[1] = out[2] = 0.0;
    return;
  }
  double A[6];
  for (int i = 0; i < 6; ++i) A[i] = cov[i] / mc;
  const double norm = A[1] * A[1] + A[2] * A[2] + A[4] * A[4];
  if (norm > 0.0) {
    const double q = (A[0] + A[3] + A[5]) / 3.0;
    const double b00 = A[0] - q, b11 = A[3] - q, b22 = A[5] - q;
    const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2.0) / 6.0);
    const double c00 = b11 * b22 - A[4] * A[4];
    const double c01 = A[1] * b22 - A[4] * A[2];
    const double c02 = A[1] * A[4] - b11 * A[2];
    const double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
    double half_det = det * 0.5;
    if (half_det < -1.0) half_det = -1.0;
    if (half_det > 1.0) half_det = 1.0;
    const double angle = acos_det(half_det) / 3.0;
    const double two_thirds_pi = 2.09439510239319549;
    const double beta2 = cos_det(angle) * 2.0;
    const double beta0 = cos_det(angle + two_thirds_pi) * 2.0;
    const double beta1 = -(beta0 + beta2);
    const double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;
    double v0[3], v1[3], v2[3];
    if (half_det >= 0.0) {
      eigvec0(A, e2, v2);
      if (e2 < e0 && e2 < e1) {
        out[0] = v2[0], out[1] = v2[1], out[2] = v2[2];
        return;
      }
      eigvec1(A, v2, e1, v1);
      if (e1 < e0 && e1 < e2) {
        out[0] = v1[0], out[1] = v1[1], out[2] = v1[2];
        return;
      }
      cross3(v1, v2, out);
    } else {
      eigvec0(A, e0, v0);
      if (e0 < e1 && e0 < e2) {
        out[0] = v0[0], out[1] = v0[1], out[2] = v0[2];
        return;
      }
      eigvec1(A, v0, e1, v1);
      if (e1 < e0 && e1 < e2) {
        out[0] = v1[0], out[1] = v1[1], out[2] = v1[2];
        return;
      }
      cross3(v0, v1, out);
    }
  } else {  // diagonal
    if (cov[0] < cov[3] && cov[0] < cov[5]) {
      out[0] = 1, out[1] = 0, out[2] = 0;
    } else if (cov[3] < cov[0] && cov[3] < cov[5]) {
      out[0] = 0, out[1] = 1, out[2] = 0;
    } else {
      out[0] = 0, out[1] = 0, out[2] = 1;
    }
  }
}

// population covariance from the nine cumulant sums {x y z xx xy xz yy yz zz} of k points ([O3D] EstimatePerPointCovariances)
__device__ __forceinline__ void cov_from_cumulants(const double s[9], int k, double cov[6]) {
  double c[9];
  for (int j = 0; j < 9; ++j) c[j] = s[j] / (double)k;
  cov[0] = c[3] - c[0] * c[0];
  cov[1] = c[4] - c[0] * c[1];
  cov[2] = c[5] - c[0] * c[2];
  cov[3] = c[6] - c[1] * c[1];
  cov[4] = c[7] - c[1] * c[2];
  cov[5] = c[8] - c[2] * c[2];
}

// ComputeNormal's result -> NormalizeNormals -> OrientNormalsTowardsCameraLocation(0,0,0), for the point (px,py,pz)
// raw: [O3D] EstimateNormals alone (a zero vector becomes (0,0,1), nothing else) -- what InitializePointCloudForGeneralizedICP uses
__device__ __forceinline__ void normalize_orient(double nv[3], double px, double py, double pz, bool raw = false) {
  double nn = sqrt(dot3(nv, nv));
  if (nn == 0.0) {
    nv[0] = 0, nv[1] = 0, nv[2] = 1;
    nn = 1.0;
  }
  if (raw) return;
  nv[0] /= nn, nv[1] /= nn, nv[2] /= nn;
  if (nv[0] != nv[0]) nv[0] = 0, nv[1] = 0, nv[2] = 1;
  const double ref[3] = {-px, -py, -pz};
  if (dot3(nv, nv) == 0.0) {
    const double rn = sqrt(dot3(ref, ref));
    if (rn == 0.0) {
      nv[0] = 0, nv[1] = 0, nv[2] = 1;
    } else {
      nv[0] = ref[0] / rn, nv[1] = ref[1] / rn, nv[2] = ref[2] / rn;
    }
  } else if (dot3(nv, ref) < 0.0) {
    nv[0] = -nv[0], nv[1] = -nv[1], nv[2] = -nv[2];
  }
}

#pragma clang fp contract(fast)

}  // namespace det
}  // namespace o3ds
