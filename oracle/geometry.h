// TEST INFRASTRUCTURE — CPU oracle for the mv-lm-icp hot path. NOT part of the shipped product.
//
// Templated (double or orc::Jet) restatements of the rotation / SE(3) helpers the reference pulls in
// from its third-party dependencies, none of which are under /root/reference:
//   * Ceres Solver (< 2.2, README.md:48-52 targets Ubuntu 18.04 libceres-dev 1.13/1.14), ceres/rotation.h:
//       AngleAxisRotatePoint      — used at icp-ceres.h:162,165,207,210,213
//       AngleAxisToRotationMatrix — used at icp-ceres.cpp:111
//       RotationMatrixToAngleAxis — used at icp-ceres.cpp:101 (through the stride-4 ColumnMajorAdapter4x3)
//   * Eigen3 Quaternion: construction from a rotation matrix (icp-ceres.cpp:237), toRotationMatrix
//     (icp-ceres.h:130,132; icp-ceres.cpp:116), quaternion*vector (icp-ceres.h:81,83), quaternion product
//     (eigen_quaternion.h:99)
//   * Sophus (stevenlovegrove/Sophus fork, SHA unpinned: .gitmodules:1-3): SE3Group::exp, SO3Group::exp,
//     group product (sophus_se3.h:16,36), SE3d(Isometry) (icp-ceres.cpp:121), rotationMatrix() (:127)
// Published algorithms restated from the upstream documentation/recollection ("[upstream]").
#pragma once
#include <limits>
#include "jet.h"

namespace orc {

// ---------------------------------------------------------------- small helpers
template <typename T> inline void cross3(const T* a, const T* b, T* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename T> inline T dot3(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// ---------------------------------------------------------------- Ceres rotation.h [upstream]
// AngleAxisRotatePoint: Rodrigues for theta^2 > eps, first-order `p + w x p` otherwise.
template <typename T> inline void AngleAxisRotatePoint(const T* aa, const T* pt, T* result) {
  const T theta2 = dot3(aa, aa);
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    const T theta = sqrt(theta2);
    const T costheta = cos(theta);
    const T sintheta = sin(theta);
    const T theta_inverse = 1.0 / theta;
    const T w[3] = {aa[0] * theta_inverse, aa[1] * theta_inverse, aa[2] * theta_inverse};
    const T w_cross_pt[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (1.0 - costheta);
    const T r0 = pt[0] * costheta + w_cross_pt[0] * sintheta + w[0] * tmp;
    const T r1 = pt[1] * costheta + w_cross_pt[1] * sintheta + w[1] * tmp;
    const T r2 = pt[2] * costheta + w_cross_pt[2] * sintheta + w[2] * tmp;
    result[0] = r0; result[1] = r1; result[2] = r2;
  } else {
    const T w_cross_pt[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2], aa[0] * pt[1] - aa[1] * pt[0]};
    const T r0 = pt[0] + w_cross_pt[0];
    const T r1 = pt[1] + w_cross_pt[1];
    const T r2 = pt[2] + w_cross_pt[2];
    result[0] = r0; result[1] = r1; result[2] = r2;
  }
}

// AngleAxisToRotationMatrix into a column-major 3x3 (R[i + 3 j]).
inline void AngleAxisToRotationMatrix(const double* aa, double* R) {
  const double theta2 = dot3(aa, aa);
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    const double theta = std::sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double c = std::cos(theta), s = std::sin(theta);
    R[0] = c + wx * wx * (1.0 - c);
    R[1] = wz * s + wx * wy * (1.0 - c);
    R[2] = -wy * s + wx * wz * (1.0 - c);
    R[3] = wx * wy * (1.0 - c) - wz * s;
    R[4] = c + wy * wy * (1.0 - c);
    R[5] = wx * s + wy * wz * (1.0 - c);
    R[6] = wy * s + wx * wz * (1.0 - c);
    R[7] = -wx * s + wy * wz * (1.0 - c);
    R[8] = c + wz * wz * (1.0 - c);
  } else {
    R[0] = 1.0;    R[1] = aa[2];  R[2] = -aa[1];
    R[3] = -aa[2]; R[4] = 1.0;    R[5] = aa[0];
    R[6] = aa[1];  R[7] = -aa[0]; R[8] = 1.0;
  }
}

// RotationMatrixToAngleAxis: RotationMatrixToQuaternion (w first) then QuaternionToAngleAxis.
// R is column-major 3x3.
inline void RotationMatrixToAngleAxis(const double* R, double* aa) {
#define RM(i, j) R[(i) + 3 * (j)]
  double q[4];
  const double trace = RM(0, 0) + RM(1, 1) + RM(2, 2);
  if (trace >= 0.0) {
    double t = std::sqrt(trace + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (RM(2, 1) - RM(1, 2)) * t;
    q[2] = (RM(0, 2) - RM(2, 0)) * t;
    q[3] = (RM(1, 0) - RM(0, 1)) * t;
  } else {
    int i = 0;
    if (RM(1, 1) > RM(0, 0)) i = 1;
    if (RM(2, 2) > RM(i, i)) i = 2;
    const int j = (i + 1) % 3;
    const int k = (j + 1) % 3;
    double t = std::sqrt(RM(i, i) - RM(j, j) - RM(k, k) + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (RM(k, j) - RM(j, k)) * t;
    q[j + 1] = (RM(j, i) + RM(i, j)) * t;
    q[k + 1] = (RM(k, i) + RM(i, k)) * t;
  }
#undef RM
  const double sin_squared_theta = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (sin_squared_theta > 0.0) {
    const double sin_theta = std::sqrt(sin_squared_theta);
    const double cos_theta = q[0];
    const double two_theta = 2.0 * ((cos_theta < 0.0) ? std::atan2(-sin_theta, -cos_theta) : std::atan2(sin_theta, cos_theta));
    const double k = two_theta / sin_theta;
    aa[0] = q[1] * k; aa[1] = q[2] * k; aa[2] = q[3] * k;
  } else {
    aa[0] = q[1] * 2.0; aa[1] = q[2] * 2.0; aa[2] = q[3] * 2.0;
  }
}

// ---------------------------------------------------------------- Eigen::Quaternion [upstream]; storage [x,y,z,w]
// Quaterniond(Matrix3d) — Eigen/src/Geometry/Quaternion.h quaternionbase_assign_impl<Other,3,3>. R column-major.
inline void EigenQuatFromRotation(const double* R, double* q) {
#define RM(i, j) R[(i) + 3 * (j)]
  double t = RM(0, 0) + RM(1, 1) + RM(2, 2);
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (RM(2, 1) - RM(1, 2)) * t;
    q[1] = (RM(0, 2) - RM(2, 0)) * t;
    q[2] = (RM(1, 0) - RM(0, 1)) * t;
  } else {
    int i = 0;
    if (RM(1, 1) > RM(0, 0)) i = 1;
    if (RM(2, 2) > RM(i, i)) i = 2;
    const int j = (i + 1) % 3;
    const int k = (j + 1) % 3;
    t = std::sqrt(RM(i, i) - RM(j, j) - RM(k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (RM(k, j) - RM(j, k)) * t;
    q[j] = (RM(j, i) + RM(i, j)) * t;
    q[k] = (RM(k, i) + RM(i, k)) * t;
  }
#undef RM
}

// QuaternionBase::toRotationMatrix (assumes unit q); column-major output.
template <typename T> inline void EigenQuatToRotation(const T* q, T* R) {
  const T tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
  const T twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const T txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const T tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1.0 - (tyy + tzz); R[3] = txy - twz;         R[6] = txz + twy;
  R[1] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[7] = tyz - twx;
  R[2] = txz - twy;         R[5] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// QuaternionBase::_transformVector: v + w*(2 u x v) + u x (2 u x v), u = q.vec().
template <typename T> inline void EigenQuatRotate(const T* q, const T* v, T* out) {
  T uv[3];
  cross3(q, v, uv);
  uv[0] = uv[0] + uv[0]; uv[1] = uv[1] + uv[1]; uv[2] = uv[2] + uv[2];
  T uuv[3];
  cross3(q, uv, uuv);
  out[0] = v[0] + q[3] * uv[0] + uuv[0];
  out[1] = v[1] + q[3] * uv[1] + uuv[1];
  out[2] = v[2] + q[3] * uv[2] + uuv[2];
}

// Quaternion product a*b (Eigen quat_product), storage [x,y,z,w].
template <typename T> inline void EigenQuatProduct(const T* a, const T* b, T* r) {
  const T w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  const T x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const T y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  const T z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z; r[3] = w;
}

// ---------------------------------------------------------------- Sophus SE3/SO3 [upstream]; data = [qx qy qz qw tx ty tz]
// SE3Group::exp(a), a = (upsilon, omega)  ->  (q_delta, t_delta).  Small-angle branches as in Sophus
// (SophusConstants<double>::epsilon() = 1e-10).
template <typename T> inline void SophusSE3Exp(const T* a, T* qd, T* td) {
  const T* ups = a;
  const T* om = a + 3;
  const T theta_sq = dot3(om, om);
  T imag_factor, real_factor;
  const bool small = value_of(theta_sq) < 1e-20;  // theta < 1e-10
  T theta = T(0.0);
  if (small) {
    const T theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real_factor = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    theta = sqrt(theta_sq);
    const T half_theta = 0.5 * theta;
    imag_factor = sin(half_theta) / theta;
    real_factor = cos(half_theta);
  }
  qd[0] = imag_factor * om[0]; qd[1] = imag_factor * om[1]; qd[2] = imag_factor * om[2]; qd[3] = real_factor;
  // V = I + (1-cos)/theta^2 Omega + (theta-sin)/theta^3 Omega^2   (V = so3.matrix() for theta < eps)
  T V[9];
  if (small) {
    EigenQuatToRotation(qd, V);
  } else {
    const T A = (1.0 - cos(theta)) / theta_sq;
    const T B = (theta - sin(theta)) / (theta_sq * theta);
    // Omega = hat(om); Omega^2 = om om^T - theta_sq I
    const T O[9] = {T(0.0), om[2], -om[1], -om[2], T(0.0), om[0], om[1], -om[0], T(0.0)};  // column-major
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) {
        T o2 = om[r] * om[c];
        if (r == c) o2 = o2 - theta_sq;
        V[r + 3 * c] = A * O[r + 3 * c] + B * o2;
        if (r == c) V[r + 3 * c] = V[r + 3 * c] + 1.0;
      }
  }
  for (int r = 0; r < 3; ++r) td[r] = V[r] * ups[0] + V[r + 3] * ups[1] + V[r + 6] * ups[2];
}

// x_plus_delta = x * exp(delta)   (sophus_se3.h:16 / :36)
template <typename T> inline void SophusSE3Plus(const T* x, const T* delta, T* out) {
  T qd[4], td[3];
  SophusSE3Exp(delta, qd, td);
  T q[4];
  EigenQuatProduct(x, qd, q);
  T rt[3];
  EigenQuatRotate(x, td, rt);
  out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
  out[4] = x[4] + rt[0]; out[5] = x[5] + rt[1]; out[6] = x[6] + rt[2];
}

}  // namespace orc
