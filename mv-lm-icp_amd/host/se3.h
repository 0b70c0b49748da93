// Host-side SE(3) helpers for the LM solve: pose <-> parameter conversions, the three local
// parameterizations' (+) operators and their maps into the canonical perturbation coordinates the HIP
// linearize kernel works in.  Plain fp64 C++ (no Eigen/Ceres/Sophus in this image).
//
// Mirrors, per parameterization (reference file:line):
//   EIGEN_QUATERNION  x = [qx qy qz qw | tx ty tz]  Quaterniond(pose.linear()) icp-ceres.cpp:237;
//                     (+): q <- [sin|d| d/|d|, cos|d|] * q, t <- t + dt      eigen_quaternion.h:89-106
//   ANGLE_AXIS        x = [w | t]                   isoToAngleAxis icp-ceres.cpp:97-107; (+): x + d (:329, no local param)
//   SOPHUS_SE3        x = [qx qy qz qw | tx ty tz]  Sophus::SE3d(pose) icp-ceres.cpp:121;
//                     (+): T <- T exp(d), d = (upsilon, omega)               sophus_se3.h:31-38
#pragma once
#include <cmath>
#include <limits>

#include "../../include/mvicp.h"

namespace mvicp {
namespace se3 {

inline int ambient(int param) { return param == MVICP_PARAM_ANGLE_AXIS ? 6 : 7; }

// ---- 3x3 column-major helpers ---------------------------------------------------------------
inline void pose_R(const double* P, double* R) {
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) R[i + 3 * j] = P[i + 4 * j];
}
inline void set_pose(const double* R, const double* t, double* P) {
  for (int i = 0; i < 16; ++i) P[i] = 0.0;
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) P[i + 4 * j] = R[i + 3 * j];
  P[12] = t[0]; P[13] = t[1]; P[14] = t[2]; P[15] = 1.0;
}

// unit quaternion [x y z w] -> rotation (Eigen QuaternionBase::toRotationMatrix)
inline void quat_to_R(const double* q, double* R) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[3] = txy - twz;       R[6] = txz + twy;
  R[1] = txy + twz;       R[4] = 1 - (txx + tzz); R[7] = tyz - twx;
  R[2] = txz - twy;       R[5] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
// rotation -> quaternion [x y z w] (Eigen's Shepperd-style branchy conversion)
inline void R_to_quat(const double* R, double* q) {
  auto m = [&](int i, int j) { return R[i + 3 * j]; };
  double t = m(0, 0) + m(1, 1) + m(2, 2);
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m(2, 1) - m(1, 2)) * t;
    q[1] = (m(0, 2) - m(2, 0)) * t;
    q[2] = (m(1, 0) - m(0, 1)) * t;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m(k, j) - m(j, k)) * t;
    q[j] = (m(j, i) + m(i, j)) * t;
    q[k] = (m(k, i) + m(i, k)) * t;
  }
}
inline void quat_mul(const double* a, const double* b, double* r) {  // [x y z w]
  const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z; r[3] = w;
}
inline void quat_normalize(double* q) {
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

// angle-axis <-> rotation (ceres/rotation.h AngleAxisToRotationMatrix / RotationMatrixToAngleAxis [upstream])
inline void aa_to_R(const double* w, double* R) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > std::numeric_limits<double>::epsilon()) {
    const double th = std::sqrt(th2), x = w[0] / th, y = w[1] / th, z = w[2] / th;
    const double c = std::cos(th), s = std::sin(th), k = 1.0 - c;
    R[0] = c + x * x * k;     R[3] = x * y * k - z * s; R[6] = y * s + x * z * k;
    R[1] = z * s + x * y * k; R[4] = c + y * y * k;     R[7] = -x * s + y * z * k;
    R[2] = -y * s + x * z * k; R[5] = x * s + y * z * k; R[8] = c + z * z * k;
  } else {
    R[0] = 1;     R[3] = -w[2]; R[6] = w[1];
    R[1] = w[2];  R[4] = 1;     R[7] = -w[0];
    R[2] = -w[1]; R[5] = w[0];  R[8] = 1;
  }
}
inline void R_to_aa(const double* R, double* w) {
  auto m = [&](int i, int j) { return R[i + 3 * j]; };
  double q[4];  // [w x y z]
  const double tr = m(0, 0) + m(1, 1) + m(2, 2);
  if (tr >= 0.0) {
    double t = std::sqrt(tr + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (m(2, 1) - m(1, 2)) * t; q[2] = (m(0, 2) - m(2, 0)) * t; q[3] = (m(1, 0) - m(0, 1)) * t;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m(k, j) - m(j, k)) * t; q[j + 1] = (m(j, i) + m(i, j)) * t; q[k + 1] = (m(k, i) + m(i, k)) * t;
  }
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  double k = 2.0;
  if (s2 > 0.0) {
    const double s = std::sqrt(s2), c = q[0];
    const double two_theta = 2.0 * (c < 0.0 ? std::atan2(-s, -c) : std::atan2(s, c));
    k = two_theta / s;
  }
  w[0] = q[1] * k; w[1] = q[2] * k; w[2] = q[3] * k;
}

// ---- pose <-> ambient parameters ------------------------------------------------------------
inline void pose_to_x(int param, const double* P, double* x) {
  double R[9];
  pose_R(P, R);
  if (param == MVICP_PARAM_ANGLE_AXIS) {
    R_to_aa(R, x);
    x[3] = P[12]; x[4] = P[13]; x[5] = P[14];
  } else {
    R_to_quat(R, x);
    if (param == MVICP_PARAM_SOPHUS_SE3) quat_normalize(x);
    x[4] = P[12]; x[5] = P[13]; x[6] = P[14];
  }
}
inline void x_to_pose(int param, const double* x, double* P) {
  double R[9];
  if (param == MVICP_PARAM_ANGLE_AXIS) { aa_to_R(x, R); set_pose(R, x + 3, P); }
  else { quat_to_R(x, R); set_pose(R, x + 4, P); }
}

// so(3) series coefficients: A = (1-cos)/th^2, B = (th-sin)/th^3
inline void so3_coeffs(double th2, double* A, double* B) {
  if (th2 < 1e-6) {
    *A = 0.5 - th2 / 24.0 + th2 * th2 / 720.0;
    *B = 1.0 / 6.0 - th2 / 120.0 + th2 * th2 / 5040.0;
  } else {
    const double th = std::sqrt(th2);
    *A = (1.0 - std::cos(th)) / th2;
    *B = (th - std::sin(th)) / (th2 * th);
  }
}

// ---- (+) ------------------------------------------------------------------------------------
inline void plus(int param, const double* x, const double* d, double* out) {
  if (param == MVICP_PARAM_ANGLE_AXIS) { for (int i = 0; i < 6; ++i) out[i] = x[i] + d[i]; return; }
  if (param == MVICP_PARAM_EIGEN_QUATERNION) {
    const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nd > 0.0) {
      const double s = std::sin(nd) / nd;
      const double dq[4] = {s * d[0], s * d[1], s * d[2], std::cos(nd)};
      quat_mul(dq, x, out);
    } else {
      for (int i = 0; i < 4; ++i) out[i] = x[i];
    }
    out[4] = x[4] + d[3]; out[5] = x[5] + d[4]; out[6] = x[6] + d[5];
    return;
  }
  // T exp(d): q <- q * Exp(omega); t <- t + R(q) V(omega) upsilon
  const double* u = d;
  const double* w = d + 3;
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double imag, real;
  if (th2 < 1e-20) {
    imag = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0;
    real = 1.0 - 0.5 * th2 + th2 * th2 / 384.0;
  } else {
    const double th = std::sqrt(th2);
    imag = std::sin(0.5 * th) / th;
    real = std::cos(0.5 * th);
  }
  const double dq[4] = {imag * w[0], imag * w[1], imag * w[2], real};
  double A, B;
  so3_coeffs(th2, &A, &B);
  // V u = u + A (w x u) + B (w x (w x u))
  const double wu[3] = {w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]};
  const double wwu[3] = {w[1] * wu[2] - w[2] * wu[1], w[2] * wu[0] - w[0] * wu[2], w[0] * wu[1] - w[1] * wu[0]};
  const double vu[3] = {u[0] + A * wu[0] + B * wwu[0], u[1] + A * wu[1] + B * wwu[1], u[2] + A * wu[2] + B * wwu[2]};
  double R[9];
  quat_to_R(x, R);
  quat_mul(x, dq, out);
  quat_normalize(out);
  for (int i = 0; i < 3; ++i) out[4 + i] = x[4 + i] + R[i] * vu[0] + R[i + 3] * vu[1] + R[i + 6] * vu[2];
}

// ---- local -> canonical map: [upsilon; omega] = M * local, M row-major 6x6 -----------------------
// canonical = right perturbation T <- T exp([upsilon, omega]) (what the linearize kernel differentiates).
inline void local_to_canonical(int param, const double* x, double* M) {
  for (int i = 0; i < 36; ++i) M[i] = 0.0;
  if (param == MVICP_PARAM_SOPHUS_SE3) { for (int i = 0; i < 6; ++i) M[i * 6 + i] = 1.0; return; }
  double R[9];
  if (param == MVICP_PARAM_ANGLE_AXIS) aa_to_R(x, R); else quat_to_R(x, R);
  // translation block is additive in the world frame: t <- t + dt  =>  upsilon = R^T dt   (local cols 3..5)
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i * 6 + 3 + j] = R[j + 3 * i];
  if (param == MVICP_PARAM_EIGEN_QUATERNION) {
    // q <- dq * q is a LEFT rotation by angle 2|d|:  R <- Exp(2 d) R = R Exp(2 R^T d)  =>  omega = 2 R^T dtheta
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[(3 + i) * 6 + j] = 2.0 * R[j + 3 * i];
  } else {
    // additive angle-axis: R(w + dw) = R(w) Exp(Jr(w) dw), Jr = I - A [w]x + B [w]x^2
    const double* w = x;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double Jr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major
    if (th2 > std::numeric_limits<double>::epsilon()) {  // below: AngleAxisRotatePoint's first-order branch, d/dw = -[p]x
      double A, B;
      so3_coeffs(th2, &A, &B);
      const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};  // row-major [w]x
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double w2 = w[i] * w[j];
          if (i == j) w2 -= th2;
          Jr[i * 3 + j] += -A * W[i * 3 + j] + B * w2;
        }
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[(3 + i) * 6 + j] = Jr[i * 3 + j];
  }
}

}  // namespace se3
}  // namespace mvicp
