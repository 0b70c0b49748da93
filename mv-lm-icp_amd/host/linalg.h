// Minimal fp64 stand-ins for the Eigen types the reference's Frame / driver code uses (Eigen is not installed in
// this image): Vector3d and Isometry3d with the same storage (3 contiguous doubles; 4x4 COLUMN-major) and the
// handful of members the reference touches (include/frame.h:38-43, src/main_multiview.cpp, include/common.h).
#pragma once
#include <cmath>
#include <cstring>

namespace mvicp {

struct Vector3d {
  double v[3];
  Vector3d() : v{0, 0, 0} {}
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  double& x() { return v[0]; } double& y() { return v[1]; } double& z() { return v[2]; }
  double x() const { return v[0]; } double y() const { return v[1]; } double z() const { return v[2]; }
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double* data() { return v; }
  const double* data() const { return v; }
  Vector3d operator+(const Vector3d& o) const { return Vector3d(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
  Vector3d operator-(const Vector3d& o) const { return Vector3d(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
  Vector3d operator*(double s) const { return Vector3d(v[0] * s, v[1] * s, v[2] * s); }
  double dot(const Vector3d& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
  double norm() const { return std::sqrt(dot(*this)); }
};
static_assert(sizeof(Vector3d) == 24, "Vector3d must be 3 packed doubles (passed to the C ABI as xyz AoS)");

struct Isometry3d {
  double m[16];  // column-major 4x4, like Eigen::Isometry3d::data()
  Isometry3d() { setIdentity(); }
  static Isometry3d Identity() { return Isometry3d(); }
  void setIdentity() { std::memset(m, 0, sizeof(m)); m[0] = m[5] = m[10] = m[15] = 1.0; }
  double* data() { return m; }
  const double* data() const { return m; }
  double& operator()(int r, int c) { return m[r + 4 * c]; }
  double operator()(int r, int c) const { return m[r + 4 * c]; }
  Vector3d translation() const { return Vector3d(m[12], m[13], m[14]); }
  void setTranslation(const Vector3d& t) { m[12] = t[0]; m[13] = t[1]; m[14] = t[2]; }
  Vector3d operator*(const Vector3d& p) const {  // linear() * p + translation()
    return Vector3d((m[0] * p[0] + m[4] * p[1]) + m[8] * p[2] + m[12], (m[1] * p[0] + m[5] * p[1]) + m[9] * p[2] + m[13],
                    (m[2] * p[0] + m[6] * p[1]) + m[10] * p[2] + m[14]);
  }
  Vector3d rotate(const Vector3d& p) const {
    return Vector3d(m[0] * p[0] + m[4] * p[1] + m[8] * p[2], m[1] * p[0] + m[5] * p[1] + m[9] * p[2], m[2] * p[0] + m[6] * p[1] + m[10] * p[2]);
  }
  Isometry3d operator*(const Isometry3d& o) const {
    Isometry3d r;
    for (int c = 0; c < 4; ++c)
      for (int i = 0; i < 4; ++i) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += m[i + 4 * k] * o.m[k + 4 * c];
        r.m[i + 4 * c] = s;
      }
    return r;
  }
  Isometry3d inverse() const {  // rigid inverse
    Isometry3d r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i + 4 * j] = m[j + 4 * i];
    for (int i = 0; i < 3; ++i) r.m[12 + i] = -(r.m[i] * m[12] + r.m[i + 4] * m[13] + r.m[i + 8] * m[14]);
    return r;
  }
};

}  // namespace mvicp
