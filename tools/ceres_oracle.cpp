// tools/ceres_oracle.cpp — OPTIONAL checker, built only where Ceres Solver + Eigen3 are installed (tools/build_ceres_oracle.sh; neither this image
// nor the GPU box has them: tools/ceres_probe.py).  TEST INFRASTRUCTURE: nothing in the product path links or runs it.
//
// One multiview LM solve through the REAL Ceres with the reference's options, so that the part of the LM the reference's only published
// vector does not reach — SoftLOneLoss(edge.weight), several poses, point-to-plane, the function-tolerance stop, the SE(3) local parameterization
// — can be pinned wherever Ceres exists.  Own code: the reference's functors are RESTATED against ceres::AutoDiffCostFunction
// (no Sophus needed: the 7-parameter block is Sophus' storage [qx qy qz qw | tx ty tz], rotated with Eigen's quaternion-times-vector formula):
//   options                 src/internal/icp-ceres.cpp:66-89 (getOptionsMedium: SPARSE_NORMAL_CHOLESKY, use_explicit_schur_complement, 50 iterations)
//   problem construction    icp-ceres.cpp:325-395 (angle-axis, param 1), :398-475 (SophusSE3, param 2): one residual block per correspondence of every
//                           edge whose SOURCE is not fixed, SoftLOneLoss(weight) if robust, frame 0 fixed, fixed frames SetParameterBlockConstant
//   functors                include/icp-ceres.h:143-234 (angle-axis: ceres::AngleAxisRotatePoint), :236-316 (SophusSE3: q * p + t)
//   local parameterization  include/sophus_se3.h:10-19,64-73: AutoDiffLocalParameterization<SophusSE3Plus, 7, 6>, x * exp(delta), delta = (upsilon, omega)
//                           (the reference's default, automaticDiffLocalParam = true); none for angle-axis (icp-ceres.cpp:329,376)
//   pose <-> parameters     icp-ceres.cpp:97-134
// Input (binary, little endian; written by tests/test_ceres_oracle.py):  "MVCERES1" | int32 K, E, param, plane, robust | per frame: int32 n, fixed;
//   n x 3 pts; n x 3 nor; 16 pose (column-major 4x4) | per edge: int32 src, dst, C; float64 weight; C x int32 first; C x int32 second.
// Output (binary): K x 16 poses (column-major) | int32 iterations, termination_type | float64 initial_cost, final_cost.
#include <ceres/ceres.h>
#include <ceres/rotation.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

struct FrameIn { int n = 0, fixed = 0; std::vector<double> pts, nor; double pose[16]; };
struct EdgeIn { int src = 0, dst = 0; double weight = 0; std::vector<int> first, second; };

// Eigen's Quaternion * Vector3:  uv = 2 u x v;  v + w uv + u x uv   (q = [x y z w])
template <typename T>
inline void quat_rotate(const T* q, const T* v, T* out) {
  const T uv0 = T(2) * (q[1] * v[2] - q[2] * v[1]), uv1 = T(2) * (q[2] * v[0] - q[0] * v[2]), uv2 = T(2) * (q[0] * v[1] - q[1] * v[0]);
  out[0] = v[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
  out[1] = v[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
  out[2] = v[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
}

struct PlaneSE3 {   // icp-ceres.h:276-316
  const double *d, *s, *n;
  PlaneSE3(const double* dst, const double* src, const double* nor) : d(dst), s(src), n(nor) {}
  template <typename T> bool operator()(const T* cam1, const T* cam2, T* r) const {
    const T src[3] = {T(s[0]), T(s[1]), T(s[2])}, dst[3] = {T(d[0]), T(d[1]), T(d[2])}, nor[3] = {T(n[0]), T(n[1]), T(n[2])};
    T p[3], p2[3], n2[3];
    quat_rotate(cam1, src, p); quat_rotate(cam2, dst, p2); quat_rotate(cam2, nor, n2);
    for (int i = 0; i < 3; ++i) { p[i] += cam1[4 + i]; p2[i] += cam2[4 + i]; }
    r[0] = (p[0] - p2[0]) * n2[0] + (p[1] - p2[1]) * n2[1] + (p[2] - p2[2]) * n2[2];
    return true;
  }
};
struct PointSE3 {   // icp-ceres.h:236-274
  const double *d, *s;
  PointSE3(const double* dst, const double* src) : d(dst), s(src) {}
  template <typename T> bool operator()(const T* cam1, const T* cam2, T* r) const {
    const T src[3] = {T(s[0]), T(s[1]), T(s[2])}, dst[3] = {T(d[0]), T(d[1]), T(d[2])};
    T p[3], p2[3];
    quat_rotate(cam1, src, p); quat_rotate(cam2, dst, p2);
    for (int i = 0; i < 3; ++i) r[i] = (p[i] + cam1[4 + i]) - (p2[i] + cam2[4 + i]);
    return true;
  }
};
struct PlaneAA {    // icp-ceres.h:185-234: cam = [angle-axis | t]
  const double *d, *s, *n;
  PlaneAA(const double* dst, const double* src, const double* nor) : d(dst), s(src), n(nor) {}
  template <typename T> bool operator()(const T* cam1, const T* cam2, T* r) const {
    const T src[3] = {T(s[0]), T(s[1]), T(s[2])}, dst[3] = {T(d[0]), T(d[1]), T(d[2])}, nor[3] = {T(n[0]), T(n[1]), T(n[2])};
    T p[3], p2[3], n2[3];
    ceres::AngleAxisRotatePoint(cam1, src, p); ceres::AngleAxisRotatePoint(cam2, dst, p2); ceres::AngleAxisRotatePoint(cam2, nor, n2);
    for (int i = 0; i < 3; ++i) { p[i] += cam1[3 + i]; p2[i] += cam2[3 + i]; }
    r[0] = (p[0] - p2[0]) * n2[0] + (p[1] - p2[1]) * n2[1] + (p[2] - p2[2]) * n2[2];
    return true;
  }
};
struct PointAA {    // icp-ceres.h:143-183
  const double *d, *s;
  PointAA(const double* dst, const double* src) : d(dst), s(src) {}
  template <typename T> bool operator()(const T* cam1, const T* cam2, T* r) const {
    const T src[3] = {T(s[0]), T(s[1]), T(s[2])}, dst[3] = {T(d[0]), T(d[1]), T(d[2])};
    T p[3], p2[3];
    ceres::AngleAxisRotatePoint(cam1, src, p); ceres::AngleAxisRotatePoint(cam2, dst, p2);
    for (int i = 0; i < 3; ++i) r[i] = (p[i] + cam1[3 + i]) - (p2[i] + cam2[3 + i]);
    return true;
  }
};

// sophus_se3.h:10-19: x_plus_delta = x * SE3::exp(delta), delta = (upsilon, omega).  SE3::exp as Sophus publishes it: q_delta = [sin(theta/2)/theta omega,
// cos(theta/2)], t_delta = V upsilon, V = I + (1 - cos theta)/theta^2 [omega]x + (theta - sin theta)/theta^3 [omega]x^2, with the Taylor forms below
// theta < 1e-10 (the Jacobian is taken at delta = 0, i.e. ALWAYS in the small-angle branch; the branch must not differentiate sqrt at 0)
struct SE3Plus {
  template <typename T> bool operator()(const T* x, const T* delta, T* out) const {
    const T* u = delta; const T* w = delta + 3;
    const T th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    T qd[4], td[3];
    if (th2 < T(1e-20)) {
      const T imag = T(0.5) - th2 * T(1.0 / 48.0), real = T(1.0) - th2 * T(1.0 / 8.0);
      qd[0] = imag * w[0]; qd[1] = imag * w[1]; qd[2] = imag * w[2]; qd[3] = real;
      // V ~ I + [omega]x / 2
      td[0] = u[0] + T(0.5) * (w[1] * u[2] - w[2] * u[1]); td[1] = u[1] + T(0.5) * (w[2] * u[0] - w[0] * u[2]); td[2] = u[2] + T(0.5) * (w[0] * u[1] - w[1] * u[0]);
    } else {
      const T th = sqrt(th2), half = T(0.5) * th, imag = sin(half) / th;
      qd[0] = imag * w[0]; qd[1] = imag * w[1]; qd[2] = imag * w[2]; qd[3] = cos(half);
      const T a = (T(1.0) - cos(th)) / th2, b = (th - sin(th)) / (th2 * th);
      const T wu[3] = {w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]};
      const T wwu[3] = {w[1] * wu[2] - w[2] * wu[1], w[2] * wu[0] - w[0] * wu[2], w[0] * wu[1] - w[1] * wu[0]};
      for (int i = 0; i < 3; ++i) td[i] = u[i] + a * wu[i] + b * wwu[i];
    }
    // q' = q (x) q_delta  (Eigen / Hamilton product, storage [x y z w]); t' = q * t_delta + t
    const T* q = x;
    out[0] = q[3] * qd[0] + q[0] * qd[3] + q[1] * qd[2] - q[2] * qd[1];
    out[1] = q[3] * qd[1] + q[1] * qd[3] + q[2] * qd[0] - q[0] * qd[2];
    out[2] = q[3] * qd[2] + q[2] * qd[3] + q[0] * qd[1] - q[1] * qd[0];
    out[3] = q[3] * qd[3] - q[0] * qd[0] - q[1] * qd[1] - q[2] * qd[2];
    T rt[3];
    quat_rotate(q, td, rt);
    for (int i = 0; i < 3; ++i) out[4 + i] = rt[i] + x[4 + i];
    return true;
  }
};

// icp-ceres.cpp:97-134 (column-major 4x4 <-> parameters)
void pose_to_aa(const double* P, double* c) {
  double R[9];
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) R[i + 3 * j] = P[i + 4 * j];
  ceres::RotationMatrixToAngleAxis(R, c);   // column-major, like ColumnMajorAdapter4x3 on the pose matrix
  for (int i = 0; i < 3; ++i) c[3 + i] = P[12 + i];
}
void aa_to_pose(const double* c, double* P) {
  double R[9];
  ceres::AngleAxisToRotationMatrix(c, R);
  std::memset(P, 0, 16 * sizeof(double));
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) P[i + 4 * j] = R[i + 3 * j];
  for (int i = 0; i < 3; ++i) P[12 + i] = c[3 + i];
  P[15] = 1.0;
}
void pose_to_se3(const double* P, double* c) {   // Sophus::SE3d(Isometry.matrix()) -> unit quaternion (Eigen's Quaterniond(Matrix3d)) + translation
  Eigen::Matrix3d R;
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) R(i, j) = P[i + 4 * j];
  Eigen::Quaterniond q(R);
  q.normalize();
  c[0] = q.x(); c[1] = q.y(); c[2] = q.z(); c[3] = q.w();
  for (int i = 0; i < 3; ++i) c[4 + i] = P[12 + i];
}
void se3_to_pose(const double* c, double* P) {
  Eigen::Quaterniond q(c[3], c[0], c[1], c[2]);
  const Eigen::Matrix3d R = q.normalized().toRotationMatrix();
  std::memset(P, 0, 16 * sizeof(double));
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) P[i + 4 * j] = R(i, j);
  for (int i = 0; i < 3; ++i) P[12 + i] = c[4 + i];
  P[15] = 1.0;
}

template <typename T> bool rd(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: ceres_oracle problem.bin poses_out.bin\n"); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 2; }
  char magic[8];
  int32_t hdr[5];
  if (!rd(f, magic, 8) || std::memcmp(magic, "MVCERES1", 8) != 0 || !rd(f, hdr, 5)) { std::fprintf(stderr, "bad header\n"); return 2; }
  const int K = hdr[0], E = hdr[1], param = hdr[2], plane = hdr[3], robust = hdr[4];
  if (param != 1 && param != 2) { std::fprintf(stderr, "param %d: only 1 (angle-axis) and 2 (SophusSE3) are restated here\n", param); return 2; }
  std::vector<FrameIn> fr(K);
  for (FrameIn& F : fr) {
    int32_t h[2];
    if (!rd(f, h, 2)) return 2;
    F.n = h[0]; F.fixed = h[1];
    F.pts.resize(3 * (size_t)F.n); F.nor.resize(3 * (size_t)F.n);
    if (!rd(f, F.pts.data(), F.pts.size()) || !rd(f, F.nor.data(), F.nor.size()) || !rd(f, F.pose, 16)) return 2;
  }
  std::vector<EdgeIn> ed(E);
  for (EdgeIn& e : ed) {
    int32_t h[3];
    if (!rd(f, h, 3) || !rd(f, &e.weight, 1)) return 2;
    e.src = h[0]; e.dst = h[1];
    e.first.resize(h[2]); e.second.resize(h[2]);
    if (h[2] && (!rd(f, e.first.data(), e.first.size()) || !rd(f, e.second.data(), e.second.size()))) return 2;
  }
  std::fclose(f);
  if (K > 0) fr[0].fixed = 1;   // icp-ceres.cpp:341,417

  const int np = param == 1 ? 6 : 7;
  std::vector<double> cams((size_t)K * np);
  for (int i = 0; i < K; ++i) { if (param == 1) pose_to_aa(fr[i].pose, &cams[(size_t)i * np]); else pose_to_se3(fr[i].pose, &cams[(size_t)i * np]); }
  ceres::Problem problem;
  for (const EdgeIn& e : ed) {
    if (fr[e.src].fixed) continue;   // icp-ceres.cpp:351,426
    for (size_t k = 0; k < e.first.size(); ++k) {
      const double* s = &fr[e.src].pts[3 * (size_t)e.first[k]];
      const double* d = &fr[e.dst].pts[3 * (size_t)e.second[k]];
      const double* n = &fr[e.dst].nor[3 * (size_t)e.second[k]];
      ceres::CostFunction* cost;
      if (param == 1) cost = plane ? (ceres::CostFunction*)new ceres::AutoDiffCostFunction<PlaneAA, 1, 6, 6>(new PlaneAA(d, s, n))
                                   : (ceres::CostFunction*)new ceres::AutoDiffCostFunction<PointAA, 3, 6, 6>(new PointAA(d, s));
      else cost = plane ? (ceres::CostFunction*)new ceres::AutoDiffCostFunction<PlaneSE3, 1, 7, 7>(new PlaneSE3(d, s, n))
                        : (ceres::CostFunction*)new ceres::AutoDiffCostFunction<PointSE3, 3, 7, 7>(new PointSE3(d, s));
      ceres::LossFunction* loss = robust ? new ceres::SoftLOneLoss(e.weight) : nullptr;   // icp-ceres.cpp:374,449 (weight is a float there; the caller passes (double)(float)w)
      problem.AddResidualBlock(cost, loss, &cams[(size_t)e.src * np], &cams[(size_t)e.dst * np]);
    }
  }
  ceres::LocalParameterization* lp = param == 2 ? new ceres::AutoDiffLocalParameterization<SE3Plus, 7, 6> : nullptr;   // one instance for all blocks (icp-ceres.cpp:457-461)
  for (int i = 0; i < K; ++i) {
    if (!problem.HasParameterBlock(&cams[(size_t)i * np])) continue;   // (a frame no edge touches: the reference would crash in SetParameterization)
    if (lp) problem.SetParameterization(&cams[(size_t)i * np], lp);
    if (fr[i].fixed) problem.SetParameterBlockConstant(&cams[(size_t)i * np]);
  }
  ceres::Solver::Options options;   // icp-ceres.cpp:66-89
  options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;
  options.use_explicit_schur_complement = true;
  options.max_num_iterations = 50;
  ceres::Solver::Summary summary;
  ceres::Solve(options, &problem, &summary);
  std::printf("%s\n", summary.BriefReport().c_str());
  FILE* o = std::fopen(argv[2], "wb");
  if (!o) { std::perror(argv[2]); return 2; }
  for (int i = 0; i < K; ++i) {
    double P[16];
    if (param == 1) aa_to_pose(&cams[(size_t)i * np], P); else se3_to_pose(&cams[(size_t)i * np], P);   // icp-ceres.cpp:386-394,472-474: every frame round-trips
    std::fwrite(P, sizeof(double), 16, o);
  }
  const int32_t tail[2] = {(int32_t)(summary.iterations.size() > 0 ? summary.iterations.size() - 1 : 0), (int32_t)summary.termination_type};
  std::fwrite(tail, sizeof(int32_t), 2, o);
  const double costs[2] = {summary.initial_cost, summary.final_cost};
  std::fwrite(costs, sizeof(double), 2, o);
  std::fclose(o);
  return 0;
}
