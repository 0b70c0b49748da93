// Closed-form pairwise solvers on the host: the reference's comparison baselines / sanity oracles
// (include/icp-closedform.h:10-11, src/internal/icp-closedform.cpp:9-26 point-to-point, :30-54 point-to-plane),
// SURVEY.md §8 row f4.  They run next to the LM path in bin/pairwise exactly like main_pairwise.cpp:74-76,93-95, and give the
// tests an independent answer for the point-to-point / point-to-plane normal equations the GPU assembles.
//
// point-to-point: the least-squares rigid transform between index-aligned sets.  The reference takes the SVD of the 3x3
// correlation matrix (Kabsch / Eggert et al.); here the same optimum is obtained with Horn's quaternion form — the rotation is
// the eigenvector of the largest eigenvalue of a symmetric 4x4 matrix built from the same correlation matrix (cyclic Jacobi,
// fp64) — which needs no reflection fix-up and no SVD.  t = q_mean - R p_mean as in icp-closedform.cpp:25.
// point-to-plane: one Gauss-Newton step of sum ((R p + t - q) . n)^2 linearised at identity with x = [alpha beta gamma | t]
// (icp-closedform.cpp:34-45: C x = d, rows [p x n ; n]); solved by a 6x6 LDL^T; R = Rx(alpha) Ry(beta) Rz(gamma) (:47-51).
#include <cmath>
#include <cstring>

#include "../../include/mvicp.h"
#include "../csrc/abi_guard.h"

namespace mvicp {
void set_error(const char* fmt, ...);
}

namespace {

// cyclic Jacobi on a symmetric n x n matrix (row-major, destroyed); V row-major holds eigenvectors in its columns
template <int N>
void jacobi_eig(double (&A)[N][N], double (&V)[N][N], double (&w)[N]) {
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < N; ++i) { diag += A[i][i] * A[i][i]; for (int j = i + 1; j < N; ++j) off += A[i][j] * A[i][j]; }
    if (off <= 1e-36 * (diag + 1e-300)) break;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
        for (int k = 0; k < N; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
        for (int k = 0; k < N; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
      }
  }
  for (int i = 0; i < N; ++i) w[i] = A[i][i];
}

void set_pose(double* P, const double R[3][3], const double t[3]) {
  std::memset(P, 0, 16 * sizeof(double));
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P[i + 4 * j] = R[i][j]; P[12 + i] = t[i]; }
  P[15] = 1.0;
}

}  // namespace

extern "C" {

int mvicp_closedform_point_to_point(const double* src, const double* dst, int n, double* pose_out) try {
  if (!src || !dst || !pose_out || n < 3) { mvicp::set_error("closedform_point_to_point: need >= 3 pairs and non-null buffers"); return MVICP_ERR_ARG; }
  double pm[3] = {0, 0, 0}, qm[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) { pm[a] += src[3 * (size_t)i + a]; qm[a] += dst[3 * (size_t)i + a]; }
  for (int a = 0; a < 3; ++a) { pm[a] /= n; qm[a] /= n; }
  // S[a][b] = sum (p - pm)_a (q - qm)_b
  double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int i = 0; i < n; ++i) {
    const double p[3] = {src[3 * (size_t)i] - pm[0], src[3 * (size_t)i + 1] - pm[1], src[3 * (size_t)i + 2] - pm[2]};
    const double q[3] = {dst[3 * (size_t)i] - qm[0], dst[3 * (size_t)i + 1] - qm[1], dst[3 * (size_t)i + 2] - qm[2]};
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S[a][b] += p[a] * q[b];
  }
  // Horn 1987, eq. (N): unit quaternion (w, x, y, z) maximising q^T N q rotates src onto dst
  double Nm[4][4] = {
      {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
      {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
      {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
      {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
  double V[4][4], w[4];
  jacobi_eig<4>(Nm, V, w);
  int best = 0;
  for (int i = 1; i < 4; ++i) if (w[i] > w[best]) best = i;
  double qw = V[0][best], qx = V[1][best], qy = V[2][best], qz = V[3][best];
  const double nq = std::sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  if (!(nq > 0.0) || !std::isfinite(nq)) { mvicp::set_error("closedform_point_to_point: degenerate input"); return MVICP_ERR_NUMERIC; }
  qw /= nq; qx /= nq; qy /= nq; qz /= nq;
  const double R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)},
                          {2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)},
                          {2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)}};
  double t[3];
  for (int a = 0; a < 3; ++a) t[a] = qm[a] - (R[a][0] * pm[0] + R[a][1] * pm[1] + R[a][2] * pm[2]);
  set_pose(pose_out, R, t);
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_closedform_point_to_plane(const double* src, const double* dst, const double* nor, int n, double* pose_out) try {
  if (!src || !dst || !nor || !pose_out || n < 6) { mvicp::set_error("closedform_point_to_plane: need >= 6 pairs and non-null buffers"); return MVICP_ERR_ARG; }
  double C[6][6], d[6];
  for (int i = 0; i < 6; ++i) { d[i] = 0.0; for (int j = 0; j < 6; ++j) C[i][j] = 0.0; }
  for (int i = 0; i < n; ++i) {
    const double* p = src + 3 * (size_t)i; const double* q = dst + 3 * (size_t)i; const double* m = nor + 3 * (size_t)i;
    const double u[6] = {p[1] * m[2] - p[2] * m[1], p[2] * m[0] - p[0] * m[2], p[0] * m[1] - p[1] * m[0], m[0], m[1], m[2]};   // [p x n ; n]
    const double r = (p[0] - q[0]) * m[0] + (p[1] - q[1]) * m[1] + (p[2] - q[2]) * m[2];
    for (int a = 0; a < 6; ++a) { d[a] -= u[a] * r; for (int b = a; b < 6; ++b) C[a][b] += u[a] * u[b]; }
  }
  for (int a = 0; a < 6; ++a) for (int b = 0; b < a; ++b) C[a][b] = C[b][a];
  // LDL^T without pivoting (C is symmetric positive definite for non-degenerate geometry)
  double L[6][6], D[6], x[6];
  for (int j = 0; j < 6; ++j) {
    double v = C[j][j];
    for (int k = 0; k < j; ++k) v -= L[j][k] * L[j][k] * D[k];
    if (!(std::fabs(v) > 0.0) || !std::isfinite(v)) { mvicp::set_error("closedform_point_to_plane: singular normal matrix"); return MVICP_ERR_NUMERIC; }
    D[j] = v; L[j][j] = 1.0;
    for (int i = j + 1; i < 6; ++i) {
      double s = C[i][j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k] * D[k];
      L[i][j] = s / v;
    }
  }
  for (int i = 0; i < 6; ++i) { double s = d[i]; for (int k = 0; k < i; ++k) s -= L[i][k] * x[k]; x[i] = s; }
  for (int i = 0; i < 6; ++i) x[i] /= D[i];
  for (int i = 5; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k]; x[i] = s; }
  const double ca = std::cos(x[0]), sa = std::sin(x[0]), cb = std::cos(x[1]), sb = std::sin(x[1]), cg = std::cos(x[2]), sg = std::sin(x[2]);
  // Rx(alpha) Ry(beta) Rz(gamma)
  const double R[3][3] = {{cb * cg, -cb * sg, sb},
                          {sa * sb * cg + ca * sg, -sa * sb * sg + ca * cg, -sa * cb},
                          {-ca * sb * cg + sa * sg, ca * sb * sg + sa * cg, ca * cb}};
  const double t[3] = {x[3], x[4], x[5]};
  set_pose(pose_out, R, t);
  return MVICP_OK;
} MVICP_GUARD_ABI

}  // extern "C"
