// K5 — per-correspondence residual + analytic SE(3) Jacobian, robust weighting, block-reduced
// J^T J / J^T r / cost per edge.  HBM-bandwidth bound: each LM evaluation streams the packed operand
// stream once (72 B / correspondence point-to-plane, 48 B point-to-point), fully coalesced.
//
// Replaces what Ceres does with one AutoDiffCostFunction + SoftLOneLoss per correspondence
// (src/internal/icp-ceres.cpp:270-292,360-378,435-453 on the functors of include/icp-ceres.h:49-316):
// evaluate r and dr/d(pose_s, pose_d), scale both by sqrt(rho') (Ceres corrector for rho'' <= 0), and
// accumulate the normal equations.  The Jacobian is taken in canonical right-perturbation coordinates
// T <- T exp([upsilon, omega]) for both poses; the host LM maps it to the selected parameterization
// (host/lm.cpp).  Derivation (docs/mv-lm-icp.tex:109-112,306-319), with
//   A = R_d^T R_s, t = R_d^T (t_s - t_d), p~ = A p + t   (src point in the dst frame), m = A^T n:
//   point-to-plane  r = n . (p~ - q)        J = [ m , p x m , -n , n x p~ ]                    (1 x 12)
//   point-to-point  r = p~ - q  (= R_d^T (a-b), same norm)
//                   J_k = [ A(k,:) , p x A(k,:) , -e_k , [q]x(k,:) ]   k = 0..2                 (3 x 12)
//   rho(s) = 2 a^2 (sqrt(1 + s/a^2) - 1),  rho' = 1/sqrt(1 + s/a^2),  a = edge.weight  (SoftLOneLoss)
//   H += rho' J^T J,  g += rho' J^T r,  cost += rho/2
//
// Mapping: one 256-thread workgroup per chunk of kLinChunk correspondences of ONE edge; each lane owns
// two adjacent correspondences per step (16-B loads from each SoA stream) and keeps the 91 running
// sums (78 upper-triangular H + 12 g + cost) in registers; wave64 xor-shuffle reduction, LDS across the
// 4 waves, one 91-double partial per workgroup; a second tiny kernel sums the partials of each edge in
// fixed order -> results are deterministic and independent of how edges are sharded across GPUs.
#include "common.h"

namespace mvicp {

namespace {

constexpr int NT = kLinThreads;
constexpr int NB = MVICP_EDGE_BLOCK;  // 91

template <bool PLANE, bool ROBUST>
__device__ __forceinline__ void accumulate(double (&acc)[NB], const double* __restrict__ A, const double* __restrict__ t, double inv_a2, double a2,
                                           double p0, double p1, double p2, double q0, double q1, double q2, double n0, double n1, double n2) {
  const double pt0 = A[0] * p0 + A[3] * p1 + A[6] * p2 + t[0];
  const double pt1 = A[1] * p0 + A[4] * p1 + A[7] * p2 + t[1];
  const double pt2 = A[2] * p0 + A[5] * p1 + A[8] * p2 + t[2];
  const double f0 = pt0 - q0, f1 = pt1 - q1, f2 = pt2 - q2;
  if (PLANE) {
    const double r = n0 * f0 + n1 * f1 + n2 * f2;
    const double m0 = A[0] * n0 + A[1] * n1 + A[2] * n2;
    const double m1 = A[3] * n0 + A[4] * n1 + A[5] * n2;
    const double m2 = A[6] * n0 + A[7] * n1 + A[8] * n2;
    double J[12];
    J[0] = m0; J[1] = m1; J[2] = m2;
    J[3] = p1 * m2 - p2 * m1; J[4] = p2 * m0 - p0 * m2; J[5] = p0 * m1 - p1 * m0;
    J[6] = -n0; J[7] = -n1; J[8] = -n2;
    J[9] = n1 * pt2 - n2 * pt1; J[10] = n2 * pt0 - n0 * pt2; J[11] = n0 * pt1 - n1 * pt0;
    const double s = r * r;
    double w = 1.0;
    if (ROBUST) {
      const double tmp = sqrt(1.0 + s * inv_a2);
      w = 1.0 / tmp;
      acc[90] += a2 * (tmp - 1.0);
    } else {
      acc[90] += 0.5 * s;
    }
    int o = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const double wj = w * J[i];
#pragma unroll
      for (int j = i; j < 12; ++j) acc[o++] += wj * J[j];
      acc[78 + i] += wj * r;
    }
  } else {
    const double s = f0 * f0 + f1 * f1 + f2 * f2;
    double w = 1.0;
    if (ROBUST) {
      const double tmp = sqrt(1.0 + s * inv_a2);
      w = 1.0 / tmp;
      acc[90] += a2 * (tmp - 1.0);
    } else {
      acc[90] += 0.5 * s;
    }
    const double fr[3] = {f0, f1, f2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double a0 = A[k], a1 = A[k + 3], a2k = A[k + 6];  // row k of A
      double J[12];
      J[0] = a0; J[1] = a1; J[2] = a2k;
      J[3] = p1 * a2k - p2 * a1; J[4] = p2 * a0 - p0 * a2k; J[5] = p0 * a1 - p1 * a0;
      J[6] = k == 0 ? -1.0 : 0.0; J[7] = k == 1 ? -1.0 : 0.0; J[8] = k == 2 ? -1.0 : 0.0;
      // [q]x = [[0,-q2,q1],[q2,0,-q0],[-q1,q0,0]]
      J[9] = k == 0 ? 0.0 : (k == 1 ? q2 : -q1);
      J[10] = k == 0 ? -q2 : (k == 1 ? 0.0 : q0);
      J[11] = k == 0 ? q1 : (k == 1 ? -q0 : 0.0);
      int o = 0;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const double wj = w * J[i];
#pragma unroll
        for (int j = i; j < 12; ++j) acc[o++] += wj * J[j];
        acc[78 + i] += wj * fr[k];
      }
    }
  }
}

template <bool PLANE, bool ROBUST>
__global__ __launch_bounds__(NT, 2) void linearize_kernel(const int* __restrict__ chunk_edge, const int* __restrict__ chunk_start,
                                                           const int* __restrict__ count, const long long* __restrict__ cap_off, long long total_cap,
                                                           const double* __restrict__ rel, const double* __restrict__ a_scale,
                                                           const double* __restrict__ stream, double* __restrict__ partials) {
  const int c = blockIdx.x;
  const int e = chunk_edge[c];
  const int start = chunk_start[c];
  const int cnt = count[e];
  if (start >= cnt) return;
  const int end = min(cnt, start + kLinChunk);
  __shared__ double srel[kEdgeRel];
  __shared__ double red[NT / 64][NB];
  if (threadIdx.x < kEdgeRel) srel[threadIdx.x] = rel[(size_t)e * kEdgeRel + threadIdx.x];
  __syncthreads();
  double A[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) A[i] = srel[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = srel[9 + i];
  double a2 = 1.0, inv_a2 = 1.0;
  if (ROBUST) { const double a = a_scale[e]; a2 = a * a; inv_a2 = 1.0 / a2; }

  double acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = 0.0;

  const size_t base = (size_t)cap_off[e];  // multiple of 64 -> 16-B aligned double2 loads
  const double* __restrict__ s0 = stream + base;
  constexpr int NS = PLANE ? 9 : 6;
  for (int pos = start + 2 * threadIdx.x; pos < end; pos += 2 * NT) {
    double v0[9], v1[9];
    const bool two = pos + 1 < end;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const double* sp = s0 + (size_t)j * total_cap + pos;
      if (two) {
        const double2 d = *reinterpret_cast<const double2*>(sp);
        v0[j] = d.x; v1[j] = d.y;
      } else {
        v0[j] = sp[0]; v1[j] = 0.0;
      }
    }
    if (!PLANE) { v0[6] = v0[7] = v0[8] = 0.0; v1[6] = v1[7] = v1[8] = 0.0; }
    accumulate<PLANE, ROBUST>(acc, A, t, inv_a2, a2, v0[0], v0[1], v0[2], v0[3], v0[4], v0[5], v0[6], v0[7], v0[8]);
    if (two) accumulate<PLANE, ROBUST>(acc, A, t, inv_a2, a2, v1[0], v1[1], v1[2], v1[3], v1[4], v1[5], v1[6], v1[7], v1[8]);
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    double v = acc[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NB) {
    double v = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) v += red[w][threadIdx.x];
    partials[(size_t)c * NB + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(128) void reduce_kernel(const int* __restrict__ chunk_first, const int* __restrict__ count,
                                                     const double* __restrict__ partials, double* __restrict__ out) {
  const int e = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid >= NB) return;
  const int c0 = chunk_first[e];
  const int nchunks = min(chunk_first[e + 1] - c0, (count[e] + kLinChunk - 1) / kLinChunk);
  double v = 0.0;
  for (int c = 0; c < nchunks; ++c) v += partials[(size_t)(c0 + c) * NB + tid];
  out[(size_t)e * NB + tid] = v;
}

}  // namespace

int launch_linearize(mvicp_ctx* c, int plane, int robust) {
  if (c->E == 0) return MVICP_OK;
  if (c->n_chunks > 0) {
    double bytes = 0;
    for (int e = 0; e < c->E; ++e) if (c->owned[e]) bytes += (plane ? 72.0 : 48.0) * c->h_count[e];
    ProfScope ps(c, "linearize", bytes);
#define LAUNCH(P, R)                                                                                                                        \
  hipLaunchKernelGGL((linearize_kernel<P, R>), dim3(c->n_chunks), dim3(NT), 0, c->stream, c->d_chunk_edge, c->d_chunk_start, c->d_count,    \
                     c->d_cap_off, c->total_cap, c->d_rel, c->d_a, c->d_stream, c->d_partials)
    if (plane && robust) LAUNCH(true, true);
    else if (plane) LAUNCH(true, false);
    else if (robust) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
  }
  {
    ProfScope ps(c, "reduce", 0.0);
    hipLaunchKernelGGL(reduce_kernel, dim3(c->E), dim3(128), 0, c->stream, c->d_chunk_first, c->d_count, c->d_partials, c->d_out);
  }
  MV_HIP(hipGetLastError());
  return MVICP_OK;
}

}  // namespace mvicp
