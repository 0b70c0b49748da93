// K5 — per-correspondence residual + analytic SE(3) Jacobian, robust weighting, block-reduced
// J^T J / J^T r / cost per edge.  HBM-bandwidth bound: each LM evaluation streams the packed operand
// stream once (56 B / correspondence point-to-plane, 48 B point-to-point), fully coalesced.
//
// Operand stream (SoA, `total_cap` doubles per array, written once per round by the gather kernel, corr.hip):
//   arrays 0-2 p (source point, src frame) | 3-5 n (dst normal) | 6 c = n . q | 7-9 q (dst point).
// Point-to-plane needs q only through the scalar c:  r = n . (p~ - q) = n . p~ - c,  so it reads 7 arrays instead
// of 9 (-22 % bytes on the kernel that dominates the LM phase).  The two forms of r differ by the rounding of two
// O(|p~|) dot products, ~3e-16 absolute, i.e. ~1e-14 relative in the assembled g: far inside the 1e-11 block tolerance.
//
// Replaces what Ceres does with one AutoDiffCostFunction + SoftLOneLoss per correspondence
// (src/internal/icp-ceres.cpp:270-292,360-378,435-453 on the functors of include/icp-ceres.h:49-316):
// evaluate r and dr/d(pose_s, pose_d), scale both by sqrt(rho') (Ceres corrector for rho'' <= 0), and
// accumulate the normal equations, in canonical right-perturbation coordinates T <- T exp([upsilon, omega])
// for both poses (the host LM maps them to the selected parameterization, host/lm.cpp).
//
// Structure exploited (docs/mv-lm-icp.tex:109-112,306-319).  With the RELATIVE transform
//   A = R_d^T R_s,  t = R_d^T (t_s - t_d),  p~ = A p + t  (source point in the dst frame),
// and Ad = [[A, [t]x A], [0, A]] its adjoint, every Jacobian row is a fixed linear image of a 6-vector
// that lives in the dst frame:
//   point-to-plane  r = n . (p~ - q),  u = [n ; p~ x n],           J = [ Ad^T u ; -u ]
//   point-to-point  r = p~ - q,        u_k = [e_k ; p~ x e_k],     J_k = [ Ad^T u_k ; -u_k + [0 ; r x e_k] ]
// so instead of 78 + 12 + 1 running sums per lane only the weighted 6x6 moment block is accumulated:
//   plane (28 sums):  U = sum w u u^T (21), v = sum w r u (6), cost
//   point (29 sums):  sum w {1, p~ (3), p~ p~^T (6), r (3), p~ r^T (9), r r^T (6)}, cost
// with w = rho'(|r|^2) = 1/sqrt(1 + |r|^2 / a^2), cost = sum rho/2, rho = 2 a^2 (sqrt(1 + s/a^2) - 1)
// (ceres::SoftLOneLoss(a = edge.weight) [upstream]), and the 12x12 block is expanded ONCE per edge:
//   H_ss = Ad^T S Ad, H_sd = -Ad^T (S - X), H_dd = S - X - X^T + Y, g = [Ad^T v ; -v]
//   (plane: S = U, X = Y = 0; point: S = sum w [[I, -[p~]x],[[p~]x, -[p~]x^2]], X = sum w [[0,-[r]x],[0,-[p~]x[r]x]],
//    Y = sum w [[0,0],[0,-[r]x^2]], v = sum w [r ; p~ x r]).
//
// Mapping: one 256-thread workgroup per chunk of `chunk` correspondences of ONE edge; ~60 accumulator
// VGPRs per lane leave room to keep the next correspondences' loads in flight; transposed LDS block
// reduction; one partial per workgroup; a second kernel sums each edge's partials in fixed order and does
// the expansion -> deterministic, and independent of how edges are sharded across GPUs.
#include "common.h"

namespace mvicp {

namespace {

constexpr int NT = kLinThreads;
constexpr int NB = MVICP_EDGE_BLOCK;  // 91
constexpr int NACC = kLinPartial;     // padded partial width (28 plane / 29 point)

__device__ __forceinline__ double fast_rsqrt(double y) {
  // y in [1, huge): v_rsq_f64 seed + two Newton steps (each squares the error) -> ~1 ulp
  double r = __builtin_amdgcn_rsq(y);
  r = r * (1.5 - 0.5 * y * r * r);
  r = r * (1.5 - 0.5 * y * r * r);
  return r;
}

template <bool PLANE, bool ROBUST>
__device__ __forceinline__ void accumulate(double (&acc)[NACC], const double* __restrict__ A, const double* __restrict__ t, double inv_a2, double a2,
                                           double p0, double p1, double p2, double q0, double q1, double q2, double n0, double n1, double n2) {
  // PLANE: (q0, q1, q2) = (c, -, -) with c = n . q;  POINT: (n0, n1, n2) unused
  const double x0 = A[0] * p0 + A[3] * p1 + A[6] * p2 + t[0];
  const double x1 = A[1] * p0 + A[4] * p1 + A[7] * p2 + t[1];
  const double x2 = A[2] * p0 + A[5] * p1 + A[8] * p2 + t[2];
  if (PLANE) {
    const double r = (n0 * x0 + n1 * x1 + n2 * x2) - q0;
    double u[6];
    u[0] = n0; u[1] = n1; u[2] = n2;
    u[3] = x1 * n2 - x2 * n1; u[4] = x2 * n0 - x0 * n2; u[5] = x0 * n1 - x1 * n0;
    const double s = r * r;
    double w = 1.0;
    if (ROBUST) {
      const double y = 1.0 + s * inv_a2;
      w = fast_rsqrt(y);
      acc[27] += a2 * (y * w - 1.0);
    } else {
      acc[27] += 0.5 * s;
    }
    int o = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double wu = w * u[i];
#pragma unroll
      for (int j = i; j < 6; ++j) acc[o++] += wu * u[j];
      acc[21 + i] += wu * r;
    }
  } else {
    const double f0 = x0 - q0, f1 = x1 - q1, f2 = x2 - q2;
    const double s = f0 * f0 + f1 * f1 + f2 * f2;
    double w = 1.0;
    if (ROBUST) {
      const double y = 1.0 + s * inv_a2;
      w = fast_rsqrt(y);
      acc[28] += a2 * (y * w - 1.0);
    } else {
      acc[28] += 0.5 * s;
    }
    const double wx0 = w * x0, wx1 = w * x1, wx2 = w * x2;
    const double wf0 = w * f0, wf1 = w * f1, wf2 = w * f2;
    acc[0] += w;
    acc[1] += wx0; acc[2] += wx1; acc[3] += wx2;
    acc[4] += wx0 * x0; acc[5] += wx0 * x1; acc[6] += wx0 * x2; acc[7] += wx1 * x1; acc[8] += wx1 * x2; acc[9] += wx2 * x2;
    acc[10] += wf0; acc[11] += wf1; acc[12] += wf2;
    acc[13] += wx0 * f0; acc[14] += wx0 * f1; acc[15] += wx0 * f2;
    acc[16] += wx1 * f0; acc[17] += wx1 * f1; acc[18] += wx1 * f2;
    acc[19] += wx2 * f0; acc[20] += wx2 * f1; acc[21] += wx2 * f2;
    acc[22] += wf0 * f0; acc[23] += wf0 * f1; acc[24] += wf0 * f2; acc[25] += wf1 * f1; acc[26] += wf1 * f2; acc[27] += wf2 * f2;
  }
}

template <bool PLANE, bool ROBUST>
__global__ __launch_bounds__(NT) void linearize_kernel(const int* __restrict__ chunk_edge, const int* __restrict__ chunk_start, int chunk,
                                                       const int* __restrict__ count, const long long* __restrict__ cap_off, long long total_cap,
                                                       const double* __restrict__ rel, const double* __restrict__ a_scale,
                                                       const double* __restrict__ stream, double* __restrict__ partials,
                                                       const double* const* __restrict__ src_pts, const int* __restrict__ nsrc,
                                                       const int* __restrict__ chunk_first) {
  const int e = chunk_edge[blockIdx.x];
  const int start = chunk_start[blockIdx.x];
  // the partial's slot is the chunk's place in ITS EDGE's run (chunk_first[e] + k), whatever the launch order of the workgroups (api.cpp interleaves the chunks
  // of the edges that share a source cloud, so that the second reader of a piece of p finds it in the same XCD's L2); reduce_expand_kernel sums an edge's slots
  // in order, so the result does not depend on that order
  const int c = chunk_first[e] + start / chunk;
  const int cnt = count[e];
  if (start >= cnt) return;
  const int end = min(cnt, start + chunk);
  __shared__ double srel[kEdgeRel];
  __shared__ double red[NACC / 2][NT + 1];
  if (threadIdx.x < kEdgeRel) srel[threadIdx.x] = rel[(size_t)e * kEdgeRel + threadIdx.x];
  __syncthreads();
  double A[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) A[i] = srel[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = srel[9 + i];
  double a2 = 1.0, inv_a2 = 1.0;
  if (ROBUST) { const double a = a_scale[e]; a2 = a * a; inv_a2 = 1.0 / a2; }

  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;

  const size_t base = (size_t)cap_off[e];  // multiple of 64 -> 16-B aligned double2 loads
  const double* __restrict__ s0 = stream + base;
  constexpr int NS = PLANE ? 7 : 6;
  // register slot j -> stream array: plane p n c = arrays 0..6; point p q = arrays 0-2, 7-9
  auto arr = [](int j) { return PLANE ? j : (j < 3 ? j : j + 4); };
  // two adjacent correspondences per lane per step (16-B loads); the next step's loads are issued before
  // the current step's arithmetic so two steps of HBM latency overlap.
  int pos = start + 2 * threadIdx.x;
  double2 cur[9], nxt[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) { cur[j] = make_double2(0.0, 0.0); nxt[j] = make_double2(0.0, 0.0); }
  // Every query accepted (count == N_src: the usual case once the clouds overlap within the cutoff) makes the list the identity,
  // first[pos] = pos, so p is the source cloud itself in its sorted order: read it from there (AoS, three 16-B loads per pair of
  // correspondences, ordinary cacheable loads) instead of from the stream's private copy.  The two edges of a source frame are
  // consecutive in the chunk order and share that 24 B per point through L2 / MALL — same values, so results are bit-identical.
  const double* __restrict__ sp = (src_pts != nullptr && cnt == nsrc[e]) ? src_pts[e] : nullptr;
  auto load = [&](double2 (&v)[9], int at) {
    if (at + 1 < end) {
      if (sp != nullptr) {
        const double2 a = *reinterpret_cast<const double2*>(sp + 3 * (size_t)at);
        const double2 b = *reinterpret_cast<const double2*>(sp + 3 * (size_t)at + 2);
        const double2 c2 = *reinterpret_cast<const double2*>(sp + 3 * (size_t)at + 4);
        v[0] = make_double2(a.x, b.y); v[1] = make_double2(a.y, c2.x); v[2] = make_double2(b.x, c2.y);
      }
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        if (j < 3 && sp != nullptr) continue;
        // non-temporal: the stream is read exactly once per evaluation and is larger than the caches; keeping it from
        // allocating there is worth +15 % bandwidth (cfg4 136 -> 117 us, cfg5 6.3 -> 7.0 TB/s)
        typedef double d2v __attribute__((ext_vector_type(2)));
        const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(s0 + (size_t)arr(j) * total_cap + at));
        v[j] = make_double2(t.x, t.y);
      }
    } else if (at < end) {
#pragma unroll
      for (int j = 0; j < NS; ++j) { v[j].x = (j < 3 && sp != nullptr) ? sp[3 * (size_t)at + j] : s0[(size_t)arr(j) * total_cap + at]; v[j].y = 0.0; }
    }
  };
  load(cur, pos);
  while (pos < end) {
    const int npos = pos + 2 * NT;
    load(nxt, npos);
    // slots: plane 0-2 p, 3-5 n, 6 c;  point 0-2 p, 3-5 q
    if (PLANE) {
      accumulate<PLANE, ROBUST>(acc, A, t, inv_a2, a2, cur[0].x, cur[1].x, cur[2].x, cur[6].x, 0.0, 0.0, cur[3].x, cur[4].x, cur[5].x);
      if (pos + 1 < end)
        accumulate<PLANE, ROBUST>(acc, A, t, inv_a2, a2, cur[0].y, cur[1].y, cur[2].y, cur[6].y, 0.0, 0.0, cur[3].y, cur[4].y, cur[5].y);
    } else {
      accumulate<PLANE, ROBUST>(acc, A, t, inv_a2, a2, cur[0].x, cur[1].x, cur[2].x, cur[3].x, cur[4].x, cur[5].x, 0.0, 0.0, 0.0);
      if (pos + 1 < end)
        accumulate<PLANE, ROBUST>(acc, A, t, inv_a2, a2, cur[0].y, cur[1].y, cur[2].y, cur[3].y, cur[4].y, cur[5].y, 0.0, 0.0, 0.0);
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) cur[j] = nxt[j];
    pos = npos;
  }

  // Block reduction through LDS, transposed: every thread stores value j into row j (stride-1 across lanes:
  // conflict-free ds_write_b64), then 16 threads per row add 16 columns each and finish with a 4-step xor-shuffle
  // inside their lane group (two passes of 16 rows keep the LDS footprint at 33 KB).  Fixed association order -> deterministic.
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int j = 0; j < NACC / 2; ++j) red[j][threadIdx.x] = acc[pass * (NACC / 2) + j];
    __syncthreads();
    const int row = threadIdx.x >> 4, part = threadIdx.x & 15;  // 16 rows x 16 parts
    double sum = 0.0;
#pragma unroll 8
    for (int k = 0; k < NT / 16; ++k) sum += red[row][part + 16 * k];
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64);
    sum += __shfl_xor(sum, 8, 64);
    if (part == 0) partials[(size_t)c * NACC + pass * (NACC / 2) + row] = sum;
  }
}

// ---- per edge: fixed-order sum of the workgroup partials, then expansion to the canonical 12x12 block
__device__ __forceinline__ void cross_mat(const double* a, double* M) {  // row-major [a]x
  M[0] = 0; M[1] = -a[2]; M[2] = a[1];
  M[3] = a[2]; M[4] = 0; M[5] = -a[0];
  M[6] = -a[1]; M[7] = a[0]; M[8] = 0;
}

template <bool PLANE>
__global__ __launch_bounds__(256) void reduce_expand_kernel(const int* __restrict__ chunk_first, int chunk, const int* __restrict__ count,
                                                           const double* __restrict__ rel, const double* __restrict__ partials,
                                                           double* __restrict__ out) {
  const int e = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ double m[8][NACC];
  __shared__ double S[36], X[36], Y[36], Ad[36], T1[36], T2[36], H[144], v[6];
  const int c0 = chunk_first[e];
  const int nchunks = min(chunk_first[e + 1] - c0, (count[e] + chunk - 1) / chunk);
  {
    // 256 threads: eight interleaved fixed-order partial sums per value (short dependent-load chains), combined in fixed order
    const int val = tid & (NACC - 1), part = tid >> 5;
    double s = 0.0;
    for (int c = part; c < nchunks; c += 8) s += partials[(size_t)(c0 + c) * NACC + val];
    m[part][val] = s;
  }
  if (tid < 36) { S[tid] = 0.0; X[tid] = 0.0; Y[tid] = 0.0; Ad[tid] = 0.0; }
  __syncthreads();
  if (tid < NACC) {
    double s = m[0][tid];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += m[k][tid];
    m[0][tid] = s;
  }
  __syncthreads();
  const double* mm = m[0];
  const double* A = rel + (size_t)e * kEdgeRel;  // column-major 3x3
  const double* t = A + 9;
  if (tid < 9) {
    // Ad = [[A, [t]x A],[0, A]]   (row-major 6x6)
    const int i = tid / 3, j = tid % 3;
    double tx[9];
    cross_mat(t, tx);
    const double a = A[i + 3 * j];
    Ad[i * 6 + j] = a;
    Ad[(3 + i) * 6 + 3 + j] = a;
    Ad[i * 6 + 3 + j] = tx[i * 3 + 0] * A[0 + 3 * j] + tx[i * 3 + 1] * A[1 + 3 * j] + tx[i * 3 + 2] * A[2 + 3 * j];
    if (PLANE) {
      if (tid < 6) v[tid] = mm[21 + tid];
    } else {
      const double w = mm[0];
      const double px[3] = {mm[1], mm[2], mm[3]};
      const double P[9] = {mm[4], mm[5], mm[6], mm[5], mm[7], mm[8], mm[6], mm[8], mm[9]};            // sum w p p^T
      const double rr[3] = {mm[10], mm[11], mm[12]};
      const double PR[9] = {mm[13], mm[14], mm[15], mm[16], mm[17], mm[18], mm[19], mm[20], mm[21]};  // sum w p r^T (row-major)
      const double RR[9] = {mm[22], mm[23], mm[24], mm[23], mm[25], mm[26], mm[24], mm[26], mm[27]};
      double pxm[9], rxm[9];
      cross_mat(px, pxm);
      cross_mat(rr, rxm);
      const double trP = P[0] + P[4] + P[8], trRR = RR[0] + RR[4] + RR[8], trPR = PR[0] + PR[4] + PR[8];
      const double I = i == j ? 1.0 : 0.0;
      // S = sum w [[I, -[p]x],[[p]x, |p|^2 I - p p^T]]
      S[i * 6 + j] = w * I;
      S[i * 6 + 3 + j] = -pxm[i * 3 + j];
      S[(3 + i) * 6 + j] = pxm[i * 3 + j];
      S[(3 + i) * 6 + 3 + j] = trP * I - P[i * 3 + j];
      // X = sum w [[0, -[r]x],[0, -[p]x[r]x]],  [p]x[r]x = r p^T - (p.r) I
      X[i * 6 + 3 + j] = -rxm[i * 3 + j];
      X[(3 + i) * 6 + 3 + j] = -(PR[j * 3 + i] - trPR * I);
      // Y = sum w [[0,0],[0, |r|^2 I - r r^T]]
      Y[(3 + i) * 6 + 3 + j] = trRR * I - RR[i * 3 + j];
      if (tid == 0) {
        // v = sum w [r ; p x r],  (p x r) from the antisymmetric part of p r^T
        v[0] = rr[0]; v[1] = rr[1]; v[2] = rr[2];
        v[3] = PR[1 * 3 + 2] - PR[2 * 3 + 1];
        v[4] = PR[2 * 3 + 0] - PR[0 * 3 + 2];
        v[5] = PR[0 * 3 + 1] - PR[1 * 3 + 0];
      }
    }
  }
  if (PLANE && tid >= 16 && tid < 16 + 21) {
    // unpack U (upper triangle, row-major) into the full symmetric S
    const int k = tid - 16;
    int i = 0, o = k;
    while (o >= 6 - i) { o -= 6 - i; ++i; }
    const int j = i + o;
    S[i * 6 + j] = mm[k];
    S[j * 6 + i] = mm[k];
  }
  __syncthreads();
  if (tid < 36) {
    const int i = tid / 6, j = tid % 6;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += S[i * 6 + k] * Ad[k * 6 + j];
    T1[tid] = s;                 // S Ad
    T2[tid] = S[tid] - X[tid];   // S - X
  }
  __syncthreads();
  double* o = out + (size_t)e * NB;
  if (tid < 36) {
    const int i = tid / 6, j = tid % 6;
    double hss = 0.0, hsd = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { hss += Ad[k * 6 + i] * T1[k * 6 + j]; hsd += Ad[k * 6 + i] * T2[k * 6 + j]; }
    H[i * 12 + j] = hss;                                                             // H_ss = Ad^T S Ad
    H[i * 12 + 6 + j] = -hsd;                                                        // H_sd = -Ad^T (S - X)
    H[(6 + i) * 12 + 6 + j] = S[i * 6 + j] - X[i * 6 + j] - X[j * 6 + i] + Y[i * 6 + j];  // H_dd
  } else if (tid < 42) {
    const int i = tid - 36;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += Ad[k * 6 + i] * v[k];
    o[78 + i] = s;           // g_s = Ad^T v
    o[84 + i] = -v[i];       // g_d = -v
  } else if (tid == 42) {
    o[90] = count[e] > 0 ? mm[PLANE ? 27 : 28] : 0.0;
  }
  __syncthreads();
  for (int k = tid; k < 78; k += 256) {
    int i = 0, r = k;
    while (r >= 12 - i) { r -= 12 - i; ++i; }
    const int j = i + r;
    // H_ss and H_dd are symmetric up to rounding: average the two triangles so the block is exactly symmetric
    o[k] = (i < 6 && j >= 6) ? H[i * 12 + j] : 0.5 * (H[i * 12 + j] + H[j * 12 + i]);
  }
}

}  // namespace

int launch_linearize(mvicp_ctx* c, int plane, int robust) {
  if (c->E == 0) return MVICP_OK;
  const int chunk = c->lin_chunk;
  if (c->n_chunks > 0) {
    // algorithmic bytes of the launch: 32 B (plane: n, n.q) / 24 B (point: q) per correspondence + the source point p, 24 B — per correspondence when the edge
    // reads its private copy from the stream, ONCE PER SOURCE POINT for the edges of one source cloud that read the shared sorted cloud side by side (identity
    // lists, lin_share_p + lin_interleave: the second edge's read is an L2 hit by construction; PMC: 0.695 -> 0.551 GB per launch at cfg4)
    double bytes = 0;
    std::vector<char> src_counted((size_t)c->n_frames, 0);
    for (int e = 0; e < c->E; ++e) {
      if (!c->owned[e]) continue;
      const double cnt = c->h_count[e];
      const int s = c->esrc[e];
      bytes += (plane ? 32.0 : 24.0) * cnt;
      if (c->lin_share_p && c->lin_interleave && c->h_count[e] == c->frames[s].n) { if (!src_counted[s]) { bytes += 24.0 * cnt; src_counted[s] = 1; } }
      else bytes += 24.0 * cnt;
    }
    // per-edge sorted source clouds (identity-list fast path of the kernel); table cached by content
    const double* const* d_src = nullptr;
    if (c->lin_share_p) {
      std::vector<const double*> tab((size_t)c->E, nullptr);
      for (int e = 0; e < c->E; ++e) if (c->owned[e]) tab[e] = c->frames[c->esrc[e]].grid.spts;
      MV_CHECK(cached_upload(c, "lin_src", tab.data(), sizeof(void*) * tab.size(), (void**)&d_src));
    }
    ProfScope ps(c, "linearize", bytes);
#define LAUNCH(P, R)                                                                                                                           \
  hipLaunchKernelGGL((linearize_kernel<P, R>), dim3(c->n_chunks), dim3(NT), 0, c->stream, c->d_chunk_edge, c->d_chunk_start, chunk, c->d_count, \
                     c->d_cap_off, c->total_cap, c->d_rel, c->d_a, c->d_stream, c->d_partials, d_src, (const int*)c->d_nsrc, (const int*)c->d_chunk_first)
    if (plane && robust) LAUNCH(true, true);
    else if (plane) LAUNCH(true, false);
    else if (robust) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
  }
  {
    ProfScope ps(c, "reduce", 0.0);
    if (plane)
      hipLaunchKernelGGL((reduce_expand_kernel<true>), dim3(c->E), dim3(256), 0, c->stream, c->d_chunk_first, chunk, c->d_count, c->d_rel, c->d_partials, c->lin_out ? c->lin_out : c->d_out);
    else
      hipLaunchKernelGGL((reduce_expand_kernel<false>), dim3(c->E), dim3(256), 0, c->stream, c->d_chunk_first, chunk, c->d_count, c->d_rel, c->d_partials, c->lin_out ? c->lin_out : c->d_out);
  }
  MV_HIP(hipGetLastError());
  return MVICP_OK;
}

}  // namespace mvicp
