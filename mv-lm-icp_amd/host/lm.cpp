// S2 — the host Levenberg-Marquardt solve over the per-edge normal-equation blocks produced on the GPU.
//
// Replaces ceres::Solve as driven by ICP_Ceres::ceresOptimizer{,_ceresAngleAxis,_sophusSE3}
// (src/internal/icp-ceres.cpp:220-475 with the options of getOptionsMedium, :66-89: max 50 iterations,
// SPARSE_NORMAL_CHOLESKY, everything else Ceres default).  Ceres itself is third-party and absent from
// the reference tree; the loop below follows its published trust-region algorithm
// (TrustRegionMinimizer + LevenbergMarquardtStrategy, Ceres 1.13-2.1 [upstream]):
//   * Jacobi scaling 1/(1+||J_col||) computed once at iteration 0
//   * step: solve (H_s + diag(clamp(diag H_s, 1e-6, 1e32))/radius) y = g_s, step = -y (dense Cholesky;
//     the (6(K-1))^2 system is tiny, K <= 64)
//   * model_cost_change = -(step.g_s + step^T H_s step / 2) must be > 0, else the step is invalid
//   * candidate = x (+) (step .* scale); parameter tolerance 1e-8, function tolerance 1e-6 (both stop
//     WITHOUT taking the candidate), accept iff cost_change / model_cost_change > 1e-3
//   * radius <- radius / max(1/3, 1 - (2 rho - 1)^3) on accept, radius /= decrease_factor (2,4,8..) on reject
//   * gradient tolerance 1e-10 on the max-norm of the unscaled gradient
// One device evaluation per LM iteration: the candidate's cost AND normal equations come out of the
// same kernel pass (they are needed as soon as the step is accepted, the common case).
//
// The device blocks are in canonical right-perturbation coordinates; the chain rule to the selected
// parameterization is a per-pose 6x6 map M (se3.h): H' = M^T H M, g' = M^T g.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../csrc/common.h"
#include "se3.h"

namespace mvicp {

int evaluate_blocks(mvicp_ctx* c, const double* poses, int plane, int robust, double* out);  // csrc/api.cpp

namespace {

struct Assembler {
  int K, E, param;
  const int* src; const int* dst;
  std::vector<int> fidx;
  int nfree = 0;
  std::vector<double> M;  // K x 36
  Assembler(int K_, int E_, const int* s, const int* d, const unsigned char* fixed, int p) : K(K_), E(E_), param(p), src(s), dst(d), fidx(K_, -1), M((size_t)K_ * 36) {
    for (int i = 0; i < K; ++i) if (!fixed[i]) fidx[i] = nfree++;
  }
  int n() const { return 6 * nfree; }

  // blocks (E x 91, canonical) at parameters x -> H (n x n row-major; only the LOWER ENVELOPE env[i] <= j <= i of every row
  // is written and meaningful: the system is block-banded for a turntable graph, and every consumer below stays inside the
  // envelope, so a solve costs O(n * bandwidth) per pass instead of O(n^2)), g (n); returns cost
  double assemble(const double* x, const double* blocks, double* H, double* g, const int* env) {
    const int A = se3::ambient(param), nn = n();
    if (param != MVICP_PARAM_SOPHUS_SE3) for (int i = 0; i < K; ++i) se3::local_to_canonical(param, x + (size_t)i * A, &M[(size_t)i * 36]);
    for (int i = 0; i < nn; ++i) std::fill(H + (size_t)i * nn + env[i], H + (size_t)i * nn + (i / 6) * 6 + 6, 0.0);   // up to the end of the diagonal block
    std::fill(g, g + nn, 0.0);
    double cost = 0.0;
    double Hc[12][12], T[6][6], Hl[6][6];
    if (param == MVICP_PARAM_SOPHUS_SE3) {
      // M = I for every pose: the canonical blocks ARE the local ones (packed upper triangle of the 12x12, row by row)
      for (int e = 0; e < E; ++e) {
        if (fidx[src[e]] < 0) continue;   // icp-ceres.cpp:255,351,426: a fixed SOURCE contributes no residual blocks at all
        const double* b = blocks + (size_t)e * MVICP_EDGE_BLOCK;
        cost += b[90];
        const int fr[2] = {fidx[src[e]], fidx[dst[e]]};
        for (int bi = 0; bi < 2; ++bi)
          if (fr[bi] >= 0) for (int l = 0; l < 6; ++l) g[fr[bi] * 6 + l] += b[78 + bi * 6 + l];
        int o = 0;
        for (int i = 0; i < 12; ++i)
          for (int j = i; j < 12; ++j, ++o) {
            const int bi = i / 6, bj = j / 6;
            if (fr[bi] < 0 || fr[bj] < 0) continue;
            const int r = fr[bi] * 6 + i % 6, c2 = fr[bj] * 6 + j % 6;
            // lower block triangle; inside a diagonal block both triangles are kept (the factorisation reads the lower one)
            if (fr[bi] > fr[bj]) H[(size_t)r * nn + c2] += b[o];
            else if (fr[bi] < fr[bj]) H[(size_t)c2 * nn + r] += b[o];
            else { H[(size_t)r * nn + c2] += b[o]; if (r != c2) H[(size_t)c2 * nn + r] += b[o]; }
          }
      }
      return cost;
    }
    for (int e = 0; e < E; ++e) {
      if (fidx[src[e]] < 0) continue;   // fixed source: edge excluded (see above)
      const double* b = blocks + (size_t)e * MVICP_EDGE_BLOCK;
      cost += b[90];
      int o = 0;
      for (int i = 0; i < 12; ++i) for (int j = i; j < 12; ++j) { Hc[i][j] = b[o]; Hc[j][i] = b[o]; ++o; }
      const int fr[2] = {fidx[src[e]], fidx[dst[e]]};
      const double* Mb[2] = {&M[(size_t)src[e] * 36], &M[(size_t)dst[e] * 36]};
      for (int bi = 0; bi < 2; ++bi) {
        if (fr[bi] < 0) continue;
        // g' = M^T g
        for (int l = 0; l < 6; ++l) {
          double s = 0.0;
          for (int k = 0; k < 6; ++k) s += Mb[bi][k * 6 + l] * b[78 + bi * 6 + k];
          g[fr[bi] * 6 + l] += s;
        }
        for (int bj = 0; bj < 2; ++bj) {
          if (fr[bj] < 0 || fr[bj] > fr[bi]) continue;   // lower block triangle only
          // Hl = Mi^T Hc[bi][bj] Mj
          for (int k = 0; k < 6; ++k)
            for (int l = 0; l < 6; ++l) {
              double s = 0.0;
              for (int q = 0; q < 6; ++q) s += Hc[bi * 6 + k][bj * 6 + q] * Mb[bj][q * 6 + l];
              T[k][l] = s;
            }
          for (int k = 0; k < 6; ++k)
            for (int l = 0; l < 6; ++l) {
              double s = 0.0;
              for (int q = 0; q < 6; ++q) s += Mb[bi][q * 6 + k] * T[q][l];
              Hl[k][l] = s;
            }
          for (int k = 0; k < 6; ++k) {
            double* row = H + (size_t)(fr[bi] * 6 + k) * nn + fr[bj] * 6;
            for (int l = 0; l < 6; ++l) row[l] += Hl[k][l];
          }
        }
      }
    }
    return cost;
  }
};

// Envelope (skyline) Cholesky: row i of the lower triangle is only touched from its first structural non-zero
// column env[i]; fill-in stays inside the envelope.  The pose graph of a turntable sequence (knn ring neighbours,
// frame 0 eliminated) is block-banded, so the factorisation costs O(n b^2) instead of O(n^3 / 3); for an arbitrary
// graph the envelope degenerates to the dense triangle and this is plain dense Cholesky.  (The reference hands the
// same system to Ceres' SPARSE_NORMAL_CHOLESKY, icp-ceres.cpp:76.)
bool cholesky_solve(std::vector<double>& A, int n, const int* env, const double* b, double* y) {
  for (int i = 0; i < n; ++i) {
    double* Ai = &A[(size_t)i * n];
    for (int j = env[i]; j <= i; ++j) {
      const double* Aj = &A[(size_t)j * n];
      const int k0 = std::max(env[i], env[j]);
      double v = Ai[j];
      for (int k = k0; k < j; ++k) v -= Ai[k] * Aj[k];
      if (j < i) {
        Ai[j] = v / Aj[j];
      } else {
        if (!(v > 0.0) || !std::isfinite(v)) return false;
        Ai[i] = std::sqrt(v);
      }
    }
  }
  for (int i = 0; i < n; ++i) {
    const double* Ai = &A[(size_t)i * n];
    double v = b[i];
    for (int k = env[i]; k < i; ++k) v -= Ai[k] * y[k];
    y[i] = v / Ai[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    y[i] /= A[(size_t)i * n + i];
    const double yi = y[i];
    const double* Ai = &A[(size_t)i * n];
    for (int k = env[i]; k < i; ++k) y[k] -= Ai[k] * yi;
  }
  for (int i = 0; i < n; ++i) if (!std::isfinite(y[i])) return false;
  return true;
}

}  // namespace

int lm_solve(int K, int E, const int* src, const int* dst, double* poses, unsigned char* fixed, int param, int max_iterations,
             mvicp_eval_fn eval, void* user, mvicp_summary* sm) {
  if (K <= 0 || !poses || !fixed || !eval || !sm) { set_error("lm_solve: bad arguments"); return MVICP_ERR_ARG; }
  if (param < 0 || param > 2) { set_error("unknown parameterization %d", param); return MVICP_ERR_ARG; }
  fixed[0] = 1;  // icp-ceres.cpp:244,341,417
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double max_radius = 1e16, min_radius = 1e-32, min_relative_decrease = 1e-3, min_diag = 1e-6, max_diag = 1e32;
  const int max_invalid = 5;
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int consecutive_invalid = 0;
  std::memset(sm, 0, sizeof(*sm));

  const int A = se3::ambient(param);
  Assembler as(K, E, src, dst, fixed, param);
  const int n = as.n();
  static thread_local std::vector<double> ws_x, ws_xc, ws_pc, ws_blocks;   // reused across solves (no allocation / zeroing per ICP round)
  ws_x.resize((size_t)K * A); ws_xc.resize((size_t)K * A); ws_pc.resize((size_t)K * 16); ws_blocks.resize((size_t)std::max(E, 1) * MVICP_EDGE_BLOCK);
  std::vector<double>& x = ws_x; std::vector<double>& xc = ws_xc; std::vector<double>& pc = ws_pc; std::vector<double>& blocks = ws_blocks;
  for (int i = 0; i < K; ++i) se3::pose_to_x(param, poses + 16 * (size_t)i, &x[(size_t)i * A]);
  auto poses_of = [&](const std::vector<double>& xv, double* P) { for (int i = 0; i < K; ++i) se3::x_to_pose(param, &xv[(size_t)i * A], P + 16 * (size_t)i); };
  auto xnorm = [&](const std::vector<double>& v) { double s = 0; for (int i = 0; i < K; ++i) if (!fixed[i]) for (int a = 0; a < A; ++a) s += v[i * A + a] * v[i * A + a]; return std::sqrt(s); };

  // n x n work matrices live in a per-thread workspace: only their envelope is ever touched, so they are neither zeroed nor
  // reallocated between solves (an ICP round calls this once)
  static thread_local std::vector<double> wsH, wsHn, wsHs, wsAw;
  const size_t n2 = (size_t)n * n;
  if (wsH.size() < n2) { wsH.resize(n2); wsHn.resize(n2); wsHs.resize(n2); wsAw.resize(n2); }
  std::vector<double>& H = wsH; std::vector<double>& Hn = wsHn; std::vector<double>& Hs = wsHs; std::vector<double>& Aw = wsAw;
  std::vector<double> g(n), gn(n), scale(n), gs(n), diag(n), step(n), delta(n);
  // structural envelope of the normal matrix from the pose graph: block row b starts at its lowest-numbered neighbour
  std::vector<int> env(n);
  {
    std::vector<int> first(as.nfree);
    for (int b = 0; b < as.nfree; ++b) first[b] = b;
    for (int e = 0; e < E; ++e) {
      const int a = as.fidx[src[e]], b = as.fidx[dst[e]];
      if (a < 0 || b < 0) continue;
      first[std::max(a, b)] = std::min(first[std::max(a, b)], std::min(a, b));
    }
    for (int b = 0; b < as.nfree; ++b) for (int l = 0; l < 6; ++l) env[b * 6 + l] = first[b] * 6;
  }
  poses_of(x, pc.data());
  MV_CHECK(eval(user, pc.data(), blocks.data()));
  sm->evaluations = 1;
  double cost = as.assemble(x.data(), blocks.data(), H.data(), g.data(), env.data());
  sm->initial_cost = sm->final_cost = cost;
  // Poses go back through the parameterization like the reference's write-back (icp-ceres.cpp:312-321,386-394,472-474) — but only when
  // the solve moved something: x -> pose -> x is not bit-idempotent for every pose (a last-bit 2-cycle), and a solve that ends
  // without taking a step must leave the caller's poses exactly as they were, so that a converged registration is a true fixed
  // point of the round (bit-identical query transforms round after round).
  auto finish = [&]() { if (sm->successful_steps > 0) poses_of(x, poses); sm->final_cost = cost; return MVICP_OK; };
  if (n == 0) { sm->termination = 1; return finish(); }
  double x_norm = xnorm(x);
  auto gmax_of = [&](const std::vector<double>& gv) { double m = 0; for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(gv[i])); return m; };
  for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(H[(size_t)i * n + i]));
  auto rescale = [&]() {
    for (int i = 0; i < n; ++i) { gs[i] = g[i] * scale[i]; for (int j = env[i]; j <= i; ++j) Hs[(size_t)i * n + j] = H[(size_t)i * n + j] * scale[i] * scale[j]; }
  };
  rescale();
  if (gmax_of(g) <= gradient_tolerance) { sm->termination = 1; return finish(); }

  int iter = 0;
  while (true) {
    if (iter >= max_iterations) { sm->termination = 0; break; }
    ++iter;
    sm->iterations = iter;
    if (!reuse_diagonal) for (int i = 0; i < n; ++i) diag[i] = std::min(std::max(Hs[(size_t)i * n + i], min_diag), max_diag);
    for (int i = 0; i < n; ++i) {
      for (int j = env[i]; j <= i; ++j) Aw[(size_t)i * n + j] = Hs[(size_t)i * n + j];
      Aw[(size_t)i * n + i] += diag[i] / radius;
    }
    bool valid = cholesky_solve(Aw, n, env.data(), gs.data(), step.data());
    reuse_diagonal = true;
    double model_cost_change = 0.0;
    if (valid) {
      double sg = 0.0, sHs = 0.0;
      for (int i = 0; i < n; ++i) step[i] = -step[i];
      for (int i = 0; i < n; ++i) {   // step^T Hs step from the lower envelope (Hs is symmetric)
        sg += step[i] * gs[i];
        double t = 0.0;
        for (int j = env[i]; j < i; ++j) t += Hs[(size_t)i * n + j] * step[j];
        sHs += step[i] * (2.0 * t + Hs[(size_t)i * n + i] * step[i]);
      }
      model_cost_change = -(sg + 0.5 * sHs);
      valid = model_cost_change > 0.0;
    }
    if (!valid) {
      if (++consecutive_invalid >= max_invalid) { sm->termination = -1; finish(); set_error("LM: %d consecutive invalid steps", max_invalid); return MVICP_ERR_NUMERIC; }
      radius /= decrease_factor; decrease_factor *= 2.0;
      if (radius < min_radius) { sm->termination = 4; break; }
      continue;
    }
    consecutive_invalid = 0;
    for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];
    for (int i = 0; i < K; ++i) {
      if (as.fidx[i] < 0) { for (int a = 0; a < A; ++a) xc[i * A + a] = x[i * A + a]; continue; }
      se3::plus(param, &x[(size_t)i * A], &delta[as.fidx[i] * 6], &xc[(size_t)i * A]);
    }
    poses_of(xc, pc.data());
    MV_CHECK(eval(user, pc.data(), blocks.data()));
    sm->evaluations++;
    const double cand_cost = as.assemble(xc.data(), blocks.data(), Hn.data(), gn.data(), env.data());
    double sn = 0.0;
    for (int i = 0; i < K; ++i) if (!fixed[i]) for (int a = 0; a < A; ++a) { const double dd = x[i * A + a] - xc[i * A + a]; sn += dd * dd; }
    if (std::sqrt(sn) <= parameter_tolerance * (x_norm + parameter_tolerance)) { sm->termination = 2; break; }
    const double cost_change = cost - cand_cost;
    if (std::fabs(cost_change) <= function_tolerance * cost) { sm->termination = 3; break; }
    const double relative_decrease = cost_change / model_cost_change;
    if (relative_decrease > min_relative_decrease) {
      x.swap(xc);
      x_norm = xnorm(x);
      cost = cand_cost;
      H.swap(Hn); g.swap(gn);
      rescale();
      sm->successful_steps++;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
      radius = std::min(max_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
      if (gmax_of(g) <= gradient_tolerance) { sm->termination = 1; break; }
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0;
      if (radius < min_radius) { sm->termination = 4; break; }
    }
  }
  return finish();
}

}  // namespace mvicp

using namespace mvicp;

namespace {
struct CtxEval { mvicp_ctx* c; int plane, robust; };
int ctx_eval(void* user, const double* poses, double* blocks) {
  CtxEval* u = (CtxEval*)user;
  return evaluate_blocks(u->c, poses, u->plane, u->robust, blocks);
}
}  // namespace

extern "C" {

int mvicp_lm_solve(int n_frames, int n_edges, const int* src, const int* dst, double* poses, unsigned char* fixed, int param, int max_iterations,
                   mvicp_eval_fn eval, void* user, mvicp_summary* summary) try {
  return lm_solve(n_frames, n_edges, src, dst, poses, fixed, param, max_iterations, eval, user, summary);
} MVICP_GUARD_ABI

int mvicp_optimize(mvicp_ctx* c, double* poses, unsigned char* fixed, int param, int point_to_plane, int robust, int max_iterations,
                   mvicp_summary* summary) try {
  if (!c) { set_error("null context"); return MVICP_ERR_ARG; }
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) { set_error("hipSetDevice: %s", hipGetErrorString(e)); return MVICP_ERR_HIP; }
  if (c->E == 0) { set_error("no graph"); return MVICP_ERR_STATE; }
  HostScope hs(c, "host.optimize");
  // what the NEXT search may evaluate ahead of time (api.cpp, speculative first evaluation)
  c->spec_flags_valid = true; c->spec_param = param; c->spec_plane = point_to_plane ? 1 : 0; c->spec_robust = robust ? 1 : 0;
  CtxEval u{c, point_to_plane ? 1 : 0, robust ? 1 : 0};
  const int st = lm_solve(c->n_frames, c->E, c->esrc.data(), c->edst.data(), poses, fixed, param, max_iterations, ctx_eval, &u, summary);
  if (st != MVICP_OK) { c->spec_flags_valid = false; c->spec_ready = false; c->spec2_ready = false; c->last_cand_poses.clear(); }   // a failed solve must not arm the next search's queued evaluation
  if (st == MVICP_OK && summary) {   // feeds the AUTO kernel choice of the next search (api.cpp): RMS residual the solve ended on
    double n = 0.0;
    for (int e = 0; e < c->E; ++e) n += c->h_count[e];
    c->last_rms = n > 0.0 && summary->final_cost >= 0.0 ? std::sqrt(2.0 * summary->final_cost / n) : -1.0;
  }
  if (c->profile) prof_collect_lazy(c);
  return st;
} MVICP_GUARD_ABI

}  // extern "C"
