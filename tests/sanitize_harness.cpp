// ASan + UBSan harness for the HOST code of the product: host/lm.cpp (the LM solve, its banded factorisation, the
// chain rule to every parameterization), host/closedform.cpp, csrc/kdvisit.h (the nanoflann-equivalent tree builder, explicit stack) and
// the oracle they are checked against.  Built and run by tests/test_sanitizers.py with
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=undefined
// No GPU, no HIP runtime: the two HIP entry points lm.cpp's device wrapper (mvicp_optimize) refers to are stubbed below and never called;
// the solve is driven through mvicp_lm_solve with the oracle as the evaluator, like tests/test_host_lm.py does from Python.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../include/mvicp.h"
#include "../mv-lm-icp_amd/csrc/kdvisit.h"

// ---- what lm.cpp / closedform.cpp expect from the rest of the library (csrc/api.cpp) — host stand-ins ---------------------------------
#include "../mv-lm-icp_amd/csrc/common.h"
namespace mvicp {
static char g_err[512];
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); }
int abi_exception() noexcept { std::snprintf(g_err, sizeof(g_err), "exception"); return MVICP_ERR_INTERNAL; }
int evaluate_blocks(mvicp_ctx*, const double*, int, int, double*) { return MVICP_ERR_STATE; }   // (device evaluator: not in this build)
void prof_collect(mvicp_ctx*) {}
void prof_collect_lazy(mvicp_ctx*) {}
HostScope::HostScope(mvicp_ctx* ctx, const char* nm) : c(ctx), name(nm), t0(0.0), on(false) {}
HostScope::~HostScope() {}
}  // namespace mvicp
extern "C" {
hipError_t hipSetDevice(int) { return hipErrorNoDevice; }
const char* hipGetErrorString(hipError_t) { return "no device in the sanitizer harness"; }
const char* mvicp_last_error(void) { return mvicp::g_err; }
}

// ---- oracle API (oracle/oracle.cpp) ---------------------------------------------------------------------------------------------------
struct orc_problem {
  int K; const double* pts; const double* nor; const int* foff; const unsigned char* fixed;
  int E; const int* esrc; const int* edst; const int* eoff; const int* first; const int* second; const float* eweight;
  int param, plane, robust;
};
struct orc_summary { double initial_cost, final_cost; int iterations, successful_steps, termination, jacobian_evals, cost_evals; };
extern "C" {
double orc_evaluate(const orc_problem* p, const double* poses, double* H, double* g);
void orc_optimize(const orc_problem* p, double* poses, int max_iterations, orc_summary* out);
int orc_correspond_edge(const double* src, int n_src, const double* pose_src, const double* dst, int n_dst, const double* pose_dst, float thresh, int* first,
                        int* second, double* dist, float* weight, int* nn_idx, double* nn_d2);
void orc_add_noise(const double* pose16, double sigma, double sigmat, int reset, double* out16);
void orc_pose_diff(const double* P1, const double* P2, double* diff_tra, double* diff_rot_deg);
}

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } } while (0)

struct Scene {
  int K, N;
  std::vector<std::vector<double>> pts, nor;   // per view, local frame
  std::vector<double> gt, init;                // K x 16 column-major
  std::vector<int> src, dst;
  std::vector<std::vector<int>> first, second;
  std::vector<float> w;
};

static void pose_of(double yaw, double* P) {   // Ry(yaw) * T(0,0,-0.4), column-major 4x4
  const double c = std::cos(yaw), s = std::sin(yaw);
  const double R[9] = {c, 0, -s, 0, 1, 0, s, 0, c};   // column-major
  std::memset(P, 0, 128);
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) P[i + 4 * j] = R[i + 3 * j];
  P[12] = R[6] * -0.4; P[13] = R[7] * -0.4; P[14] = R[8] * -0.4; P[15] = 1.0;
}

static Scene make_scene(int K, int N, unsigned seed) {
  Scene S; S.K = K; S.N = N;
  std::mt19937 g(seed);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::normal_distribution<double> G(0.0, 1e-4);
  S.gt.resize(16 * (size_t)K); S.init.resize(16 * (size_t)K);
  for (int k = 0; k < K; ++k) {
    pose_of(0.2 * k, &S.gt[16 * (size_t)k]);
    if (k == 0) std::memcpy(&S.init[0], &S.gt[0], 128);
    else orc_add_noise(&S.gt[16 * (size_t)k], 0.02, 0.01, k == 1, &S.init[16 * (size_t)k]);
    // points on a bumpy sphere of radius 0.1 m seen from the camera direction, stored in the view's local frame
    std::vector<double> p(3 * (size_t)N), n(3 * (size_t)N);
    const double* P = &S.gt[16 * (size_t)k];
    for (int i = 0; i < N; ++i) {
      double d[3];
      double len;
      do { d[0] = U(g); d[1] = U(g); d[2] = U(g); len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); } while (len < 0.2 || len > 1.0);
      for (double& v : d) v /= len;
      // bias towards the camera side (-z of the rotated frame)
      const double cam[3] = {-P[8], -P[9], -P[10]};
      if (d[0] * cam[0] + d[1] * cam[1] + d[2] * cam[2] < 0) for (double& v : d) v = -v;
      const double r = 0.1 * (1.0 + 0.1 * std::sin(5 * d[0]) * std::cos(4 * d[1]));
      double w[3] = {r * d[0] + G(g), r * d[1] + G(g), r * d[2] + G(g)};
      for (int a = 0; a < 3; ++a) {   // local = R^T (w - t), normal = R^T d
        p[3 * (size_t)i + a] = P[0 + 4 * a] * (w[0] - P[12]) + P[1 + 4 * a] * (w[1] - P[13]) + P[2 + 4 * a] * (w[2] - P[14]);
        n[3 * (size_t)i + a] = P[0 + 4 * a] * d[0] + P[1 + 4 * a] * d[1] + P[2 + 4 * a] * d[2];
      }
    }
    S.pts.push_back(p); S.nor.push_back(n);
  }
  for (int k = 1; k < K; ++k) { S.src.push_back(k); S.dst.push_back(k - 1); if (k + 1 < K) { S.src.push_back(k); S.dst.push_back(k + 1); } }
  for (size_t e = 0; e < S.src.size(); ++e) {
    const int s = S.src[e], d = S.dst[e];
    std::vector<int> f(N), sec(N), nn(N);
    std::vector<double> dist(N), d2(N);
    float w = 0.f;
    const int c = orc_correspond_edge(S.pts[s].data(), N, &S.init[16 * (size_t)s], S.pts[d].data(), N, &S.init[16 * (size_t)d], 0.05f, f.data(), sec.data(), dist.data(), &w,
                                      nn.data(), d2.data());
    CHECK(c > N / 4);
    f.resize(c); sec.resize(c);
    S.first.push_back(f); S.second.push_back(sec); S.w.push_back(w);
  }
  return S;
}

struct EvalCtx { const Scene* S; int plane, robust; };

// per-edge canonical 91-blocks through the oracle: a 2-frame SophusSE3 problem per edge with both poses free (tests/orclib.py edge_blocks)
static int eval_cb(void* user, const double* poses, double* blocks) {
  const EvalCtx* C = (const EvalCtx*)user;
  const Scene& S = *C->S;
  for (size_t e = 0; e < S.src.size(); ++e) {
    const int s = S.src[e], d = S.dst[e], N = S.N;
    std::vector<double> pts(6 * (size_t)N), nor(6 * (size_t)N);
    std::memcpy(&pts[0], S.pts[s].data(), 24 * (size_t)N); std::memcpy(&pts[3 * (size_t)N], S.pts[d].data(), 24 * (size_t)N);
    std::memcpy(&nor[0], S.nor[s].data(), 24 * (size_t)N); std::memcpy(&nor[3 * (size_t)N], S.nor[d].data(), 24 * (size_t)N);
    const int foff[3] = {0, N, 2 * N};
    const unsigned char fixed[2] = {0, 0};
    const int es = 0, ed = 1, eoff[2] = {0, (int)S.first[e].size()};
    orc_problem pb{2, pts.data(), nor.data(), foff, fixed, 1, &es, &ed, eoff, S.first[e].data(), S.second[e].data(), &S.w[e], 2, C->plane, C->robust};
    double P2[32];
    std::memcpy(P2, poses + 16 * (size_t)s, 128); std::memcpy(P2 + 16, poses + 16 * (size_t)d, 128);
    double H[144], g[12];
    const double cost = orc_evaluate(&pb, P2, H, g);
    double* b = blocks + (size_t)e * MVICP_EDGE_BLOCK;
    int t = 0;
    for (int i = 0; i < 12; ++i) for (int j = i; j < 12; ++j) b[t++] = H[i * 12 + j];
    for (int i = 0; i < 12; ++i) b[78 + i] = g[i];
    b[90] = cost;
  }
  return 0;
}

static void solve_case(const Scene& S, int param, int plane, int robust) {
  const int K = S.K, N = S.N, E = (int)S.src.size();
  // the oracle's own solve of the whole problem
  std::vector<double> pts, nor;
  std::vector<int> foff(1, 0), eoff(1, 0), first, second;
  for (int k = 0; k < K; ++k) { pts.insert(pts.end(), S.pts[k].begin(), S.pts[k].end()); nor.insert(nor.end(), S.nor[k].begin(), S.nor[k].end()); foff.push_back(foff.back() + N); }
  for (int e = 0; e < E; ++e) { first.insert(first.end(), S.first[e].begin(), S.first[e].end()); second.insert(second.end(), S.second[e].begin(), S.second[e].end()); eoff.push_back((int)first.size()); }
  std::vector<unsigned char> fixed(K, 0); fixed[0] = 1;
  orc_problem pb{K, pts.data(), nor.data(), foff.data(), fixed.data(), E, S.src.data(), S.dst.data(), eoff.data(), first.data(), second.data(), S.w.data(), param, plane, robust};
  std::vector<double> Pref(S.init);
  orc_summary smr;
  orc_optimize(&pb, Pref.data(), 50, &smr);
  // the product's host solve over oracle blocks
  std::vector<double> P(S.init);
  std::vector<unsigned char> fx(K, 0);
  mvicp_summary sm;
  EvalCtx C{&S, plane, robust};
  const int st = mvicp_lm_solve(K, E, S.src.data(), S.dst.data(), P.data(), fx.data(), param, 50, eval_cb, &C, &sm);
  CHECK(st == MVICP_OK);
  CHECK(sm.iterations == smr.iterations && sm.termination == smr.termination);
  for (int k = 0; k < K; ++k) {
    double dt, dr;
    orc_pose_diff(&P[16 * (size_t)k], &Pref[16 * (size_t)k], &dt, &dr);
    CHECK(dt < 1e-8);
  }
  std::printf("lm param %d plane %d robust %d: %d iterations, termination %d, cost %.6e -> %.6e\n", param, plane, robust, sm.iterations, sm.termination, sm.initial_cost, sm.final_cost);
}

static void closedform_case(const Scene& S) {
  const int N = S.N;
  const double* P = &S.init[16];
  std::vector<double> dst(3 * (size_t)N), dn(3 * (size_t)N);
  for (int i = 0; i < N; ++i)
    for (int a = 0; a < 3; ++a) {
      dst[3 * (size_t)i + a] = P[a] * S.pts[1][3 * (size_t)i] + P[a + 4] * S.pts[1][3 * (size_t)i + 1] + P[a + 8] * S.pts[1][3 * (size_t)i + 2] + P[12 + a];
      dn[3 * (size_t)i + a] = P[a] * S.nor[1][3 * (size_t)i] + P[a + 4] * S.nor[1][3 * (size_t)i + 1] + P[a + 8] * S.nor[1][3 * (size_t)i + 2];
    }
  double T[16], dt, dr;
  CHECK(mvicp_closedform_point_to_point(S.pts[1].data(), dst.data(), N, T) == MVICP_OK);
  orc_pose_diff(T, P, &dt, &dr);
  CHECK(dt < 1e-12);
  CHECK(mvicp_closedform_point_to_plane(S.pts[1].data(), dst.data(), dn.data(), N, T) == MVICP_OK);
  CHECK(mvicp_closedform_point_to_point(S.pts[1].data(), dst.data(), 0, T) < 0);   // error path
  std::printf("closed form ok (point-to-point recovers the transform to %.1e m)\n", dt);
}

static void tree_case(unsigned seed) {
  std::mt19937 g(seed);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  for (int variant = 0; variant < 4; ++variant) {
    const int n = variant == 3 ? 1 : 5000;
    std::vector<double> xyz(3 * (size_t)n);
    for (int i = 0; i < n; ++i)
      for (int a = 0; a < 3; ++a) {
        double v = U(g);
        if (variant == 1) v = std::floor(v * 8) / 8;                               // lattice: duplicates, ties
        if (variant == 2) v = a == 0 ? std::ldexp(1.0, (i % 1800) - 900) : 0.0;    // geometric progression on one axis: a tree as deep as the exponent range
        xyz[3 * (size_t)i + a] = v;
      }
    std::vector<mvicp::VisitNode> nodes; std::vector<int> slot;
    mvicp::build_visit_tree(xyz.data(), n, nodes, slot);
    CHECK((int)nodes.size() == 2 * n - 1);
    std::vector<char> seen(n, 0);
    for (int i = 0; i < n; ++i) { CHECK(slot[i] >= 0 && slot[i] < n && !seen[slot[i]]); seen[slot[i]] = 1; }
    mvicp::VisitTree T{nodes.data(), slot.data()};
    int before = 0;
    for (int t = 0; t < 2000 && n > 1; ++t) {
      const int a = (int)(U(g) * n) % n, b = (a + 1 + (int)(U(g) * (n - 1))) % n;
      const bool ab = mvicp::visited_before(T, U(g), U(g), U(g), a, b);
      before += ab;
    }
    std::printf("tree variant %d: %zu nodes, %d of 2000 pairs visited a-first\n", variant, nodes.size(), before);
  }
}

int main() {
  const Scene S = make_scene(4, 450, 7);
  for (int param = 0; param < 3; ++param)
    for (int plane = 0; plane < 2; ++plane)
      for (int robust = 0; robust < 2; ++robust) solve_case(S, param, plane, robust);
  const Scene S2 = make_scene(7, 250, 11);   // a longer chain: wider band in the factorisation
  solve_case(S2, 2, 1, 1);
  closedform_case(S);
  tree_case(3);
  std::printf("SANITIZE_HARNESS_OK\n");
  return 0;
}
