// Tie fix-up of the 1-NN kernels: the reference's own exact search, on the reference's own tree, for the (rare) queries whose best
// distance is met by more than one target.  See nn_tie.h.
//
// nn_tie_kernel restates nanoflann's findNeighbors / searchLevel for a result set of capacity 1 (include/nanoflann.hpp:900-911,
// 1199-1247, 75-134 without NANOFLANN_FIRST_MATCH): initial per-axis distances to the root box (computeInitialDistances, :1177-1193),
// leaf points replace the best only when strictly nearer (:1209), the near child is the low one iff (val - divlow) + (val - divhigh) < 0
// (:1222-1233), the far child is entered iff mindistsq * epsError <= worstDist with epsError = 1 (:1240), with the reference's
// expressions in the reference's order (this TU is built with -ffp-contract=off).  The tree is kdvisit.h's restatement of the split
// structure buildIndex produces with leaf_max_size = 1 (frame.cpp:189), the one normals.hip already uses for the k-NN tie order.
#include <algorithm>
#include <thread>

#include "kdvisit.h"
#include "nn_tie.h"

namespace mvicp {

namespace {

constexpr int TIE_STACK = 128;   // pending subtrees: at most one per level of the descent; a balanced tree of 2^30 points has 30 levels, skewed clouds more
                                 // (40 B each, in scratch memory: 5 KB per lane of this small kernel only)

__device__ __forceinline__ void xf_point(const double* __restrict__ x, double p0, double p1, double p2, double& q0, double& q1, double& q2) {
  double g[3], u[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    g[i] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(x[i], p0), __dmul_rn(x[i + 3], p1)), __dmul_rn(x[i + 6], p2)), x[9 + i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = __dsub_rn(g[i], x[21 + i]);
  q0 = __dadd_rn(__dadd_rn(__dmul_rn(x[12 + 0], u[0]), __dmul_rn(x[12 + 3], u[1])), __dmul_rn(x[12 + 6], u[2]));
  q1 = __dadd_rn(__dadd_rn(__dmul_rn(x[12 + 1], u[0]), __dmul_rn(x[12 + 4], u[1])), __dmul_rn(x[12 + 7], u[2]));
  q2 = __dadd_rn(__dadd_rn(__dmul_rn(x[12 + 2], u[0]), __dmul_rn(x[12 + 5], u[1])), __dmul_rn(x[12 + 8], u[2]));
}

struct Pending { int node, first; double mind, d0, d1, d2; };   // a subtree still to enter, with the state searchLevel would enter it in

// nanoflann's answer for one query: original index of the neighbour (-1: empty tree) and its squared distance
__device__ int reference_search(const TieJob& J, double qx, double qy, double qz, double* d2_out) {
  const VisitNode* nodes = static_cast<const VisitNode*>(J.nodes);
  Pending st[TIE_STACK];
  int sp = 0;
  double worst = 1.7976931348623157e308;   // KNNResultSet::init: dists[capacity - 1] = max
  int bi = -1;
  {
    // computeInitialDistances: per axis the squared distance to the root box, summed in axis order
    const double q[3] = {qx, qy, qz};
    double d[3], s = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      d[a] = 0.0;
      if (q[a] < J.box[a]) d[a] = __dmul_rn(__dsub_rn(q[a], J.box[a]), __dsub_rn(q[a], J.box[a]));
      if (q[a] > J.box[3 + a]) d[a] = __dmul_rn(__dsub_rn(q[a], J.box[3 + a]), __dsub_rn(q[a], J.box[3 + a]));
      s = __dadd_rn(s, d[a]);
    }
    st[sp++] = Pending{0, 0, s, d[0], d[1], d[2]};
  }
  while (sp > 0) {
    Pending f = st[--sp];
    if (f.node < 0) {
      // a far child: entered iff mindistsq * epsError <= worstDist NOW — i.e. after the near subtree (everything that was above this entry
      // on the stack) has been searched, exactly where the recursive form evaluates the test
      if (!(f.mind <= worst)) continue;
      f.node = -f.node - 1;
    }
    const VisitNode nd = nodes[f.node];
    if (nd.axis < 0) {                       // leaf: slots [first, split)
      for (int s = f.first; s < nd.split; ++s) {
        const int idx = J.ord[s];
        const double* p = J.tpts + 3 * (size_t)idx;
        const double e0 = __dsub_rn(qx, p[0]), e1 = __dsub_rn(qy, p[1]), e2 = __dsub_rn(qz, p[2]);
        const double dist = __dadd_rn(__dadd_rn(__dmul_rn(e0, e0), __dmul_rn(e1, e1)), __dmul_rn(e2, e2));
        if (dist < worst) { worst = dist; bi = idx; }
      }
      continue;
    }
    const double val = nd.axis == 0 ? qx : nd.axis == 1 ? qy : qz;
    const double diff1 = __dsub_rn(val, nd.lo_cut), diff2 = __dsub_rn(val, nd.hi_cut);
    const bool low_first = __dadd_rn(diff1, diff2) < 0.0;
    const double cut = low_first ? __dmul_rn(diff2, diff2) : __dmul_rn(diff1, diff1);   // accum_dist(val, divhigh | divlow)
    const double dst = nd.axis == 0 ? f.d0 : nd.axis == 1 ? f.d1 : f.d2;
    Pending far = f;
    far.node = -(low_first ? nd.right : f.node + 1) - 1;   // (negative: "test mindistsq when popped"; child ids are >= 1)
    far.first = low_first ? nd.split : f.first;
    far.mind = __dsub_rn(__dadd_rn(f.mind, cut), dst);
    if (nd.axis == 0) far.d0 = cut; else if (nd.axis == 1) far.d1 = cut; else far.d2 = cut;
    Pending near = f;
    near.node = low_first ? f.node + 1 : nd.right;
    near.first = low_first ? f.first : nd.split;
    if (sp + 2 > TIE_STACK) break;           // (deeper than any tree this was built for: keep what was found so far)
    st[sp++] = far;
    st[sp++] = near;
  }
  *d2_out = worst;
  return bi;
}

__global__ __launch_bounds__(64) void nn_tie_kernel(const TieJob* __restrict__ jobs, int n_jobs, const unsigned long long* __restrict__ list,
                                                    const unsigned int* __restrict__ count, unsigned int* __restrict__ next_count, unsigned int cap,
                                                    double bound, unsigned int* __restrict__ seen) {
  const unsigned int reported = *count;
  if (blockIdx.x == 0 && threadIdx.x == 0) { *next_count = 0u; *seen = reported; }   // two counters alternate between launches (like the grid kernel's far list)
  if (reported == 0u) return;
  const bool everything = reported > cap;                       // the list overflowed: re-answer every query of the launch
  const long long total = everything ? jobs[n_jobs - 1].q_begin + jobs[n_jobs - 1].n : (long long)reported;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int jb, i;
    if (everything) {
      int lo = 0, hi = n_jobs - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].q_begin <= t) lo = mid; else hi = mid - 1; }
      jb = lo; i = (int)(t - jobs[lo].q_begin);
    } else {
      const unsigned long long e = list[t];
      jb = (int)(e >> 32); i = (int)(e & 0xffffffffull);
    }
    const TieJob& J = jobs[jb];
    if (i >= J.n) continue;
    if (J.nodes == nullptr) { seen[1] = 1u; continue; }   // lazy trees (api.cpp): this target has none yet — the host builds it and repeats the search
    const int cur = J.out_idx[i];
    if (cur < 0) continue;
    double qx, qy, qz;
    {
      const double p0 = J.q[3 * (size_t)i], p1 = J.q[3 * (size_t)i + 1], p2 = J.q[3 * (size_t)i + 2];
      if (J.xf != nullptr) xf_point(J.xf, p0, p1, p2, qx, qy, qz);   // the same rounded operations as the search kernels
      else { qx = p0; qy = p1; qz = p2; }
    }
    double d2 = 0.0;
    const int bi = reference_search(J, qx, qy, qz, &d2);
    if (bi < 0 || d2 != J.out_d2[i]) continue;   // (cannot happen: both are the exact minimum)
    const int pos = J.inv ? J.inv[bi] : bi;
    if (pos != cur) {
      J.out_idx[i] = pos;
      if (J.list.dirty) update_list_entry(J.list, i, pos, d2, bound, false);
    }
  }
}

}  // namespace

// ---- host ----------------------------------------------------------------------------------------------------------------------------
void free_tie(FrameDev& f) {
  if (f.tie_nodes) (void)hipFree(f.tie_nodes);
  if (f.tie_ord) (void)hipFree(f.tie_ord);
  if (f.tie_slot) (void)hipFree(f.tie_slot);
  f.tie_nodes = nullptr; f.tie_ord = nullptr; f.tie_slot = nullptr; f.has_tie = false;
}

int ensure_tie_trees(mvicp_ctx* c, const std::vector<int>& frames) {
  std::vector<int> todo;
  for (int f : frames)
    if (f >= 0 && f < c->n_frames && !c->frames[f].has_tie && c->frames[f].n > 0 && std::find(todo.begin(), todo.end(), f) == todo.end()) todo.push_back(f);
  if (todo.empty()) return MVICP_OK;
  struct Built { std::vector<double> xyz; std::vector<VisitNode> nodes; std::vector<int> slot, ord; double box[6]; };
  std::vector<Built> B(todo.size());
  for (size_t k = 0; k < todo.size(); ++k) {
    const FrameDev& F = c->frames[todo[k]];
    B[k].xyz.resize(3 * (size_t)F.n);
    MV_HIP(hipMemcpy(B[k].xyz.data(), F.pts, sizeof(double) * 3 * (size_t)F.n, hipMemcpyDeviceToHost));
  }
  // the trees are independent: build them side by side (0.07 s per 200 k points, 0.5 s per 1 M on one core)
  auto work = [&](size_t k) {
    Built& b = B[k];
    const int n = (int)(b.xyz.size() / 3);
    build_visit_tree(b.xyz.data(), n, b.nodes, b.slot);
    b.ord.assign(n, 0);
    for (int i = 0; i < n; ++i) b.ord[b.slot[i]] = i;
    for (int a = 0; a < 3; ++a) b.box[a] = b.box[3 + a] = b.xyz[a];
    for (int i = 1; i < n; ++i)
      for (int a = 0; a < 3; ++a) { const double v = b.xyz[3 * (size_t)i + a]; if (v < b.box[a]) b.box[a] = v; if (v > b.box[3 + a]) b.box[3 + a] = v; }
  };
  const unsigned int nthreads = std::max(1u, std::min<unsigned int>({(unsigned int)todo.size(), std::thread::hardware_concurrency(), 16u}));
  if (nthreads <= 1) { for (size_t k = 0; k < todo.size(); ++k) work(k); }
  else {
    std::vector<std::thread> pool;
    for (unsigned int t = 0; t < nthreads; ++t) pool.emplace_back([&, t]() { for (size_t k = t; k < todo.size(); k += nthreads) work(k); });
    for (auto& th : pool) th.join();
  }
  for (size_t k = 0; k < todo.size(); ++k) {
    FrameDev& F = c->frames[todo[k]];
    auto upload = [&]() -> int {
      MV_HIP(hipMalloc(&F.tie_nodes, sizeof(VisitNode) * B[k].nodes.size()));
      MV_HIP(hipMemcpy(F.tie_nodes, B[k].nodes.data(), sizeof(VisitNode) * B[k].nodes.size(), hipMemcpyHostToDevice));
      MV_HIP(hipMalloc((void**)&F.tie_ord, sizeof(int) * B[k].ord.size()));
      MV_HIP(hipMemcpy(F.tie_ord, B[k].ord.data(), sizeof(int) * B[k].ord.size(), hipMemcpyHostToDevice));
      MV_HIP(hipMalloc((void**)&F.tie_slot, sizeof(int) * B[k].slot.size()));
      MV_HIP(hipMemcpy(F.tie_slot, B[k].slot.data(), sizeof(int) * B[k].slot.size(), hipMemcpyHostToDevice));
      return MVICP_OK;
    };
    const int st = upload();
    if (st != MVICP_OK) { free_tie(F); return st; }   // a half-uploaded tree is released, not leaked at the next attempt
    for (int a = 0; a < 6; ++a) F.tie_box[a] = B[k].box[a];
    F.has_tie = true;
  }
  return MVICP_OK;
}

TieRef tie_ref(mvicp_ctx* c, size_t launch_queries, unsigned int job) {
  TieRef T{nullptr, nullptr, 0u, job};
  if (!c->tie_rule) return T;
  // capacity: every query of the launch up to 4 M entries (32 MB); beyond that an overflow makes the fix-up re-answer the whole launch
  const size_t want = std::min<size_t>(std::max<size_t>(launch_queries, 1024), (size_t)4 << 20);
  if (want > c->tie_cap) {
    if (c->d_tie_list) { if (hipStreamSynchronize(c->stream) != hipSuccess || hipFree(c->d_tie_list) != hipSuccess) return T; c->d_tie_list = nullptr; c->tie_cap = 0; }
    if (hipMalloc((void**)&c->d_tie_list, sizeof(unsigned long long) * want) != hipSuccess) return T;
    c->tie_cap = want;
  }
  if (!c->d_tie_count) {
    if (hipMalloc((void**)&c->d_tie_count, 2 * sizeof(unsigned int)) != hipSuccess) return T;
    if (hipMemset(c->d_tie_count, 0, 2 * sizeof(unsigned int)) != hipSuccess) return T;
    c->tie_parity = 0;
  }
  T.list = c->d_tie_list; T.count = c->d_tie_count + c->tie_parity; T.cap = (unsigned int)c->tie_cap;
  return T;
}

void tie_job_fill(const FrameDev& F, TieJob& j) {
  j.nodes = F.has_tie ? F.tie_nodes : nullptr; j.ord = F.tie_ord; j.tpts = F.pts;
  for (int a = 0; a < 6; ++a) j.box[a] = F.tie_box[a];
}

int launch_tie_fixup(mvicp_ctx* c, const std::vector<TieJob>& jobs, double d2_bound) {
  if (!c->tie_rule || jobs.empty() || !c->d_tie_count) return MVICP_OK;
  if (c->tie_skip) return MVICP_OK;   // (the launch's counter stays zero and keeps its turn)
  if (!c->h_tie_seen) {
    MV_HIP(hipHostMalloc((void**)&c->h_tie_seen, 2 * sizeof(unsigned int), hipHostMallocMapped));   // [0] reports of the launch, [1] "a reported query's target has no tree"
    c->h_tie_seen[0] = 1u;   // unknown until a launch has written it
    c->h_tie_seen[1] = 0u;
    MV_HIP(hipHostGetDevicePointer((void**)&c->d_tie_seen, c->h_tie_seen, 0));
  }
  std::vector<TieJob> tab(jobs);
  long long off = 0;
  for (TieJob& j : tab) { j.q_begin = off; off += j.n; }
  TieJob* d_tab = nullptr;
  MV_CHECK(cached_upload(c, "tie_jobs", tab.data(), sizeof(TieJob) * tab.size(), (void**)&d_tab));
  unsigned int* cnt = c->d_tie_count + c->tie_parity;
  unsigned int* nxt = c->d_tie_count + (c->tie_parity ^ 1);
  c->tie_parity ^= 1;
  c->h_tie_seen[1] = 0u;   // (host store to the mapped word before the launch; the kernel only ever stores 1)
  ProfScope ps(c, "nn_tie", 0.0);
  hipLaunchKernelGGL(nn_tie_kernel, dim3(256), dim3(64), 0, c->stream, d_tab, (int)tab.size(), c->d_tie_list, cnt, nxt, (unsigned int)c->tie_cap, d2_bound, c->d_tie_seen);
  MV_HIP(hipGetLastError());
  return MVICP_OK;
}

}  // namespace mvicp
