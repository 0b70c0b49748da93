// Implementation of the Frame / ICP_Ceres mirror (host/frame.h) over the C ABI.
#include "frame.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace mvicp {

static void check(int st) {
  if (st < 0) throw std::runtime_error(std::string("mvicp: ") + mvicp_last_error());
}

Session& Session::get() {
  static Session s;
  return s;
}

void Session::reset() {
  if (ctx) mvicp_destroy(ctx);
  if (side_ctx) mvicp_destroy(side_ctx);
  ctx = nullptr; side_ctx = nullptr; side_owner = nullptr;
  frames_key = nullptr;
  frame_keys.clear();
  last_poses.clear();
}

void Session::invalidate() {
  invalidate_lists();
  frames_key = nullptr;
  frame_keys.clear();
  last_poses.clear();
  side_owner = nullptr;
}

int Session::frame_index(const Frame* f) const {
  for (size_t i = 0; i < frame_keys.size(); ++i) if (frame_keys[i].f == f) return (int)i;
  return -1;
}

void Session::bind(std::vector<std::shared_ptr<Frame>>& frames) {
  // graph in the reference's loop order: src ascending, neighbour order (main_multiview.cpp:119-127)
  std::vector<int> s, d;
  for (size_t i = 0; i < frames.size(); ++i)
    for (const OutgoingEdge& e : frames[i]->neighbours) { s.push_back((int)i); d.push_back(e.neighbourIdx); }
  std::vector<FrameKey> keys(frames.size());
  for (size_t i = 0; i < frames.size(); ++i) {
    const Frame& f = *frames[i];
    keys[i] = FrameKey{&f, f.pts.empty() ? nullptr : (const void*)f.pts[0].data(), f.pts.size(), f.nor.empty() ? nullptr : (const void*)f.nor[0].data(), f.version};
  }
  if (ctx && frames_key == (const void*)&frames && s == esrc && d == edst && keys == frame_keys) return;
  if (!ctx) check(mvicp_create(device, &ctx));
  check(mvicp_set_num_frames(ctx, (int)frames.size()));
  for (size_t i = 0; i < frames.size(); ++i) {
    Frame& f = *frames[i];
    const double* nrm = f.nor.size() == f.pts.size() && !f.nor.empty() ? f.nor[0].data() : nullptr;
    check(mvicp_set_frame(ctx, (int)i, f.pts.empty() ? nullptr : f.pts[0].data(), nrm, (int)f.pts.size()));
  }
  check(mvicp_set_graph(ctx, (int)s.size(), s.data(), d.data()));
  esrc = s; edst = d;
  frames_key = (const void*)&frames;
  frame_keys = keys;
  last_poses.clear();
  held.assign(esrc.size(), Held());
  epochs = nullptr;
}

void Session::correspond(std::vector<std::shared_ptr<Frame>>& frames, float thresh) {
  bind(frames);
  std::vector<double> P(16 * frames.size());
  std::vector<unsigned char> fx(frames.size());
  for (size_t i = 0; i < frames.size(); ++i) { std::memcpy(&P[16 * i], frames[i]->pose.data(), 128); fx[i] = frames[i]->fixed; }
  if (P == last_poses && thresh == last_thresh && fx == last_fixed) return;  // same round: the batched result is still valid
  counts.assign(esrc.size(), 0);
  weights.assign(esrc.size(), 0.f);
  check(mvicp_correspond(ctx, P.data(), fx.data(), thresh, nn_method, counts.data(), weights.data()));
  last_poses = P;
  last_thresh = thresh;
  last_fixed = fx;
  corr = nullptr; corr_off = nullptr;
  // the literal drop-in contract: every list in the reference's layout, un-sorted on the device, ONE copy per round for all edges
  if (copy_back) {
    check(mvicp_map_correspondences_async(ctx, &corr, &corr_off));   // (no device work when the search reproduced every list: mvicp.h); chunks arrive frame by frame
    check(mvicp_correspondence_epochs(ctx, &epochs));
  }
}

void Session::optimize(std::vector<std::shared_ptr<Frame>>& frames, int param, bool pointToPlane, bool robust, mvicp_summary* out) {
  bind(frames);
  std::vector<double> P(16 * frames.size());
  std::vector<unsigned char> fx(frames.size());
  for (size_t i = 0; i < frames.size(); ++i) { std::memcpy(&P[16 * i], frames[i]->pose.data(), 128); fx[i] = frames[i]->fixed; }
  mvicp_summary sm;
  check(mvicp_optimize(ctx, P.data(), fx.data(), param, pointToPlane, robust, 50 /* icp-ceres.cpp:81 */, &sm));
  if (!frames.empty()) frames[0]->fixed = true;  // icp-ceres.cpp:244,341,417
  for (size_t i = 0; i < frames.size(); ++i) std::memcpy(frames[i]->pose.data(), &P[16 * i], 128);
  if (out) *out = sm;
}

void Frame::computePoseNeighboursKnn(std::vector<std::shared_ptr<Frame>>* frames, int i, int k) {
  neighbours.clear();
  const Vector3d t1 = pose.translation();
  for (int j = 0; j < (int)frames->size(); ++j) {
    if (i == j) continue;
    const float diff_tra = (float)(t1 - (*frames)[j]->pose.translation()).norm();
    neighbours.push_back(OutgoingEdge{j, diff_tra, {}});
  }
  auto less = [](const OutgoingEdge& a, const OutgoingEdge& b) { return a.weight < b.weight; };
  if ((int)neighbours.size() < k) {
    std::sort(neighbours.begin(), neighbours.end(), less);
  } else {
    std::partial_sort(neighbours.begin(), neighbours.begin() + k, neighbours.end(), less);
    neighbours.resize(k);
  }
}

// dst[i] <- src[i] for a few (destination, source, bytes) pieces on up to `threads` host threads: a frame's lists are a few megabytes
// each; one thread copies at ~5 GB/s, the memory system takes eight such streams.  The workers are PERSISTENT (created on first use, parked
// on a condition variable): a driver calls this once per frame per round, and creating eight threads per call cost more than the copy.
namespace {
struct CopyJob { char* d; const char* s; size_t n; };
class CopyPool {
 public:
  static CopyPool& get() { static CopyPool p; return p; }
  void run(const std::vector<CopyJob>& jobs, int threads) {
    const int nt = (int)std::min<size_t>((size_t)std::max(1, threads), jobs.size());
    if (nt <= 1) { for (const CopyJob& j : jobs) std::memcpy(j.d, j.s, j.n); return; }
    std::unique_lock<std::mutex> lk(m_);
    while ((int)workers_.size() < nt - 1) workers_.emplace_back([this]() { loop(); });
    jobs_ = &jobs; next_.store(0); pending_ = (int)workers_.size(); ++epoch_;
    cv_.notify_all();
    lk.unlock();
    drain();                                   // the caller is a worker too
    lk.lock();
    done_.wait(lk, [this]() { return pending_ == 0; });
    jobs_ = nullptr;
  }
  ~CopyPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++epoch_; }
    cv_.notify_all();
    for (std::thread& t : workers_) t.join();
  }
 private:
  void drain() {
    for (;;) {
      const size_t k = next_.fetch_add(1);
      if (k >= jobs_->size()) return;
      const CopyJob& j = (*jobs_)[k];
      std::memcpy(j.d, j.s, j.n);
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&]() { return epoch_ != seen; });
      seen = epoch_;
      if (stop_) return;
      lk.unlock();
      drain();
      lk.lock();
      if (--pending_ == 0) done_.notify_one();
    }
  }
  std::mutex m_; std::condition_variable cv_, done_;
  std::vector<std::thread> workers_;
  const std::vector<CopyJob>* jobs_ = nullptr;
  std::atomic<size_t> next_{0};
  int pending_ = 0; unsigned long long epoch_ = 0; bool stop_ = false;
};
}  // namespace

static void parallel_copy(const std::vector<std::pair<std::pair<char*, const char*>, size_t>>& pieces, int threads) {
  size_t total = 0;
  for (const auto& p : pieces) total += p.second;
  const size_t chunk = 256 << 10;
  std::vector<CopyJob> jobs;
  for (const auto& p : pieces)
    for (size_t o = 0; o < p.second; o += chunk) jobs.push_back(CopyJob{p.first.first + o, p.first.second + o, std::min(chunk, p.second - o)});
  CopyPool::get().run(jobs, total < (1u << 20) ? 1 : threads);
}

void Frame::computeClosestPointsToNeighbours(std::vector<std::shared_ptr<Frame>>* frames, float thresh) {
  if (fixed) return;  // frame.cpp:93
  Session& S = Session::get();
  S.correspond(*frames, thresh);
  size_t e = 0;
  for (size_t i = 0; i < frames->size(); ++i) {
    Frame& f = *(*frames)[i];
    if (&f != this) { e += f.neighbours.size(); continue; }
    std::vector<std::pair<std::pair<char*, const char*>, size_t>> pieces;
    for (OutgoingEdge& edge : f.neighbours) {
      edge.weight = S.weights[e];
      const size_t n = S.copy_back && S.corr ? (size_t)(S.corr_off[e + 1] - S.corr_off[e]) : 0;
      // frame.cpp:110,158: clear(), then one push_back per kept pair — here the finished slice of the mapped triples (same layout).  A list
      // the vector already holds (same epoch of the library's list, same buffer, same length) is left as it is: after the registration
      // has converged every round reproduces every list, and re-copying cfg4's 198 MB of unchanged triples was 2.3 ms per round
      Session::Held* h = S.copy_back && S.corr && S.epochs && e < S.held.size() ? &S.held[e] : nullptr;
      if (h && h->epoch == S.epochs[e] && h->epoch != 0 && h->n == n && edge.correspondances.size() == n && (const void*)edge.correspondances.data() == h->data) {
        ++S.edges_skipped; ++e; continue;
      }
      edge.correspondances.resize(n);
      if (n) check(mvicp_wait_correspondences(S.ctx, (int)e));   // this edge's chunk has landed (later frames' chunks are still on the bus)
      if (n) pieces.push_back(std::make_pair(std::make_pair((char*)edge.correspondances.data(), (const char*)(S.corr + S.corr_off[e])), n * sizeof(Correspondance)));
      if (h) { h->epoch = S.epochs[e]; h->data = (const void*)edge.correspondances.data(); h->n = n; }
      ++S.edges_copied;
      ++e;
    }
    parallel_copy(pieces, S.copy_threads);
    break;
  }
}

void Frame::recomputeNormals() {
  if (pts.size() < 10) throw std::runtime_error("mvicp: recomputeNormals needs >= 10 points");
  mvicp_ctx* c = nullptr;
  check(mvicp_create(Session::get().device, &c));
  nor.resize(pts.size());
  int st = mvicp_set_num_frames(c, 1);
  if (st == MVICP_OK) st = mvicp_set_frame(c, 0, pts[0].data(), nullptr, (int)pts.size());
  if (st == MVICP_OK) st = mvicp_recompute_normals(c, 0, 10, nor[0].data(), nullptr);
  mvicp_destroy(c);
  check(st);
  ++version;   // a bound session re-uploads this cloud (new normals) at its next bind
}

const std::vector<int>& Frame::getNeighbourIndices(size_t num_results) {
  if (pts.empty()) throw std::runtime_error("mvicp: getNeighbours on an empty cloud (nanoflann throws here: nanoflann.hpp:904)");
  if (num_results < 3 || num_results > 16 || num_results > pts.size())
    throw std::runtime_error("mvicp: getNeighbours supports 3 <= num_results <= 16 (and <= the cloud's size); the reference asks for 10 (frame.cpp:249)");
  if (knn_k_ == num_results && knn_pts_ == (const void*)pts[0].data() && knn_n_ == pts.size()) return knn_table_;
  mvicp_ctx* c = nullptr;
  check(mvicp_create(Session::get().device, &c));
  std::vector<int> table(pts.size() * num_results);
  int st = mvicp_set_num_frames(c, 1);
  if (st == MVICP_OK) st = mvicp_set_frame(c, 0, pts[0].data(), nullptr, (int)pts.size());
  if (st == MVICP_OK) st = mvicp_recompute_normals(c, 0, (int)num_results, nullptr, table.data());
  mvicp_destroy(c);
  check(st);
  knn_table_.swap(table); knn_k_ = num_results; knn_pts_ = (const void*)pts[0].data(); knn_n_ = pts.size();
  return knn_table_;
}

std::vector<Vector3d> Frame::getNeighbours(int queryIdx, size_t num_results) {
  if (queryIdx < 0 || (size_t)queryIdx >= pts.size()) throw std::runtime_error("mvicp: getNeighbours: queryIdx out of range");
  const std::vector<int>& t = getNeighbourIndices(num_results);
  std::vector<Vector3d> out;
  out.reserve(num_results);
  for (size_t i = 0; i < num_results; ++i) out.push_back(pts[t[(size_t)queryIdx * num_results + i]]);   // frame.cpp:222-226
  return out;
}

mvicp_ctx* Session::query_context(Frame* f, int* slot) {
  // frame.cpp:187-206.  The reference builds the frame's KD-tree lazily on first use; here the frame's structure already lives in the
  // bound session (uploaded by computeClosestPointsToNeighbours / ceresOptimizer*), found by frame index.  A frame that is not part
  // of the bound vector gets a one-cloud side context owned by the session (built on first use, like the reference's lazy tree;
  // rebuilt when the cloud changes).
  if (f->pts.empty()) throw std::runtime_error("mvicp: getClosestPoint on an empty cloud (nanoflann throws here: nanoflann.hpp:904)");
  const int fi = ctx ? frame_index(f) : -1;
  if (fi >= 0 && frame_keys[fi].pts == (const void*)f->pts[0].data() && frame_keys[fi].n == f->pts.size()) { *slot = fi; return ctx; }
  if (side_owner != f || side_version != f->version || !side_ctx) {
    if (!side_ctx) check(mvicp_create(device, &side_ctx));
    check(mvicp_set_num_frames(side_ctx, 1));
    check(mvicp_set_frame(side_ctx, 0, f->pts[0].data(), nullptr, (int)f->pts.size()));
    side_owner = f; side_version = f->version;
  }
  *slot = 0;
  return side_ctx;
}

double Frame::getClosestPoint(const Vector3d& q, size_t& ret_index) {
  int slot = 0, idx = -1;
  double d2 = 0.0;
  mvicp_ctx* c = Session::get().query_context(this, &slot);
  check(mvicp_nn_query(c, slot, q.data(), 1, MVICP_NN_AUTO, &idx, &d2));
  ret_index = (size_t)idx;
  return d2;
}

std::vector<double> Frame::getClosestPoints(const std::vector<Vector3d>& q, std::vector<size_t>& ret_index) {
  int slot = 0;
  mvicp_ctx* c = Session::get().query_context(this, &slot);
  const int n = (int)q.size();
  std::vector<double> d2(q.size());
  std::vector<int> idx(q.size());
  ret_index.resize(q.size());
  if (n == 0) return d2;
  check(mvicp_nn_query(c, slot, q[0].data(), n, MVICP_NN_AUTO, idx.data(), d2.data()));
  for (int i = 0; i < n; ++i) ret_index[i] = (size_t)idx[i];
  return d2;
}

}  // namespace mvicp

namespace ICP_Ceres {

void ceresOptimizer(std::vector<std::shared_ptr<Frame>>& frames, bool pointToPlane, bool robust) {
  mvicp::Session::get().optimize(frames, MVICP_PARAM_EIGEN_QUATERNION, pointToPlane, robust);
}
void ceresOptimizer_ceresAngleAxis(std::vector<std::shared_ptr<Frame>>& frames, bool pointToPlane, bool robust) {
  mvicp::Session::get().optimize(frames, MVICP_PARAM_ANGLE_AXIS, pointToPlane, robust);
}
void ceresOptimizer_sophusSE3(std::vector<std::shared_ptr<Frame>>& frames, bool pointToPlane, bool robust, bool) {
  mvicp::Session::get().optimize(frames, MVICP_PARAM_SOPHUS_SE3, pointToPlane, robust);
}

// icp-ceres.cpp:137-218,525-565: one residual block per index-aligned pair (dst[i], src[i]), single free pose from identity,
// no loss function.  Frame 0 = dst (fixed, identity), frame 1 = src.
static Isometry3d pairwise(std::vector<Vector3d>& src, std::vector<Vector3d>& dst, std::vector<Vector3d>* nor, int param) {
  auto check = [](int st) { if (st < 0) throw std::runtime_error(std::string("mvicp: ") + mvicp_last_error()); };
  if (src.size() != dst.size() || src.empty()) throw std::runtime_error("mvicp: pairwise needs equally sized, non-empty clouds");
  mvicp_ctx* c = nullptr;
  check(mvicp_create(mvicp::Session::get().device, &c));
  check(mvicp_set_num_frames(c, 2));
  check(mvicp_set_frame(c, 0, dst[0].data(), nor ? (*nor)[0].data() : nullptr, (int)dst.size()));
  check(mvicp_set_frame(c, 1, src[0].data(), nullptr, (int)src.size()));
  const int s = 1, d = 0;
  check(mvicp_set_graph(c, 1, &s, &d));
  std::vector<int> id(src.size());
  std::iota(id.begin(), id.end(), 0);
  check(mvicp_set_correspondences(c, 0, (int)id.size(), id.data(), id.data(), 0.f));
  double P[32];
  Isometry3d I;
  std::memcpy(P, I.data(), 128);
  std::memcpy(P + 16, I.data(), 128);
  unsigned char fixed[2] = {1, 0};
  mvicp_summary sm;
  const int st = mvicp_optimize(c, P, fixed, param, nor != nullptr, 0, 50, &sm);
  Isometry3d out;
  std::memcpy(out.data(), P + 16, 128);
  mvicp_destroy(c);
  check(st);
  return out;
}
Isometry3d pointToPoint_EigenQuaternion(std::vector<Vector3d>& s, std::vector<Vector3d>& d) { return pairwise(s, d, nullptr, MVICP_PARAM_EIGEN_QUATERNION); }
Isometry3d pointToPoint_CeresAngleAxis(std::vector<Vector3d>& s, std::vector<Vector3d>& d) { return pairwise(s, d, nullptr, MVICP_PARAM_ANGLE_AXIS); }
Isometry3d pointToPoint_SophusSE3(std::vector<Vector3d>& s, std::vector<Vector3d>& d, bool) { return pairwise(s, d, nullptr, MVICP_PARAM_SOPHUS_SE3); }
Isometry3d pointToPlane_EigenQuaternion(std::vector<Vector3d>& s, std::vector<Vector3d>& d, std::vector<Vector3d>& n) { return pairwise(s, d, &n, MVICP_PARAM_EIGEN_QUATERNION); }
Isometry3d pointToPlane_CeresAngleAxis(std::vector<Vector3d>& s, std::vector<Vector3d>& d, std::vector<Vector3d>& n) { return pairwise(s, d, &n, MVICP_PARAM_ANGLE_AXIS); }
Isometry3d pointToPlane_SophusSE3(std::vector<Vector3d>& s, std::vector<Vector3d>& d, std::vector<Vector3d>& n, bool) { return pairwise(s, d, &n, MVICP_PARAM_SOPHUS_SE3); }

}  // namespace ICP_Ceres

namespace ICP_Closedform {
static void check(int st) { if (st < 0) throw std::runtime_error(std::string("mvicp: ") + mvicp_last_error()); }
Isometry3d pointToPoint(std::vector<Vector3d>& src, std::vector<Vector3d>& dst) {
  if (src.size() != dst.size() || src.empty()) throw std::runtime_error("mvicp: closed form needs equally sized, non-empty clouds");
  Isometry3d T;
  check(mvicp_closedform_point_to_point(src[0].data(), dst[0].data(), (int)src.size(), T.data()));
  return T;
}
Isometry3d pointToPlane(std::vector<Vector3d>& src, std::vector<Vector3d>& dst, std::vector<Vector3d>& nor) {
  if (src.size() != dst.size() || src.size() != nor.size() || src.empty()) throw std::runtime_error("mvicp: closed form needs equally sized, non-empty clouds");
  Isometry3d T;
  check(mvicp_closedform_point_to_plane(src[0].data(), dst[0].data(), nor[0].data(), (int)src.size(), T.data()));
  return T;
}
}  // namespace ICP_Closedform
