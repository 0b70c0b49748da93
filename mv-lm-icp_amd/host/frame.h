// Host-side mirror of the reference's Frame / ICP_Ceres interface on top of the C ABI (include/mvicp.h).
// Same member names, argument meaning and call order as the reference, so a driver written against
// include/frame.h + include/icp-ceres.h of adrelino/mv-lm-icp reads the same here:
//   struct Correspondance / OutgoingEdge            include/frame.h:18-29
//   class Frame                                     include/frame.h:31-102
//   ICP_Ceres::ceresOptimizer*, pointToPoint/Plane_* include/icp-ceres.h:29-42
// Differences forced by the boundary: errors are thrown as std::runtime_error carrying mvicp_last_error() (the
// reference has nothing to propagate); OutgoingEdge::P_relative (never used, frame.h:28) is dropped; draw()/GL
// members are out of scope.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/mvicp.h"
#include "linalg.h"

namespace mvicp {

struct Correspondance { int first; int second; double dist; };
static_assert(sizeof(Correspondance) == sizeof(mvicp_corr), "mvicp_corr is the reference's Correspondance (include/frame.h:18-22)");

struct OutgoingEdge {
  int neighbourIdx;
  float weight;  // pose distance at graph build, then 1.5 x median correspondence distance (frame.cpp:176)
  std::vector<Correspondance> correspondances;
};

class Frame {
 public:
  std::vector<Vector3d> pts;
  std::vector<Vector3d> nor;
  bool fixed = false;
  Isometry3d pose;
  Isometry3d poseGroundTruth;
  std::vector<OutgoingEdge> neighbours;
  unsigned long long version = 0;   // bumped by recomputeNormals(): the session re-uploads the cloud when it changes

  // frame.cpp:67-89 (host; tiny)
  void computePoseNeighboursKnn(std::vector<std::shared_ptr<Frame>>* frames, int i, int k);
  // frame.cpp:91-185.  The first call after any pose change runs the batched device search for ALL frames; this
  // frame's edges are then filled from it (later calls for the other frames in the same round are copies only).
  void computeClosestPointsToNeighbours(std::vector<std::shared_ptr<Frame>>* frames, float thresh);
  // frame.cpp:187-206: query in this frame's local coordinates -> squared distance, index
  double getClosestPoint(const Vector3d& query_pt, size_t& ret_index);
  // Batched form of the same call (no counterpart in the reference, whose loop asks one query at a time, frame.cpp:138): ONE device
  // launch + ONE copy for all queries instead of a launch and a synchronisation per query.  ret_index[i] / returned[i] are exactly
  // what getClosestPoint(query_pts[i], ret_index[i]) returns.  Use this in any loop over queries.
  std::vector<double> getClosestPoints(const std::vector<Vector3d>& query_pts, std::vector<size_t>& ret_index);
  // frame.cpp:244-255: PCA normals from the 10 nearest points (self included), n_z <= 0; overwrites `nor`
  void recomputeNormals();
  // frame.cpp:208-231: the num_results nearest points of pts[queryIdx] in this cloud — the point itself first — in the order nanoflann's
  // knnSearch returns them (ascending distance, exact ties in the tree's visiting order).  The reference runs one tree query per call; here the
  // first call for a given num_results answers ALL points of the cloud in one device launch (the k-NN kernel behind recomputeNormals, 3 <= k <= 16)
  // and keeps the index table with the frame, so a loop over queryIdx — the reference's only use, frame.cpp:249 — costs one launch.
  std::vector<Vector3d> getNeighbours(int queryIdx, size_t num_results);
  // the table behind it: pts.size() x num_results original indices, row i = the neighbours of pts[i], nearest (i itself) first
  const std::vector<int>& getNeighbourIndices(size_t num_results);

 private:
  std::vector<int> knn_table_; size_t knn_k_ = 0; const void* knn_pts_ = nullptr; size_t knn_n_ = 0;   // cache of getNeighbourIndices
};

// Process-wide device session behind the Frame / ICP_Ceres calls: uploads the (static) clouds once, mirrors the
// pose graph, caches the batched correspondence search of the current poses.
struct Session {
  static Session& get();
  mvicp_ctx* ctx = nullptr;
  int device = 0;
  bool copy_back = true;  // fill Frame::neighbours[].correspondances after the search (off: device-only, faster)
  int copy_threads = 8;   // host threads that slice the mapped triples into a frame's vectors (1 = the calling thread alone)
  const mvicp_corr* corr = nullptr; const long long* corr_off = nullptr;   // mvicp_map_correspondences of the current search (library-owned, pinned)
  // Copy-back bookkeeping (frame.cpp:110,156-160 clears and refills every list every round; refilling a list with the bytes it already holds
  // is skipped): per edge the library's change counter (mvicp_correspondence_epochs) of the list the Frame's vector was last filled from, and
  // the vector's buffer and length at that time — a caller that resized, cleared or re-allocated the vector gets a fresh copy.  (A caller that
  // overwrites ELEMENTS of a filled list in place and expects the next round to repair them must call Session::invalidate_lists().)
  const unsigned long long* epochs = nullptr;
  struct Held { unsigned long long epoch = 0; const void* data = nullptr; size_t n = 0; };
  std::vector<Held> held;
  unsigned long long edges_copied = 0, edges_skipped = 0;   // statistics of the copy-back (tests, bench)
  void invalidate_lists() { held.assign(held.size(), Held()); }
  int nn_method = MVICP_NN_AUTO;
  const void* frames_key = nullptr;
  // what the device copy was built from: per frame {Frame*, pts data, size, nor data, normals version}; any difference
  // (a re-loaded cloud, recomputeNormals(), an edited point vector) triggers a fresh upload at the next bind
  struct FrameKey { const Frame* f; const void* pts; size_t n; const void* nor; unsigned long long version;
                    bool operator==(const FrameKey& o) const { return f == o.f && pts == o.pts && n == o.n && nor == o.nor && version == o.version; } };
  std::vector<FrameKey> frame_keys;
  std::vector<int> esrc, edst;
  std::vector<double> last_poses;
  std::vector<unsigned char> last_fixed;
  float last_thresh = -1.f;
  mvicp_ctx* side_ctx = nullptr; const Frame* side_owner = nullptr; unsigned long long side_version = 0;  // getClosestPoint on an unbound frame
  int frame_index(const Frame* f) const;   // position of f in the bound vector, or -1
  // context + frame slot that hold `f`'s cloud for raw queries: the bound session if f is part of it, else a one-cloud side context
  mvicp_ctx* query_context(Frame* f, int* slot);
  void invalidate();                       // forget the device copy (forces re-upload + fresh search); for in-place edits of pts/nor data
  std::vector<int> counts;
  std::vector<float> weights;
  void bind(std::vector<std::shared_ptr<Frame>>& frames);      // upload + graph (idempotent)
  void correspond(std::vector<std::shared_ptr<Frame>>& frames, float thresh);
  void optimize(std::vector<std::shared_ptr<Frame>>& frames, int param, bool pointToPlane, bool robust, mvicp_summary* sm = nullptr);
  void reset();
};

}  // namespace mvicp

namespace ICP_Ceres {
using mvicp::Frame;
using mvicp::Isometry3d;
using mvicp::Vector3d;
// multiview (icp-ceres.h:40-42)
void ceresOptimizer(std::vector<std::shared_ptr<Frame>>& frames, bool pointToPlane, bool robust);
void ceresOptimizer_ceresAngleAxis(std::vector<std::shared_ptr<Frame>>& frames, bool pointToPlane, bool robust);
void ceresOptimizer_sophusSE3(std::vector<std::shared_ptr<Frame>>& frames, bool pointToPlane, bool robust, bool automaticDiffLocalParam = true);
// pairwise (icp-ceres.h:30-36): returns the src -> dst transform, starting from identity
Isometry3d pointToPoint_EigenQuaternion(std::vector<Vector3d>& src, std::vector<Vector3d>& dst);
Isometry3d pointToPoint_CeresAngleAxis(std::vector<Vector3d>& src, std::vector<Vector3d>& dst);
Isometry3d pointToPoint_SophusSE3(std::vector<Vector3d>& src, std::vector<Vector3d>& dst, bool automaticDiffLocalParam = true);
Isometry3d pointToPlane_EigenQuaternion(std::vector<Vector3d>& src, std::vector<Vector3d>& dst, std::vector<Vector3d>& nor);
Isometry3d pointToPlane_CeresAngleAxis(std::vector<Vector3d>& src, std::vector<Vector3d>& dst, std::vector<Vector3d>& nor);
Isometry3d pointToPlane_SophusSE3(std::vector<Vector3d>& src, std::vector<Vector3d>& dst, std::vector<Vector3d>& nor, bool automaticDiffLocalParam = true);
}  // namespace ICP_Ceres

namespace ICP_Closedform {
using mvicp::Isometry3d;
using mvicp::Vector3d;
// include/icp-closedform.h:10-11 (host only): comparison baselines of main_pairwise.cpp:74-76,93-95
Isometry3d pointToPlane(std::vector<Vector3d>& src, std::vector<Vector3d>& dst, std::vector<Vector3d>& nor);
Isometry3d pointToPoint(std::vector<Vector3d>& src, std::vector<Vector3d>& dst);
}  // namespace ICP_Closedform
