// Headless counterpart of the reference's src/main_multiview.cpp (same flags, same loop, no viewer):
//   loadFrames (:53-100) -> frames[0]->fixed = true (:141) -> computePoseNeighbours (:104-117, once) ->
//   20 x { computeClosestPoints (:119-127) ; ceresOptimizer* (:158-161) }
// Extra flags: --rounds (20), --out DIR (write final poses as pose_<i>.txt, 4x4 row-major), --device, --copyback (default on: fill
// Frame::neighbours[].correspondances every round — one device un-sort + one pinned copy for all edges, sliced by --copy_threads (8) host threads),
// --drop_phantom_row (load exactly the files' rows; default: the reference's loadXYZ, which appends a duplicate of the last row),
// --noise_stream libstdc++|libc++ (std::normal_distribution's variate order for addNoise; default = this build's libstdc++), --quiet, --dump_corr DIR (after the LAST round's search
// write every Frame::neighbours[j] as corr_<src>_<j>.txt: a header line `dst weight count`, then `first second dist` rows),
// --check_nn N (re-ask Frame::getClosestPoint for the first N correspondences of every edge and report disagreements),
// --trace FILE (every round: one line `C round src j dst count weight-bits` per edge after the search and one line `P round frame m00 .. m33`
// (4x4 row-major, 17 digits) per frame after the solve: the run's whole trajectory, for parity tests against a recorded one),
// --freeze_from R (rounds >= R search but do not solve: the poses stay bit-identical), --perturb_frame K --perturb_round R (before round R's search
// frame K's translation moves by 1e-4 m), --copy_stats (one line `copyback: round r copied X skipped Y` per round): the copy-back bookkeeping of
// host/frame.h under test; --dump_corr also writes the poses of the last search (search_pose_<i>.txt); --dump_knn FILE [--knn_k 10]: Frame::getNeighbours
// (frame.cpp:208-231) of EVERY point of frame 0, asked one by one like frame.cpp:249, as raw doubles (n x k x 3), then exit.
#include <chrono>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "common_io.h"
#include "flags.h"
#include "frame.h"

using namespace mvicp;

static void loadFrames(const Flags& F, std::vector<std::shared_ptr<Frame>>& frames, const std::string& dir) {
  const std::vector<std::string> clouds = getAllTextFilesFromFolder(dir, "cloud");
  const std::vector<std::string> poses = getAllTextFilesFromFolder(dir, "pose");
  const std::vector<std::string> groundtruth = getAllTextFilesFromFolder(dir, "groundtruth");
  if (clouds.size() != poses.size()) std::cout << "unequal size" << std::endl;
  const int limit = F.i("limit", 40), step = F.i("step", 2);
  const double sigma = F.f("sigma", 0.02), sigmat = F.f("sigmat", 0.01);
  for (int i = 0; i < (int)clouds.size() && i < limit * step; i += step) {
    std::shared_ptr<Frame> f(new Frame());
    const int j = F.b("fake", false) ? 0 : i;
    loadXYZ(clouds[j], f->pts, f->nor, !F.b("drop_phantom_row", false));
    if (F.b("recomputeNormals", true)) f->recomputeNormals();  // main_multiview.cpp:49,68-70 (default on)
    if (groundtruth.size() == clouds.size()) {
      f->pose = loadMatrix4d(poses[i]);
      f->poseGroundTruth = loadMatrix4d(groundtruth[i]);
    } else {
      f->poseGroundTruth = loadMatrix4d(poses[i]);
      f->pose = (i == 0) ? f->poseGroundTruth : addNoise(f->poseGroundTruth, sigma, sigmat);
    }
    frames.push_back(f);
  }
}

int main(int argc, char** argv) {
  Flags F(argc, argv);
  const bool pointToPlane = F.b("pointToPlane", true), sophusSE3 = F.b("sophusSE3", true), angleAxis = F.b("angleAxis", false);
  const bool robust = F.b("robust", true), quiet = F.b("quiet", false);
  const float cutoff = (float)F.f("cutoff", 0.05);
  const int knn = F.i("knn", 2), rounds = F.i("rounds", 20);
  const std::string dir = F.s("dir", "../samples/Bunny_RealData"), out = F.s("out", "");
  Session::get().device = F.i("device", 0);
  Session::get().copy_back = F.b("copyback", true);
  Session::get().copy_threads = F.i("copy_threads", 8);
  noiseStream() = F.s("noise_stream", "libstdc++") == "libc++" ? 1 : F.s("noise_stream", "libstdc++") == "g++" ? 2 : 0;

  std::vector<std::shared_ptr<Frame>> frames;
  loadFrames(F, frames, dir);
  if (frames.empty()) { std::cerr << "no frames loaded from " << dir << std::endl; return 1; }
  if (!F.s("dump_knn", "").empty()) {
    try {
      const size_t k = (size_t)F.i("knn_k", 10);
      std::ofstream f(F.s("dump_knn", "").c_str(), std::ios::binary);
      for (int i = 0; i < (int)frames[0]->pts.size(); ++i) {
        const std::vector<Vector3d> nb = frames[0]->getNeighbours(i, k);
        for (const Vector3d& p : nb) f.write(reinterpret_cast<const char*>(p.data()), 24);
      }
    } catch (const std::exception& ex) { std::cerr << ex.what() << std::endl; return 2; }
    return 0;
  }
  frames[0]->fixed = true;
  for (int i = 0; i < (int)frames.size(); ++i) frames[i]->computePoseNeighboursKnn(&frames, i, knn);
  if (!quiet) {
    std::cout << "graph adjacency matrix == block structure" << std::endl;
    for (size_t i = 0; i < frames.size(); ++i) {
      std::vector<int> row(frames.size(), 0);
      for (const OutgoingEdge& e : frames[i]->neighbours) row[e.neighbourIdx] = 1;
      for (int v : row) std::cout << v << " ";
      std::cout << std::endl;
    }
  }
  double wall_search = 0.0, wall_solve = 0.0;
  std::ofstream trace;
  if (!F.s("trace", "").empty()) { trace.open(F.s("trace", "").c_str()); trace.precision(17); }
  try {
    for (int r = 0; r < rounds; ++r) {
      if (r == F.i("perturb_round", -1) && F.i("perturb_frame", -1) >= 0 && F.i("perturb_frame", -1) < (int)frames.size())
        frames[F.i("perturb_frame", -1)]->pose.m[12] += 1e-4;
      const unsigned long long c0 = Session::get().edges_copied, s0 = Session::get().edges_skipped;
      const auto t0 = std::chrono::steady_clock::now();
      for (auto& f : frames) f->computeClosestPointsToNeighbours(&frames, cutoff);
      const auto t1 = std::chrono::steady_clock::now();
      if (trace.is_open())
        for (size_t i = 0; i < frames.size(); ++i)
          for (size_t j = 0; j < frames[i]->neighbours.size(); ++j) {
            const OutgoingEdge& e = frames[i]->neighbours[j];
            unsigned int bits;
            std::memcpy(&bits, &e.weight, 4);
            trace << "C " << r << " " << i << " " << j << " " << e.neighbourIdx << " " << e.correspondances.size() << " " << bits << "\n";
          }
      if (F.b("copy_stats", false))
        std::cout << "copyback: round " << r << " copied " << Session::get().edges_copied - c0 << " skipped " << Session::get().edges_skipped - s0 << std::endl;
      if (r == rounds - 1 && !F.s("dump_corr", "").empty()) {
        for (size_t i = 0; i < frames.size(); ++i) saveMatrix4d(F.s("dump_corr", "") + "/search_pose_" + std::to_string(i) + ".txt", frames[i]->pose);
        for (size_t i = 0; i < frames.size(); ++i)
          for (size_t j = 0; j < frames[i]->neighbours.size(); ++j) {
            const OutgoingEdge& e = frames[i]->neighbours[j];
            std::ofstream f((F.s("dump_corr", "") + "/corr_" + std::to_string(i) + "_" + std::to_string(j) + ".txt").c_str());
            f.precision(17);
            f << e.neighbourIdx << " " << e.weight << " " << e.correspondances.size() << "\n";
            for (const Correspondance& c : e.correspondances) f << c.first << " " << c.second << " " << c.dist << "\n";
          }
      }
      if (r == rounds - 1 && F.i("check_nn", 0) > 0) {
        // S1' through the Frame mirror: q = dst.pose^-1 * (src.pose * p) (frame.cpp:131,136) -> dst.getClosestPoint(q) must give
        // the correspondence's neighbour and distance
        long checked = 0, bad = 0, single = 0;
        for (size_t i = 0; i < frames.size(); ++i)
          for (const OutgoingEdge& e : frames[i]->neighbours) {
            Frame& d = *frames[e.neighbourIdx];
            const Isometry3d M = d.pose.inverse() * frames[i]->pose;
            const int n = std::min<int>(F.i("check_nn", 0), (int)e.correspondances.size());
            // the batched form (one launch for the edge's queries) ...
            std::vector<Vector3d> qs(n);
            for (int k = 0; k < n; ++k) qs[k] = M * frames[i]->pts[e.correspondances[k].first];
            std::vector<size_t> idxs;
            const std::vector<double> d2s = d.getClosestPoints(qs, idxs);
            for (int k = 0; k < n; ++k) {
              const Correspondance& c = e.correspondances[k];
              ++checked;
              if ((int)idxs[k] != c.second || std::fabs(std::sqrt(d2s[k]) - c.dist) > 1e-12) ++bad;
              if (k < 8) {   // ... and the reference's one-query signature on a few of them: identical answers
                size_t idx = 0;
                const double d2 = d.getClosestPoint(qs[k], idx);
                ++single;
                if (idx != idxs[k] || d2 != d2s[k]) ++bad;
              }
            }
          }
        std::cout << "getClosestPoint check: " << checked << " queries, " << bad << " mismatches (" << single << " also asked one by one)" << std::endl;
      }
      if (F.i("freeze_from", 1 << 30) <= r) {}   // (search only: the poses stay bit-identical)
      else if (sophusSE3) ICP_Ceres::ceresOptimizer_sophusSE3(frames, pointToPlane, robust);
      else if (angleAxis) ICP_Ceres::ceresOptimizer_ceresAngleAxis(frames, pointToPlane, robust);
      else ICP_Ceres::ceresOptimizer(frames, pointToPlane, robust);
      const auto t2 = std::chrono::steady_clock::now();
      if (trace.is_open())
        for (size_t i = 0; i < frames.size(); ++i) {
          trace << "P " << r << " " << i;
          for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) trace << " " << frames[i]->pose.m[a + 4 * b];
          trace << "\n";
        }
      wall_search += std::chrono::duration<double, std::milli>(t1 - t0).count();
      wall_solve += std::chrono::duration<double, std::milli>(t2 - t1).count();
      if (!quiet)
        std::cout << "round: " << r << "  closest pts " << std::chrono::duration<double, std::milli>(t1 - t0).count() << " ms  global "
                  << std::chrono::duration<double, std::milli>(t2 - t1).count() << " ms" << std::endl;
    }
  } catch (const std::exception& ex) {
    std::cerr << ex.what() << std::endl;
    return 2;
  }
  // whole-loop wall clock of the drop-in route (Frame API + session + copy-back): parsed by bench.py / tools/dropin_bench.py
  std::cout << "loop: rounds " << rounds << " copyback " << (Session::get().copy_back ? 1 : 0) << " closest_pts_ms " << wall_search << " global_ms " << wall_solve
            << " it_per_s " << (rounds > 0 ? 1e3 * rounds / (wall_search + wall_solve) : 0.0) << std::endl;
  for (size_t i = 0; i < frames.size(); ++i) {
    if (!quiet) std::cout << "frame " << i << poseDiff(frames[i]->pose, frames[i]->poseGroundTruth);
    if (!out.empty()) saveMatrix4d(out + "/pose_" + std::to_string(i) + ".txt", frames[i]->pose);
  }
  Session::get().reset();
  return 0;
}
