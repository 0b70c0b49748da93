// TEST INFRASTRUCTURE — builds the REAL reference NN (the vendored nanoflann.hpp v1.1.9, BSD, included
// from where it lies: /root/reference/include/nanoflann.hpp, never copied into this repo) behind a
// tiny C ABI.  Output goes to oracle/_ref/ only (git-ignored, but it travels to the GPU box).
//
// This file is OUR code: a POD dataset adaptor standing in for `class Frame`
// (/root/reference/include/frame.h:31-102 needs Eigen + gflags, absent from this image), exposing
// exactly the four adaptor methods nanoflann calls (frame.h:67-92) with the identical metric
// expression (frame.h:70-76), and driving the index exactly as src/internal/frame.cpp:187-206 does:
// KDTreeSingleIndexAdaptorParams(1 /*max leaf*/), KNNResultSet<double>(1), SearchParams(32, 0, false).
#include <cstddef>
#include <vector>

#include "nanoflann.hpp"

namespace {

struct Cloud {
  const double* pts;  // n x 3, AoS (the layout of std::vector<Eigen::Vector3d>)
  size_t n;
  inline size_t kdtree_get_point_count() const { return n; }
  inline double kdtree_distance(const double* p1, const size_t idx_p2, size_t /*size*/) const {
    const double d0 = p1[0] - pts[3 * idx_p2 + 0];
    const double d1 = p1[1] - pts[3 * idx_p2 + 1];
    const double d2 = p1[2] - pts[3 * idx_p2 + 2];
    return d0 * d0 + d1 * d1 + d2 * d2;
  }
  inline double kdtree_get_pt(const size_t idx, int dim) const { return pts[3 * idx + dim]; }
  template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};

typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<double, Cloud>, Cloud, 3> tree_t;

struct Index {
  Cloud cloud;
  std::vector<double> own;  // private copy so the caller's buffer may go away
  tree_t* tree;
};

}  // namespace

extern "C" {

void* ref_nn_build(const double* pts, int n) {
  Index* ix = new Index();
  ix->own.assign(pts, pts + 3 * (size_t)n);
  ix->cloud.pts = ix->own.data();
  ix->cloud.n = (size_t)n;
  ix->tree = new tree_t(3, ix->cloud, nanoflann::KDTreeSingleIndexAdaptorParams(1 /* max leaf */));
  ix->tree->buildIndex();
  return ix;
}

void ref_nn_free(void* h) {
  Index* ix = (Index*)h;
  delete ix->tree;
  delete ix;
}

// frame.cpp:195-205 for a batch of queries already expressed in the dst frame.
void ref_nn_query(void* h, const double* queries, int n, int* idx, double* d2) {
  Index* ix = (Index*)h;
  // (queries are independent; the pragma is live only in the -fopenmp `fast` build used by bench.py's all-cores CPU figure)
#pragma omp parallel for schedule(dynamic, 2048)
  for (int k = 0; k < n; ++k) {
    size_t ret_index = 0;
    double out_dist_sqr = 0;
    nanoflann::KNNResultSet<double> resultSet(1);
    resultSet.init(&ret_index, &out_dist_sqr);
    ix->tree->findNeighbors(resultSet, queries + 3 * k, nanoflann::SearchParams(32, 0, false));
    idx[k] = (int)ret_index;
    d2[k] = out_dist_sqr;
  }
}

// frame.cpp:208-231 getNeighbours(queryIdx, num_results): knnSearch around one of the cloud's own points.
void ref_nn_knn_self(void* h, int query_idx, int k, int* idx, double* d2) {
  Index* ix = (Index*)h;
  std::vector<size_t> ri(k);
  std::vector<double> rd(k);
  ix->tree->knnSearch(ix->cloud.pts + 3 * (size_t)query_idx, (size_t)k, &ri[0], &rd[0]);
  for (int i = 0; i < k; ++i) { idx[i] = (int)ri[i]; d2[i] = rd[i]; }
}

}  // extern "C"
