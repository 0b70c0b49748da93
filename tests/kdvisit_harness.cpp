// Host-only harness around mv-lm-icp_amd/csrc/kdvisit.h (the tie order of Frame::recomputeNormals' k-NN): brute-force k-NN of every point
// of a cloud with the metric of include/frame.h:70-76 and equal distances ordered by visited_before().  Compiled by
// tests/test_knn_tie_order.py with -I <repo>/mv-lm-icp_amd/csrc; no GPU involved.
#include "kdvisit.h"
#include <cstdio>
using namespace mvicp;
extern "C" int knn_emul(const double* xyz, int n, int K, int* out) {
  std::vector<VisitNode> nodes; std::vector<int> slot;
  build_visit_tree(xyz, n, nodes, slot);
  VisitTree T{nodes.data(), slot.data()};
  #pragma omp parallel for schedule(dynamic,64)
  for (int i = 0; i < n; ++i) {
    const double qx = xyz[3*i], qy = xyz[3*i+1], qz = xyz[3*i+2];
    std::vector<double> bd(K, 1e300); std::vector<long long> bo(K, -1);
    for (int j = 0; j < n; ++j) {
      const double d0 = qx - xyz[3*j], d1 = qy - xyz[3*j+1], d2 = qz - xyz[3*j+2];
      const double d = d0*d0 + d1*d1 + d2*d2;
      double cd = d; long long co = j;
      if (!(cd < bd[K-1] || (cd == bd[K-1] && (bo[K-1] < 0 || visited_before(T, qx,qy,qz, co, bo[K-1]))))) continue;
      for (int t = 0; t < K; ++t) {
        if (cd < bd[t] || (cd == bd[t] && (bo[t] < 0 || visited_before(T, qx,qy,qz, co, bo[t])))) { std::swap(cd, bd[t]); std::swap(co, bo[t]); }
      }
    }
    for (int t = 0; t < K; ++t) out[(size_t)i*K+t] = (int)bo[t];
  }
  return (int)nodes.size();
}
